V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-check --no-extras" bash tests/run_gpu_ab.sh r3c "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_kx1.so" "ARPEGGIA_AMD_LIB=$V/libvar_kx2.so" "ARPEGGIA_AMD_LIB=$V/libvar_noexact.so" "ARPEGGIA_AMD_LIB=$V/libvar_noqueue.so" "ARPEGGIA_AMD_LIB=$V/libvar_nocompact.so" "ARPEGGIA_AMD_LIB=$V/libvar_nostore.so" && bash tests/run_gpu_pmc.sh r3c
