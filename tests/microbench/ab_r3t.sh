V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-extras --no-check" bash tests/run_gpu_ab.sh r3t "ARPEGGIA_AMD_LIB=$V/libvar_noexact.so" "ARPEGGIA_AMD_LIB=$V/libvar_noexact_nostage.so" "ARPEGGIA_AMD_LIB=$V/libvar_noexact_c64.so"
