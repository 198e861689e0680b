V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-extras" bash tests/run_gpu_ab.sh r3o "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_w5.so" "ARPEGGIA_AMD_LIB=$V/libvar_w5ra8.so" "ARPEGGIA_AMD_LIB=$V/libvar_w4ra8.so"
