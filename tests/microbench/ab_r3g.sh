V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-extras" bash tests/run_gpu_ab.sh r3g "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_w7.so" "ARPEGGIA_AMD_LIB=$V/libvar_c96.so" "A=2" "ARPEGGIA_AMD_LIB=$V/libvar_w7.so"
