V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-extras --no-check" bash tests/run_gpu_ab.sh r3q "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_fixnocopy.so" "ARPEGGIA_AMD_LIB=$V/libvar_fixnoplan.so"
