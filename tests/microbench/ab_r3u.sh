V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-extras" bash tests/run_gpu_ab.sh r3u "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_bb512.so" "ARPEGGIA_AMD_LIB=$V/libvar_bb256u8.so" "ARPEGGIA_AMD_LIB=$V/libvar_bb2048.so"
