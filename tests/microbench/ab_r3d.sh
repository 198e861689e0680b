V=$GRAFT_REPO_ROOT/tests/microbench/build
BENCH_ARGS="--no-check --no-extras" bash tests/run_gpu_ab.sh r3d "ARP_TAIL_PCT=0" "ARP_TAIL_PCT=25" "ARP_TAIL_PCT=40" "ARP_TAIL_PCT=60" "ARPEGGIA_AMD_LIB=$V/libvar_allocnostore.so" "ARPEGGIA_AMD_LIB=$V/libvar_plainstore.so" "ARPEGGIA_AMD_LIB=$V/libvar_sc1store.so"
