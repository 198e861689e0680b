V=$GRAFT_REPO_ROOT/tests/microbench/build
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider -k "groups or fuzz or random or stress or reference_files or chain" > gpurun_out/pytest_r3n.log 2>&1; tail -2 gpurun_out/pytest_r3n.log
ARPEGGIA_AMD_LIB=$V/libvar_w7.so timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider -k "groups or fuzz or random or stress or reference_files or chain or synthetic" > gpurun_out/pytest_r3n7.log 2>&1; tail -2 gpurun_out/pytest_r3n7.log
BENCH_ARGS="--no-extras" bash tests/run_gpu_ab.sh r3n "A=1" "ARPEGGIA_AMD_LIB=$V/libvar_w7.so" "A=2" "ARPEGGIA_AMD_LIB=$V/libvar_w7.so"
