#!/bin/bash
# Helper for gpurun calls: parity tests, bench line, rocprofv3 kernel stats.  Usage: bash tests/run_gpu_round.sh TAG [bench args]
TAG=${1:-r}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; rc=$?
tail -4 $OUT/pytest_$TAG.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 400 python bench.py "$@" > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; rc=$?
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
if [ $rc -ne 0 ]; then echo "bench failed rc=$rc"; exit $rc; fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --profile-steps 0 > $OUT/prof_$TAG.log 2>&1; rc=$?
tail -2 $OUT/prof_$TAG.log
find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -r head -20
exit $rc
