#!/bin/bash
# Parity tests (optionally under an env set), then A/B bench lines (each further arg = one env set).
# Usage: bash tests/run_gpu_quick.sh TAG "TEST_ENV=.." "ENV.." ...
TAG=$1; shift
TEST_ENV=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
env $TEST_ENV timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider > $OUT/pytest_$TAG.log 2>&1; rc=$?
tail -15 $OUT/pytest_$TAG.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
if [ $rc -ne 0 ]; then echo "pytest failed rc=$rc"; exit $rc; fi
BENCH_ARGS="--no-extras $BENCH_ARGS" bash tests/run_gpu_ab.sh $TAG "$@"
