#!/bin/bash
# A/B bench runs in one gpurun call.  Usage: bash tests/run_gpu_ab.sh TAG "ENV1=.. ENV2=.." "ENV.." ...   (each arg = one env set)
# BENCH_ARGS adds bench flags; ablation builds need BENCH_ARGS=--no-check
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
i=0
for envset in "$@"; do
  i=$((i+1))
  env $envset timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $OUT/ab_${TAG}_$i.json 2> $OUT/ab_${TAG}_$i.err; rc=$?
  echo "== [$envset] rc=$rc"
  python3 -c "
import json,sys
d=json.load(open('$OUT/ab_${TAG}_$i.json'))
print('ms/step %.4f  value %.3e  kernels %s' % (d['ms_per_step'], d['value'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))
" || tail -3 $OUT/ab_${TAG}_$i.err
  if [ $rc -ge 124 ]; then exit $rc; fi
done
