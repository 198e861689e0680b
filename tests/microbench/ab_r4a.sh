#!/bin/bash
# round 4: ablation variants of k_emit against the product build on one box (emit kernel time by HIP events)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in base "$@"; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-check > $OUT/$lib.json 2> $OUT/$lib.err || { echo "$lib FAILED"; tail -3 $OUT/$lib.err; continue; }
  python3 -c "
import json
d=json.load(open('$OUT/$lib.json'))
print('%-10s ms/step %.4f  kernels %s' % ('$lib', d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))"
done
