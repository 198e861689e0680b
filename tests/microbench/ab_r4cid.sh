#!/bin/bash
# round 4: where k_cellid should stop reading all the atoms itself (BOX = 2) and take k_bounds' partial boxes instead
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4cid; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in base "$@"; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  echo "== $lib"
  timeout -k 10 120 python tests/small_timing.py 2>&1 | grep "per call" | cut -c1-150
  for atoms in 700 2000 4000 8000 12000; do
    timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/$lib.$atoms.json 2> $OUT/$lib.$atoms.err || { echo FAILED; continue; }
    python3 -c "
import json
d=json.load(open('$OUT/$lib.$atoms.json')); k=d['roofline']['kernels_ms']
print('%8d atoms  step %6.1f us  %s' % ($atoms, d['ms_per_step']*1000, {n: round(v*1000,1) for n, v in k.items()}))"
  done
done
