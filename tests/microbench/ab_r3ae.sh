#!/bin/bash
# mid-size inputs (768 .. 3072 tasks): 4-wave blocks (product) against 12-wave blocks with the same four-way task split
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ae; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in base v12s4; do
  for atoms in 60000 100000 150000 190000; do
    if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
    timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/${lib}_$atoms.json 2> $OUT/${lib}_$atoms.err || exit 1
    python3 -c "
import json
d=json.load(open('$OUT/${lib}_$atoms.json'))
print('$lib $atoms ms/step %.4f  kernels %s' % (d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))"
  done
done
