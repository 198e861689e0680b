#!/bin/bash
# small inputs: 4-wave blocks with the eight-way split (product below 768 tasks) against 12-wave / 4-wave blocks with the four-way split
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3ag; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { lib=$1; atoms=$2
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/${lib}_$atoms.json 2> $OUT/${lib}_$atoms.err || exit 1
  python3 -c "
import json
d=json.load(open('$OUT/${lib}_$atoms.json'))
print('$lib $atoms emit %.1f  ms/step %.4f' % (d['roofline']['kernels_ms']['pairs_emit']*1000, d['ms_per_step']))"; }
for atoms in 2000 4000 8000 12000 20000 30000 40000; do for lib in base vD vE vF; do run $lib $atoms; done; done
