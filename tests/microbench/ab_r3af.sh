#!/bin/bash
# where the task split pays with 12-wave blocks: tiny inputs (vB: 12 waves, 8-way split) and the range above 3072 tasks (vC: 4-way split up to 8192 tasks)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3af; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
run() { lib=$1; atoms=$2
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/${lib}_$atoms.json 2> $OUT/${lib}_$atoms.err || exit 1
  python3 -c "
import json
d=json.load(open('$OUT/${lib}_$atoms.json'))
print('$lib $atoms ms/step %.4f  kernels %s' % (d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))"; }
for atoms in 4000 8000 20000 40000; do run base $atoms; run vB $atoms; done
for atoms in 250000 400000 500000; do run v12s4 $atoms; run vC $atoms; done
