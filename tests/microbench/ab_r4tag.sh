#!/bin/bash
# round 4: the residue rule tested in the prefilter (experiment build: the tag rides in the prefilter record in place of |n|^2) on S1 and S2, pair counts checked
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4tag; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in base "$@"; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  for w in s1 s2; do
    timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extras --workload $w > $OUT/$lib.$w.json 2> $OUT/$lib.$w.err || { echo "$lib $w FAILED"; tail -2 $OUT/$lib.$w.err; continue; }
    python3 -c "
import json
d=json.load(open('$OUT/$lib.$w.json'))
print('%-7s %-3s ms/step %.4f pairs %d  %s' % ('$lib', '$w', d['ms_per_step'], d['config']['pairs_per_gpu'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))"
  done
done
