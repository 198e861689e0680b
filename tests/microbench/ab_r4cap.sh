#!/bin/bash
# round 4: the cell budget (cells per atom the grid sizing allows before it coarsens kx) on the S1 cloud, the S2 cloud and the config-5 pack
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4cap; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in base "$@"; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  for w in "s1 1000000" "s2 1000000" "s1 100000" "batch5k 0"; do
    set -- $w
    args="--workload $1"; [ $2 != 0 ] && args="$args --atoms $2"
    timeout -k 10 200 python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-extras $args > $OUT/$lib.$1.$2.json 2> $OUT/$lib.$1.$2.err || { echo "$lib $w FAILED"; continue; }
    python3 -c "
import json
d=json.load(open('$OUT/$lib.$1.$2.json'))
print('%-6s %-8s %8s ms/step %.4f  %s' % ('$lib', '$1', '$2', d['ms_per_step'], {k: round(v*1000,1) for k,v in d['roofline']['kernels_ms'].items()}))"
  done
done
