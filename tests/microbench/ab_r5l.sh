#!/bin/bash
# Round 5: k_place with XCD-contiguous atom ranges -- sizes 10^5 .. 8 x 10^6, S2 and S1.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cfg in "s2 100000" "s2 1000000" "s1 1000000" "s2 2000000" "s2 4000000" "s2 8000000"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --atoms $2 > $OUT/bench_r5l_$1_$2.json 2> $OUT/bench_r5l_$1_$2.err || { tail -3 $OUT/bench_r5l_$1_$2.err; exit 1; }
  echo "$cfg: $(python3 tests/show_bench.py $OUT/bench_r5l_$1_$2.json | head -1 | cut -d: -f2-)"
done
