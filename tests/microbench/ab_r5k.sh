#!/bin/bash
# Round 5: strip height sweep (ARP_BENCH_STRIP_ROWS) at the sizes around the automatic choice.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for n in 2000000 4000000 8000000; do
  for rows in 4 8 16 32; do
    ARP_BENCH_STRIP_ROWS=$rows timeout -k 10 300 python bench.py --workload s2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --atoms $n > $OUT/bench_r5k_${n}_$rows.json 2> $OUT/bench_r5k_${n}_$rows.err || { tail -3 $OUT/bench_r5k_${n}_$rows.err; exit 1; }
    echo "$n strips of $rows: $(python3 tests/show_bench.py $OUT/bench_r5k_${n}_$rows.json | head -1 | cut -d: -f2-)"
  done
done
