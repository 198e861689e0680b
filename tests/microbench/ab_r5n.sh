#!/bin/bash
# Same-box A/B on the S1 (protein-like) cloud: layer order (ARP_BENCH_STRIP_ROWS=1) against the automatic choice, alternating, twice.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for rows in 1 auto; do
  if [ $rows = auto ]; then unset ARP_BENCH_STRIP_ROWS; else export ARP_BENCH_STRIP_ROWS=$rows; fi
  for cfg in "s1 1000000" "s1 2000000" "s1 4000000"; do
    set -- $cfg
    timeout -k 10 300 python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --atoms $2 > $OUT/bench_r5n_${rows}_$1_$2.json 2> $OUT/bench_r5n_${rows}_$1_$2.err || { tail -3 $OUT/bench_r5n_${rows}_$1_$2.err; exit 1; }
    echo "rows=$rows $cfg: $(python3 tests/show_bench.py $OUT/bench_r5n_${rows}_$1_$2.json | head -1 | cut -d: -f2-)"
  done
done
done
