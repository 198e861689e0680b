#!/bin/bash
# Same-box A/B: cell rows in layer order (ARP_BENCH_STRIP_ROWS=1) against y strips of 16 rows (=16, forced), alternating, twice.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for rows in 1 16; do
  export ARP_BENCH_STRIP_ROWS=$rows
  for cfg in "s2 2000000" "s2 3000000" "s2 4000000" "s2 6000000" "s2 8000000"; do
    set -- $cfg
    timeout -k 10 200 python bench.py --workload $1 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --atoms $2 > $OUT/bench_r5m_${rows}_$1_$2.json 2> $OUT/bench_r5m_${rows}_$1_$2.err || { tail -3 $OUT/bench_r5m_${rows}_$1_$2.err; exit 1; }
    echo "rows=$rows $cfg: $(python3 tests/show_bench.py $OUT/bench_r5m_${rows}_$1_$2.json | head -1 | cut -d: -f2-)"
  done
done
done
