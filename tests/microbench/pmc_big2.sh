#!/bin/bash
# FETCH_SIZE of k_emit at 4 x 10^6 S2 atoms: cell rows in layer order (ARP_BENCH_STRIP_ROWS=1) against y strips of 16 rows.
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_big2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for rows in 1 16; do
  export ARP_BENCH_STRIP_ROWS=$rows
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rows$rows/pass1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 --atoms 4000000 > $OUT/rows${rows}.log 2>&1
  rc=$?; if [ $rc -ge 124 ]; then echo "rows $rows: rc=$rc"; exit $rc; fi
  echo "== rows $rows"; python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT/rows$rows | grep -A2 "k_emit<12, 1, false, false, " | head -3
done
