#!/bin/bash
# Round 5: cell rows in y strips (arp_internal.h grid_row) -- parity with forced strips, then the sizes where the automatic choice switches them on.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider -k "y_strips or synthetic_clouds or table_matches_golden" > $OUT/pytest_r5j_strips.log 2>&1; rc=$?
tail -4 $OUT/pytest_r5j_strips.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for n in 1000000 2000000 4000000 8000000; do
  timeout -k 10 300 python bench.py --workload s2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --atoms $n > $OUT/bench_r5j_$n.json 2> $OUT/bench_r5j_$n.err || { tail -3 $OUT/bench_r5j_$n.err; exit 1; }
  echo "$n: $(python3 tests/show_bench.py $OUT/bench_r5j_$n.json | head -1)"
done
