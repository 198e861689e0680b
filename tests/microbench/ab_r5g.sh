#!/bin/bash
# Round 5: the cell scan folded into k_cellid's last block (inputs up to kCidScanAtoms atoms) -- parity of the small-input tests, then per-call times.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "reference_files or chain_groups or 6bft_parameters or stress or randomised or empty_and_tiny or cutoff_is or coincident or sparse_huge or bad_inputs or enqueue or batch or table or speculation or sap" > $OUT/pytest_r5g.log 2>&1; rc=$?
tail -4 $OUT/pytest_r5g.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 200 python tests/small_timing.py > $OUT/small_r5g.txt 2>&1; tail -5 $OUT/small_r5g.txt
for atoms in 700 4000 12000 20000 100000; do
  timeout -k 10 120 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/bench_r5g_$atoms.json 2> $OUT/bench_r5g_$atoms.err || exit 1
  python3 tests/show_bench.py $OUT/bench_r5g_$atoms.json | head -1
done
