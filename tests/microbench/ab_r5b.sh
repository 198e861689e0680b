#!/bin/bash
# Round 5: grid build -- k_cellid's atomics in flight, the cell scan in one launch.  Parity subset, then S2 / S1 10^6 and a few other sizes.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "synthetic_clouds or reference_files or full_size_properties or sparse_huge or packed_batch or more_than_65535 or sap or empty_and_tiny or model_ordinals or residue_rule" > $OUT/pytest_r5b.log 2>&1; rc=$?
tail -4 $OUT/pytest_r5b.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for cfg in "s2 1000000" "s1 1000000" "s2 250000" "s2 2000000" "s2 4000000" "s2 1000000"; do
  set -- $cfg
  timeout -k 10 250 python bench.py --workload $1 --atoms $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_r5b_$1_$2.json 2> $OUT/bench_r5b_$1_$2.err || { tail -5 $OUT/bench_r5b_$1_$2.err; exit 1; }
  python3 tests/show_bench.py $OUT/bench_r5b_$1_$2.json | head -1
done
