#!/bin/bash
# Round 5, first GPU call: the residue-rule kernels (k_emit<.., RES>) -- parity subset, then S2 / S1 with the kernels on and off in one run.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "residue_rule or synthetic_clouds or s1_cloud_1e6 or reference_files or config5 or more_than_65535 or contacts_only_is or hydrogen_rich" > $OUT/pytest_r5a.log 2>&1; rc=$?
tail -8 $OUT/pytest_r5a.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for cfg in "s2 auto" "s2 on" "s1 off" "s1 on" "s1 auto" "s2 auto" "s1 on"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --workload $1 --residue-runs $2 --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $OUT/bench_r5a_$1_$2.json 2> $OUT/bench_r5a_$1_$2.err || { tail -5 $OUT/bench_r5a_$1_$2.err; exit 1; }
  python3 tests/show_bench.py $OUT/bench_r5a_$1_$2.json | head -1
done
