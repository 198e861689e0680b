#!/bin/bash
# Round 5: the scratch-staged hole-free sequence of the task-split 12-wave kernels (k_emit<.., STAGE>) -- parity, then sizes 20k..300k per step.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
ARP_FUZZ_MID_CASES=12 timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "synthetic_clouds or mid_size or residue_rule or enqueue or memo or speculation or overflow or hydrogen_rich or contacts_only_is or capacity or table or properties" > $OUT/pytest_r5h.log 2>&1; rc=$?
tail -4 $OUT/pytest_r5h.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
for cfg in "s2 30000" "s2 60000" "s2 100000" "s1 100000" "s2 190000" "s2 250000" "s2 294000" "s2 400000"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --workload $1 --steps 50 --warmup 5 --no-cpu-baseline --no-extras --atoms $2 > $OUT/bench_r5h_$1_$2.json 2> $OUT/bench_r5h_$1_$2.err || { tail -3 $OUT/bench_r5h_$1_$2.err; exit 1; }
  python3 tests/show_bench.py $OUT/bench_r5h_$1_$2.json | head -1
done
