#!/bin/bash
# Round 5: does the emit kernel fall off its cache above 2 x 10^6 atoms?  FETCH_SIZE and L2 hit / miss of k_emit at 1, 2 and 4 x 10^6 S2 atoms (one counter pass each).
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_big; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for n in 1000000 2000000 4000000; do
  i=0
  for line in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/n$n/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 --atoms $n > $OUT/n${n}_pass$i.log 2>&1
    rc=$?; if [ $rc -ge 124 ]; then echo "pass $i at $n: rc=$rc"; exit $rc; fi
  done
  echo "== $n atoms"; python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT/n$n | grep -A8 "k_emit<12, 1, false, false, " | head -9
  tail -1 $OUT/n${n}_pass1.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('pairs', d['config']['pairs_per_gpu'])"
done
