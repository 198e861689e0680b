#!/bin/bash
# Ad-hoc PMC passes on one library build.  TA_* counters hang rocprofv3 on this pool: never add them.
# Usage: bash tests/run_gpu_pmc2.sh TAG [LIB] ; passes are the lines of tests/pmc_passes.txt
TAG=${1:-q}; LIB=$2
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
[ -n "$LIB" ] && export ARPEGGIA_AMD_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 --no-check > $OUT/pass$i.log 2>&1
  rc=$?; echo "pass $i [$line] rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
done < $GRAFT_REPO_ROOT/tests/pmc_passes.txt
python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
grep -A40 "k_pairs<2, false>" $OUT/summary.txt | head -45
