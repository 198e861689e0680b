#!/bin/bash
# PMC passes (separate rocprofv3 runs, kernel-trace only, as the pool requires).  TA_* counters hang rocprofv3 on this pool: never add them.  Usage: bash tests/run_gpu_pmc.sh TAG
TAG=${1:-p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
[ -f $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt ] || rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/counters_list.txt 2>&1
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $line --kernel-trace --output-format csv -d $OUT/pass$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 > $OUT/pass$i.log 2>&1
  rc=$?; echo "pass $i [$line] rc=$rc"
  if [ $rc -ge 124 ]; then exit $rc; fi
done <<'PASSES'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM
SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_INT32
FETCH_SIZE
WRITE_SIZE
TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
PASSES
python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
