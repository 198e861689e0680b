#!/bin/bash
# Instruction counters of the emit kernel on the S1 cloud, residue-rule kernels off and on, against the S2 cloud (one counter pass each).
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_s1; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "s1 off" "s1 on" "s2 off"; do
  set -- $cfg; w=$1_$2
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/$w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 --workload $1 --residue-runs $2 > $OUT/$w.log 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU_INT32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${w}b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --profile-steps 0 --workload $1 --residue-runs $2 > $OUT/${w}b.log 2>&1
  echo "== $w"; python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT/$w | grep -A10 "k_emit<12, 1, false, false, " | head -11; python3 $GRAFT_REPO_ROOT/tests/pmc_summary.py $OUT/${w}b | grep -A9 "k_emit<12, 1, false, false, " | head -10
done
