#!/bin/bash
# round 4: SAP neighbour sum, waves per task (9 / 3 / 1) over input sizes
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4sap; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in sap_s9 sap_s3 sap_s1; do
  export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so
  echo "== $lib"; timeout -k 10 400 python tests/sap_timing.py 30000 100000 300000 1000000 2>&1 | grep sum_kernel_us | python3 -c "
import sys,ast
for l in sys.stdin:
    d=ast.literal_eval(l); print(d['atoms'], d['side_chain_atoms'], 'sum %.1f us' % d['sum_kernel_us'], 'call %.1f us' % d['device_us_per_call'])"
done | tee $OUT/sap_split.txt
