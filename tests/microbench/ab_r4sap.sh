#!/bin/bash
# round 4: the SAP neighbour sum, round 3's one-thread-per-atom kernel (variant sapold) against the wave-per-64-slots kernel
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4sap; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for lib in sapold base; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  echo "== $lib"; timeout -k 10 300 python tests/sap_timing.py 100000 1000000 2>&1 | tail -3
done | tee $OUT/sap.txt
