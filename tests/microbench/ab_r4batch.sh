#!/bin/bash
# round 4: the host batch path of the product against an older build on one box (tests/batch_timing.py 2048)
cd $GRAFT_REPO_ROOT
for lib in base "$@" base "$@"; do
  if [ $lib = base ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/libvar_$lib.so; fi
  echo "== $lib"; timeout -k 10 300 python tests/batch_timing.py 2048 2>&1 | grep "batch call\|one call"
done
