#!/bin/bash
# Round 4, small inputs: parity subset, then per-call times of the file-sized structures and the size sweep.
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4small; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider > $OUT/pytest.log 2>&1; rc=$?
tail -8 $OUT/pytest.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 200 python tests/small_timing.py > $OUT/small.txt 2>&1; tail -6 $OUT/small.txt
bash tests/microbench/sweep_sizes.sh > $OUT/sweep.txt 2>&1; cat $OUT/sweep.txt
