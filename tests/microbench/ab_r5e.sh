#!/bin/bash
# Round 5: the one-launch table of PDB-sized structures (k_table_small) -- table tests, then the warm get_contacts times and stage laps.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -p no:cacheprovider \
  -k "table or contacts_drop_in or arrow or cli or batch_over_files or no_ring or c_consumer or mmcif or planes or more_than_65535 or first_table or small_tables" > $OUT/pytest_r5e.log 2>&1; rc=$?
tail -4 $OUT/pytest_r5e.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 200 python tests/e2e_timing.py > $OUT/e2e_r5e.txt 2>&1; tail -12 $OUT/e2e_r5e.txt
timeout -k 10 200 python tests/microbench/table_laps_files.py > $OUT/table_laps_r5e.txt 2>&1; tail -30 $OUT/table_laps_r5e.txt
