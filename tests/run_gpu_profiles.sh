#!/bin/bash
# Everything that goes under profiles/ for a round, in one gpurun call.  Usage: bash tests/run_gpu_profiles.sh TAG
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/profiles_$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# The counter passes first: the bench line attaches `roofline.traffic` / `issue` only from a profile whose source hash is the running library's
# (bench.py), so the profile of THIS build has to exist in profiles/ before the bench runs (the copy in the box's tree; it comes home via $OUT).
bash tests/run_gpu_pmc.sh $TAG > $OUT/pmc.log 2>&1; cp gpurun_out/pmc_$TAG/summary.txt $OUT/pmc_summary.txt; grep -A30 "k_emit<12, 1, false, false" $OUT/pmc_summary.txt | head -32
PAIRS=$(timeout -k 10 200 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --profile-steps 0 | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['config']['pairs_per_gpu'])")
python3 tests/pmc_to_json.py $OUT/pmc_summary.txt $OUT/traffic.json --pairs $PAIRS --source profiles/${TAG}_pmc_summary.txt && cp $OUT/traffic.json profiles/${TAG}_traffic.json
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -3 $OUT/bench.err; exit 1; }
echo "bench done"; python3 tests/show_bench.py $OUT/bench.json | head -4
timeout -k 10 300 python tests/batch_timing.py 2048 > $OUT/batch_5k.txt 2>&1; tail -6 $OUT/batch_5k.txt
ARP_TIMING=1 timeout -k 10 300 python tests/table_scaling.py 1000000 > $OUT/table_1e6.txt 2>&1; grep "S1 " $OUT/table_1e6.txt
timeout -k 10 200 python tests/e2e_timing.py > $OUT/e2e.txt 2>&1; tail -12 $OUT/e2e.txt
timeout -k 10 200 python tests/small_timing.py > $OUT/small.txt 2>&1; tail -6 $OUT/small.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --profile-steps 0 > $OUT/stats.log 2>&1
cp "$(ls -t $(find $OUT/stats -name "*kernel_stats.csv") | head -1)" $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv  # (gpurun merges earlier calls' trees into the local copy: take $OUT/kernel_stats.csv, not whatever find lists first)
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/tests/batch_timing.py 2048 trace > $OUT/trace.log 2>&1
python3 $GRAFT_REPO_ROOT/tests/trace_overlap.py $OUT/trace > $OUT/batch_overlap.txt 2>&1; cat $OUT/batch_overlap.txt
rm -rf $OUT/trace/*/*kernel_trace.csv $OUT/trace/*/*memory_copy_trace.csv 2>/dev/null
cd $GRAFT_REPO_ROOT
for mode in deterministic contacts-only; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --$mode > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err; python3 tests/show_bench.py $OUT/bench_$mode.json | head -2
done
timeout -k 10 300 python tests/sap_timing.py 100000 1000000 > $OUT/sap.txt 2>&1; tail -2 $OUT/sap.txt
bash tests/microbench/sweep_sizes.sh > $OUT/sweep_sizes.txt 2>&1; tail -20 $OUT/sweep_sizes.txt
