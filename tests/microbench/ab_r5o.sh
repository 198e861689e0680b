#!/bin/bash
# Round 5: allocation chunk size of the chip-filling emit kernels -- 4096 records (product) against 2048 (lib14) and 1024 (lib15): emit + fix-up, same box, twice.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in product lib14 lib15; do
  if [ $lib = product ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/$lib.so; fi
  for cfg in "s2 1000000" "s1 1000000" "s2 500000"; do
    set -- $cfg
    timeout -k 10 200 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --atoms $2 > $OUT/bench_r5o_${lib}_$1_$2.json 2> $OUT/bench_r5o_${lib}_$1_$2.err || { tail -3 $OUT/bench_r5o_${lib}_$1_$2.err; exit 1; }
    echo "$lib $cfg: $(python3 tests/show_bench.py $OUT/bench_r5o_${lib}_$1_$2.json | head -1 | cut -d: -f2-)"
  done
done
done
