#!/bin/bash
# Round 5: 7 waves per SIMD for the emit kernel -- 14-wave blocks with 112-record staged chunks (tests/microbench/build/lib14.so, built with
# -DARP_EWAVES=14 -DARP_ECHUNK=112: 72 vector registers, 80 160 B of LDS per block) against the product's 12 x 128.
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for lib in product lib14; do
  if [ $lib = product ]; then unset ARPEGGIA_AMD_LIB; else export ARPEGGIA_AMD_LIB=$GRAFT_REPO_ROOT/tests/microbench/build/$lib.so; fi
  for cfg in "s2 1000000 auto" "s1 1000000 off" "s2 2000000 auto" "s2 300000 auto"; do
    set -- $cfg
    timeout -k 10 200 python bench.py --workload $1 --steps 20 --warmup 3 --no-cpu-baseline --no-extras --atoms $2 --residue-runs $3 > $OUT/bench_r5i_${lib}_$1_$2.json 2> $OUT/bench_r5i_${lib}_$1_$2.err || { tail -3 $OUT/bench_r5i_${lib}_$1_$2.err; exit 1; }
    echo "$lib $cfg: $(python3 tests/show_bench.py $OUT/bench_r5i_${lib}_$1_$2.json | head -1)"
  done
done
