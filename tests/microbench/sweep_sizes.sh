#!/bin/bash
# emit time of the product build over input sizes (S2 cloud; bench.py --no-extras)
OUT=$GRAFT_REPO_ROOT/gpurun_out/sweep_sizes; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for atoms in 700 2000 4000 8000 12000 20000 30000 40000 60000 100000 150000 190000 250000 294000 296000 400000 500000 1000000; do
  timeout -k 10 120 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --atoms $atoms > $OUT/$atoms.json 2> $OUT/$atoms.err || exit 1
  python3 -c "
import json
d=json.load(open('$OUT/$atoms.json'))
k=d['roofline']['kernels_ms']
print('%8d atoms  emit %6.1f us  fixup %5.1f  grid %5.1f  step %7.1f us' % ($atoms, k['pairs_emit']*1000, k.get('pairs_fixup', 0.0)*1000, sum(v for n, v in k.items() if n.startswith('grid_'))*1000, d['ms_per_step']*1000))"
done
