cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/big
for a in 2000000 4000000 8000000; do
  timeout -k 10 280 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --atoms $a > gpurun_out/big/$a.json 2> gpurun_out/big/$a.err || { tail -2 gpurun_out/big/$a.err; exit 1; }
  python3 -c "
import json
d=json.load(open('gpurun_out/big/$a.json')); r=d['roofline']
print($a, 'ms/step %.3f pairs/s %.3e frac %.3f kernel_frac %.3f' % (d['ms_per_step'], d['value'], r['frac'], r['kernel_frac']), {k: round(v*1000,1) for k,v in r['kernels_ms'].items()})"
done
