cd $GRAFT_REPO_ROOT
for n in 20000 60000 150000 300000 500000; do
  for f in 0 1 4 8 -1; do
    r=$(ARP_H_SPLIT=$f python bench.py --no-cpu-baseline --no-extras --atoms $n --steps 30 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f us/step, emit %.1f' % (d['ms_per_step']*1e3, d['roofline']['kernel_ms']*1e3))")
    echo "atoms $n split $f: $r"
  done
done
