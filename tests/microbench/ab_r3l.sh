cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider -k "batch or config5" > gpurun_out/pytest_r3l.log 2>&1; tail -3 gpurun_out/pytest_r3l.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r3l.json 2> gpurun_out/bench_r3l.err || tail -3 gpurun_out/bench_r3l.err
python3 -c "
import json
d = json.load(open('gpurun_out/bench_r3l.json'))
print({k: (round(v['us_per_structure'],1), v['records_out']) for k, v in d['batch5k']['host_path'].items() if isinstance(v, dict)})
"
