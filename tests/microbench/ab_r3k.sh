cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider -k "table or batch or config5 or contacts or arrow or disk or consumer or mmcif" > gpurun_out/pytest_r3k.log 2>&1; tail -3 gpurun_out/pytest_r3k.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_r3k.json 2> gpurun_out/bench_r3k.err || tail -3 gpurun_out/bench_r3k.err
python3 -c "
import json
d = json.load(open('gpurun_out/bench_r3k.json'))
print(json.dumps(d['batch5k']['host_path'], indent=0))
print({k: (round(v['get_contacts_warm_us'],1), round(v['us_per_call_on_stream'],1)) for k, v in d['files'].items() if isinstance(v, dict)})
"
ARP_TIMING=1 timeout -k 10 300 python tests/table_scaling.py 1000000 > gpurun_out/table_1e6_r3k.txt 2>&1; grep "S1 " gpurun_out/table_1e6_r3k.txt; head -24 gpurun_out/table_1e6_r3k.txt
