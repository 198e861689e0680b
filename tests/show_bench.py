"""Print the essentials of a bench.py JSON line.  Usage: show_bench.py FILE..."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    r = d["roofline"]
    print(f"{f}: {d['ms_per_step']:.4f} ms/step  {d['value']:.3e} pairs/s  dom={r['kernel']} frac={r['frac']:.3f} pipe_frac={r['pipeline_frac']:.3f}")
    print("   ", {k: round(v * 1000, 1) for k, v in r["kernels_ms"].items()})
