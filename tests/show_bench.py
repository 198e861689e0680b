"""Print the essentials of a bench.py JSON line.  Usage: show_bench.py FILE..."""
import json
import sys


def line(tag, d):
    r = d["roofline"]
    print(f"{tag}: {d['ms_per_step']:.4f} ms/step  {d['value']:.3e} pairs/s  frac={r['frac']:.3f} kernel_frac={r['kernel_frac']:.3f}  "
          + str({k: round(v * 1000, 1) for k, v in r["kernels_ms"].items()}))


for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:  # noqa: BLE001
        print(f, "unreadable:", e)
        continue
    line(f, d)
    for k in ("s1", "s2", "s2_1e5", "s1_1e5", "batch5k"):
        if k in d:
            line("   " + k, d[k])
    if "files" in d:
        print("    files", {n: (round(v["us_per_call_on_stream"], 1), round(v["get_contacts_warm_us"], 1), round(v.get("get_contacts_c_abi_warm_us", 0.0), 1), v["table_rows"])
                            for n, v in d["files"].items() if isinstance(v, dict)})
    if "sap" in d:
        print("    sap", {n: (round(v["sum_kernel_us"], 1), round(v["device_us_per_call"], 1)) for n, v in d["sap"].items() if isinstance(v, dict)})
    r = d["roofline"]
    if r.get("issue") or r.get("traffic_note"):
        print("    issue", r.get("issue"), "|", r.get("traffic_note"))
