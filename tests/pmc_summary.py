"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) into per-kernel means.  Usage: pmc_summary.py DIR"""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"{root}/pass*/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "?").split("(")[0].replace("void ", "")
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for name in sorted(acc, key=lambda k: -sum(acc[k].get("SQ_BUSY_CYCLES", [0]))):
    if "arp::" not in name:
        continue
    print(name)
    for c, v in sorted(acc[name].items()):
        print(f"   {c:28s} mean {sum(v) / len(v):16.1f}  (n={len(v)})")
