"""Copy / compute overlap of the batch path from a rocprofv3 trace (--kernel-trace --memory-copy-trace, CSV output).
Usage: python tests/trace_overlap.py DIR   (DIR holds *_kernel_trace.csv and *_memory_copy_trace.csv)"""
import csv
import glob
import sys

root = sys.argv[1]


def intervals(pattern, name_col):
    out = []
    for f in glob.glob(f"{root}/**/{pattern}", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                out.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row.get(name_col, ""), row.get("Queue_Id", row.get("Stream_Id", ""))))
    return sorted(out)


kern = intervals("*kernel_trace.csv", "Kernel_Name")
copy = intervals("*memory_copy_trace.csv", "Direction")
if not kern or not copy:
    sys.exit("no trace rows found")


def merged(iv):
    out = []
    for s, e, *_ in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def total(m):
    return sum(e - s for s, e in m)


def overlap(a, b):
    i = j = 0
    t = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            t += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return t


# the timed part: from the first pack kernel to the last event
t0 = min(k[0] for k in kern if "k_pack" in k[2]) if any("k_pack" in k[2] for k in kern) else kern[0][0]
kern = [k for k in kern if k[0] >= t0]
copy = [c for c in copy if c[0] >= t0]
mk, mc = merged(kern), merged(copy)
span = max(kern[-1][1], copy[-1][1]) - t0
ov = overlap(mk, mc)
queues = sorted({k[3] for k in kern})
print(f"span of the batch on the device        : {span / 1e6:9.3f} ms")
print(f"kernels busy (union over streams)      : {total(mk) / 1e6:9.3f} ms   ({len(kern)} launches on {len(queues)} queues)")
print(f"copies busy (H2D + D2H, union)         : {total(mc) / 1e6:9.3f} ms   ({len(copy)} copies)")
print(f"copy time that ran UNDER a kernel      : {ov / 1e6:9.3f} ms = {100.0 * ov / max(total(mc), 1):5.1f} % of the copy time")
print(f"device idle inside the span            : {(span - total(merged(kern + copy))) / 1e6:9.3f} ms")
by_dir = {}
for s, e, d, _ in copy:
    by_dir[d] = by_dir.get(d, 0) + (e - s)
for d, t in sorted(by_dir.items()):
    print(f"   copies {d:24s}: {t / 1e6:9.3f} ms")
