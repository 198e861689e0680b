"""Emit-kernel time per task against the tasks-per-resident-wave ratio (does the last round of tasks leave waves idle?).
Usage (GPU box): python tests/tail_probe.py"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
for n in (786000, 1000000, 1179000, 1400000, 1572000, 2000000):
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", "--atoms", str(n), "--steps", "10", "--warmup", "2"],
                       capture_output=True, text=True)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    k = d["roofline"]["kernels_ms"]; na = d["config"]["atoms_per_gpu"]; p = d["config"]["pairs_per_gpu"]
    tasks = (na + 63) // 64
    print(f"n={na} tasks={tasks} tasks/wave={tasks / 6144:.2f} emit={k['pairs_emit'] * 1e3:.1f} us  ns/task={k['pairs_emit'] * 1e6 / tasks:.2f}  pairs/task={p / tasks:.0f}", flush=True)
