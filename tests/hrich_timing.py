"""Per-kernel timing on a hydrogen-rich structure (every donor carries H atoms, so the deferred probe pass is busy).
Usage (GPU box): python tests/hrich_timing.py [n_res]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402
import torch  # noqa: E402
from arpeggia_amd import _lib  # noqa: E402

n_res = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.perf_counter()
box = 28.0 * (n_res / 400.0) ** (1.0 / 3.0)
rec = synth.gen_stress(n_res=n_res, seed=5, box=box)
print(f"generated {len(rec['x'])} atoms in {time.perf_counter() - t0:.1f} s (box {box:.0f} A)")
s = aa.Structure.from_records(rec)
soa = s.soa("/")
dev = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa.items()}
keep = []
atoms = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
ctx = aa.Context(0, stream=torch.cuda.current_stream().cuda_stream)
for only in (False, True):
    prm = aa.default_params(contacts_only=only)
    n = ctx.count(atoms, prm)
    out = torch.empty((max(n, 1), 4), dtype=torch.int32, device="cuda")
    for _ in range(3):
        ctx.enqueue(atoms, prm, out.data_ptr(), n); ctx.result()
    ctx.profile(True)
    acc = {}
    for _ in range(5):
        ctx.enqueue(atoms, prm, out.data_ptr(), n); ctx.result()
        for k, v in ctx.profile_read().items():
            acc[k] = acc.get(k, 0.0) + v / 5
    ctx.profile(False)
    heavy = int((soa["attr"] & _lib.ATTR["H"] == 0).sum())
    print(f"contacts_only={only}: {len(soa['x'])} atoms ({heavy} heavy), {n} pairs out; kernels us:", {k: round(v * 1e3, 1) for k, v in acc.items()}, "total", round(sum(acc.values()) * 1e3, 1))
