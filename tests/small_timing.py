"""Device-side cost of the two reference files (BASELINE config 1-2: launch-bound, reported in microseconds, not against a roofline).
Usage (GPU box): python tests/small_timing.py"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402
import torch  # noqa: E402
from arpeggia_amd import _lib  # noqa: E402

for name in ("1ubq", "6bft"):
    s = aa.load_model(str(ROOT / "tests" / "data" / f"{name}.pdb"))
    soa = s.soa("/")
    dev = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa.items()}
    keep = []
    atoms = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    stream = torch.cuda.current_stream()
    ctx = aa.Context(0, stream=stream.cuda_stream)
    for only in (False, True):
        prm = aa.default_params(contacts_only=only)
        n = ctx.count(atoms, prm)
        out = torch.empty((max(n, 1), 4), dtype=torch.int32, device="cuda")
        for _ in range(5):
            ctx.enqueue(atoms, prm, out.data_ptr(), n); ctx.result()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        torch.cuda.synchronize()
        e0.record(stream)
        t0 = time.perf_counter()
        for _ in range(50):
            ctx.enqueue(atoms, prm, out.data_ptr(), n)
        host = (time.perf_counter() - t0) / 50 * 1e6  # host time of one enqueue (launches + the result copy): the stream cannot run faster than this
        e1.record(stream)
        ctx.result()
        whole = e0.elapsed_time(e1) / 50 * 1e3
        ctx.profile(True)
        ctx.enqueue(atoms, prm, out.data_ptr(), n); ctx.result()
        prof = {k: round(v * 1e3, 1) for k, v in ctx.profile_read().items()}
        ctx.profile(False)
        print(f"{name}: {s.n_atoms} atoms, {n} pairs out (contacts_only={only}): {whole:.0f} us per call on the stream (host side of an enqueue: {host:.0f} us); kernels us: {prof}")
