"""SAP neighbour sum: device time of the grid build and of the sum kernel on S1 clouds, through arp_sap_neighbor_sum with
the engine's per-kernel HIP events.  Usage: python tests/sap_timing.py [atoms ...]   (ARPEGGIA_AMD_LIB selects a variant build)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402


def measure(ctx, n_atoms, reps=5, rec=None):
    rec = synth.gen_s1(n_atoms) if rec is None else rec
    n = len(rec["x"])
    backbone = np.isin(rec["name"], [b"N", b"CA", b"C", b"O", b"OXT"])
    side = (~backbone) & (rec["resn"] != b"HOH") & (rec["element"] != b"H")
    w = np.random.default_rng(3).uniform(-0.5, 0.5, n).astype(np.float32)
    aa.sap_neighbor_sum(ctx, rec["x"], rec["y"], rec["z"], side, w, 5.0)
    ctx.profile(True)
    acc = {}
    for _ in range(reps):
        out = aa.sap_neighbor_sum(ctx, rec["x"], rec["y"], rec["z"], side, w, 5.0)
        for k, v in ctx.profile_read().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    ctx.profile(False)
    n_sc = int(side.sum())
    total = sum(acc.values())
    alg = 36.0 * n_sc + 4.0 * n_sc  # 36 B per side-chain atom read + 4 B written
    return {"atoms": n, "side_chain_atoms": n_sc, "kernels_us": {k: round(v * 1e3, 1) for k, v in acc.items()}, "device_us_per_call": total * 1e3,
            "sum_kernel_us": acc.get("sap_sum", 0.0) * 1e3, "atoms_per_s": n_sc / (total * 1e-3) if total else None,
            "algorithmic_GBps": alg / (total * 1e-3) / 1e9 if total else None, "checksum": float(np.abs(out).sum())}


if __name__ == "__main__":
    ctx = aa.Context(0)
    for a in [int(v) for v in sys.argv[1:]] or [100_000, 1_000_000]:
        print(measure(ctx, a), flush=True)
