"""PCIe-inclusive rate of the synchronous host-buffer entry point (arp_contacts_atomic with ARP_MEM_HOST in and out).
Never the bench `value`; DESIGN.md quotes it.  Usage: python tests/host_path_timing.py [atoms] > profiles/rNN_host_path.txt"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
rec = synth.gen_s2(n)
soa = aa.Structure.from_records(rec, hierarchy=True).soa("/")
ctx = aa.Context(0)
for only in (False, True):
    prm = aa.default_params(contacts_only=only)
    pairs = ctx.atomic_contacts(soa, prm)  # warm-up: allocations, first-touch
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pairs = ctx.atomic_contacts(soa, prm); ts.append(time.perf_counter() - t0)
    best = min(ts)
    print(f"S2 {n} atoms, contacts_only={only}, {len(pairs)} pairs out: host SoA (pageable) -> one pinned H2D block -> single-pass emit -> D2H of {len(pairs) * 16 / 1e6:.0f} MB -> numpy copy")
    print(f"  best of 5: {best * 1e3:.1f} ms = {28702955 / best if n == 1_000_000 else float('nan'):.3e} classified candidate pairs/s PCIe-inclusive (median {sorted(ts)[2] * 1e3:.1f} ms)")
