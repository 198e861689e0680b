"""Where the table path spends its time on a large structure (S1: ubiquitin copies on a lattice).  ARP_TIMING=1 prints the stages.
Usage (GPU box): ARP_TIMING=1 python tests/table_scaling.py [n_atoms]"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rec = synth.gen_s1(n)
s = aa.Structure.from_records(rec, hierarchy=True)
ctx = aa.Context(0)
from arpeggia_amd import _lib  # noqa: E402

ref = None
for threads in (1, 4, 16):
    _lib.lib.arp_set_num_threads(threads)
    ctx.get_contacts(s)
    t0 = time.perf_counter()
    cols = ctx.get_contacts(s)
    dt = time.perf_counter() - t0
    print(f"S1 {s.n_atoms} atoms, {threads} host thread(s): get_contacts {dt * 1e3:.1f} ms, {len(cols['model'])} rows", file=sys.stderr)
    key = tuple(cols[c].tobytes() for c in ("model", "interaction", "distance", "from_atom", "to_atom", "sc_dihedral"))
    assert ref is None or key == ref, "the table must not depend on the thread count"
    ref = key
