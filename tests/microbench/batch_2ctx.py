"""why do two contexts on one device halve the host batch path?  The bench line's host_path leg (512 structures of ~5k atoms,
contacts only and all candidates) with one and with two contexts, best of 3, then ONE more call of each with the library's stage laps on
(arp_debug_set("timing", 1): launch_pack / finalize_pack per device thread, stderr).  Usage: python tests/microbench/batch_2ctx.py [n_structures]"""
import ctypes as C
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402
from arpeggia_amd import _lib  # noqa: E402

import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sizes = bench.batch5k_sizes(n)
soas = bench.batch5k_share(aa, synth, sizes, list(range(n)), pool=16)
keep = []
views = [aa.atoms_from_arrays(s, keep=keep) for s in soas]
arr = (C.POINTER(_lib.arp_atoms) * n)(*[C.pointer(v) for v in views])
c2, c3 = aa.Context(0), aa.Context(0)


def run(n_ctx, only):
    handles = (C.c_void_p * n_ctx)(*[c._h for c in (c2, c3)[:n_ctx]])
    hp = aa.default_params(0.1, 6.5, contacts_only=only)
    outs = (_lib.arp_pairs * n)()
    t0 = time.perf_counter()
    st = _lib.lib.arp_contacts_atomic_batch(handles, n_ctx, arr, n, C.byref(hp), outs)
    dt = time.perf_counter() - t0
    assert st == 0, _lib.lib.arp_last_error()
    rec = sum(int(outs[k].n) for k in range(n))
    for k in range(n):
        _lib.lib.arp_pairs_free(C.byref(outs[k]))
    return dt, rec


for only in (True, False):
    for n_ctx in (1, 2, 1, 2):
        best = min(run(n_ctx, only)[0] for _ in range(3))
        print(f"{n} structures, contacts_only={only}, {n_ctx} context(s): best of 3 = {best * 1e3:8.3f} ms = {best / n * 1e6:6.2f} us per structure", flush=True)
for n_ctx in (1, 2):
    print(f"---- laps, contacts only, {n_ctx} context(s) ----", file=sys.stderr, flush=True)
    aa.debug_set("timing", 1)
    dt, rec = run(n_ctx, True)
    aa.debug_set("timing", 0)
    print(f"     whole call {dt * 1e3:8.3f} ms, {rec} records", file=sys.stderr, flush=True)

# The bench line's order of legs (fresh contexts): contacts only 1 ctx, all candidates 1 ctx, contacts only 2 ctx, all candidates 2 ctx -- every iteration printed,
# the third 2-context contacts-only iteration with laps.
print("---- bench order, fresh contexts ----", flush=True)
c2, c3 = aa.Context(0), aa.Context(0)
for name, only, n_ctx in (("contacts_only", True, 1), ("all_candidates", False, 1), ("contacts_only_2ctx", True, 2), ("all_candidates_2ctx", False, 2), ("contacts_only_2ctx again", True, 2)):
    its = []
    for k in range(3):
        if name.startswith("contacts_only_2ctx") and k == 2:
            print(f"---- laps: {name}, iteration 3 ----", file=sys.stderr, flush=True)
            aa.debug_set("timing", 1)
        its.append(run(n_ctx, only)[0])
        aa.debug_set("timing", 0)
    print(f"{name:26s}: " + "  ".join(f"{t * 1e3:7.3f} ms" for t in its) + f"   best {min(its) / n * 1e6:6.2f} us per structure", flush=True)
