"""End-to-end cost of the file-level drop-in `contacts(path)` on the two reference structures: parse + SoA, GPU pairs,
host table, Arrow hand-over.  Usage (GPU box): python tests/e2e_timing.py > profiles/rNN_e2e.txt"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402

if os.environ.get("ARP_TIMING"):  # (a switch of THIS script: the library reads no environment)
    aa.debug_set("timing", 1)


def best(fn, n=7):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, r


for name in ("1ubq", "6bft"):
    p = str(ROOT / "tests" / "data" / f"{name}.pdb")
    aa.contacts(p)
    t_all, df = best(lambda: aa.contacts(p))
    t_load, s = best(lambda: aa.load_model(p))
    ctx = aa.api._context(0)
    t_tab, cols = best(lambda: ctx.get_contacts(s))
    t_arrow, _ = best(lambda: aa.get_contacts(s))
    t_pairs, pr = best(lambda: ctx.atomic_contacts(s.view("/")))
    t_only, po = best(lambda: ctx.atomic_contacts(s.view("/"), aa.default_params(contacts_only=True)))
    rows = df.num_rows if hasattr(df, "num_rows") else df.height
    print(f"{name}: {s.n_atoms} atoms, {len(pr)} candidate pairs, {len(po)} contacts, {rows} table rows")
    print(f"  contacts(path) end to end          {t_all:8.2f} ms")
    print(f"    load_model (parse + hierarchy)   {t_load:8.2f} ms")
    print(f"    get_contacts -> Arrow table      {t_arrow:8.2f} ms   (column-accessor path: {t_tab:.2f} ms)")
    print(f"    of which atomic pairs, host in/out: all candidates {t_pairs:.2f} ms, contacts only {t_only:.2f} ms")
