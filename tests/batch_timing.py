"""Config 5 of BASELINE.json on one GPU: a batch of ~5k-atom structures through arp_contacts_atomic_batch (host arrays in, host pair lists
out: PCIe inclusive), against one call per structure; full candidate lists and ARP_FLAG_CONTACTS_ONLY.
Usage (GPU box): python tests/batch_timing.py [n_structures]"""
import ctypes as C
import os
import sys
import time

import numpy as np

_here = __import__("pathlib").Path(__file__).resolve().parent
sys.path[:0] = [str(_here), str(_here.parent)]
import arpeggia_amd as aa  # noqa: E402

if os.environ.get("ARP_TIMING"):  # (a switch of THIS script: the library reads no environment)
    aa.debug_set("timing", 1)
import synth  # noqa: E402
from arpeggia_amd import _lib  # noqa: E402

n_structs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
trace_only = len(sys.argv) > 2  # under rocprofv3: only the packed contacts-only batch, once warmed up
rng = np.random.default_rng(5)
sizes = np.clip(np.rint(rng.normal(5000.0, 500.0, 32)), 3000, 7000).astype(int)
base = [aa.Structure.from_records(synth.gen_s1(int(n), seed=900 + k), hierarchy=True) for k, n in enumerate(sizes)]
structs = [base[k % len(base)] for k in range(n_structs)]
views = [s.view("/") for s in structs]
ctx = aa.Context(0)
atoms = sum(s.n_atoms for s in structs)
canon = lambda a: a[np.lexsort((a["j"], a["i"]))]
arr = (C.POINTER(_lib.arp_atoms) * n_structs)(*[C.pointer(v) for v in views])
handles = (C.c_void_p * 1)(ctx._h)
modes = {"all": (False,), "contacts": (True,)}.get(__import__("os").environ.get("ARP_BATCH_MODE", ""), (True,) if trace_only else (True, False))
for only in modes:
    prm = aa.default_params(contacts_only=only)
    n_single = min(n_structs, 8 if trace_only else 256)
    ctx.atomic_contacts(views[0], prm)
    t0 = time.perf_counter()
    singles = [ctx.atomic_contacts(v, prm) for v in views[:n_single]]
    t_single = (time.perf_counter() - t0) / n_single
    outs = (_lib.arp_pairs * n_structs)()
    for rep in range(3):  # the C entry point alone (no Python per structure); first pass allocates the staging blocks
        t0 = time.perf_counter()
        st = _lib.lib.arp_contacts_atomic_batch(handles, 1, arr, n_structs, C.byref(prm), outs)
        t_pack = time.perf_counter() - t0
        assert st == 0, _lib.lib.arp_last_error()
        pairs = sum(int(outs[k].n) for k in range(n_structs))
        if rep == 2:
            for k in (0, 1, n_single - 1):
                got = np.frombuffer((C.c_char * (outs[k].n * 16)).from_address(outs[k].data), dtype=aa.PAIR_DTYPE)
                assert np.array_equal(canon(got), canon(singles[k]))
        for k in range(n_structs):
            _lib.lib.arp_pairs_free(C.byref(outs[k]))
    print(f"{n_structs} structures, {atoms} atoms, {pairs} pairs out, contacts_only={only} (host buffers in, host pairs out)")
    print(f"  one call per structure : {t_single * 1e6:7.1f} us/structure  {atoms / n_structs / t_single:.3e} atoms/s")
    print(f"  batch call (packed)    : {t_pack * 1e3:9.1f} ms  {t_pack / n_structs * 1e6:7.1f} us/structure  {atoms / t_pack:.3e} atoms/s  {pairs / t_pack:.3e} pairs/s")
