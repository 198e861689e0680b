"""Config 5 of BASELINE.json on one GPU: a batch of ~5k-atom structures, packed launches vs one call per structure,
full candidate lists vs ARP_FLAG_CONTACTS_ONLY.  Usage (GPU box): python tests/batch_timing.py [n_structures]"""
import sys
import time

import numpy as np

_here = __import__("pathlib").Path(__file__).resolve().parent
sys.path[:0] = [str(_here), str(_here.parent)]
import arpeggia_amd as aa  # noqa: E402
import synth  # noqa: E402

n_structs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = [aa.Structure.from_records(synth.gen_s1(5000, seed=900 + k), hierarchy=True) for k in range(16)]
structs = [base[k % 16] for k in range(n_structs)]
views = [s.view("/") for s in structs]
ctx = aa.Context(0)
atoms = sum(s.n_atoms for s in structs)
canon = lambda a: a[np.lexsort((a["j"], a["i"]))]
for only in (False, True):
    prm = aa.default_params(contacts_only=only)
    ctx.atomic_contacts(views[0], prm)
    t0 = time.perf_counter()
    singles = [ctx.atomic_contacts(v, prm) for v in views]
    t_single = time.perf_counter() - t0
    pairs = sum(len(p) for p in singles)
    aa.atomic_contacts_batch([ctx], views[:4], prm)
    t0 = time.perf_counter()
    packed = aa.atomic_contacts_batch([ctx], views, prm)
    t_pack = time.perf_counter() - t0
    assert [len(p) for p in packed] == [len(p) for p in singles]
    for k in (0, 1, n_structs - 1):
        assert np.array_equal(canon(packed[k]), canon(singles[k]))
    print(f"{n_structs} structures, {atoms} atoms, {pairs} pairs out, contacts_only={only} (host buffers in, host pairs out)")
    print(f"  one call per structure : {t_single * 1e3:9.1f} ms  {t_single / n_structs * 1e6:7.0f} us/structure  {atoms / t_single:.3e} atoms/s")
    print(f"  batch call (packs if contacts-only): {t_pack * 1e3:9.1f} ms  {t_pack / n_structs * 1e6:7.0f} us/structure  {atoms / t_pack:.3e} atoms/s")
