import sys, numpy as np, ctypes as C
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import arpeggia_amd as aa, synth
from arpeggia_amd import _lib
ctx = aa.Context(0)
canon = lambda p: p[np.lexsort((p["j"], p["i"]))]
structs = [aa.load_model(str(synth.DATA / "1ubq.pdb")), aa.load_model(str(synth.DATA / "6bft.pdb"))]
structs += [aa.Structure.from_records(synth.gen_stress(n_res=120 + 40 * k, seed=60 + k)) for k in range(4)]
two = synth.gen_stress(n_res=90, seed=77, n_models=2)
structs.append(aa.Structure.from_records(two))
far = synth.gen_stress(n_res=80, seed=78); far["x"] += 5.0e4
structs.append(aa.Structure.from_records(far))
empty = {k: v[:0] for k, v in two.items()}
structs.append(aa.Structure.from_records(empty))
views = [s.view("/") for s in structs]
singles = [ctx.atomic_contacts(v) for v in views]
for det in (False, True):
    got = aa.atomic_contacts_batch([ctx], views, aa.default_params(deterministic=det))
    packed = aa.atomic_contacts_batch([ctx], views, aa.default_params(deterministic=det, contacts_only=True))
    for k in range(len(views)):
        g, w = canon(got[k]), canon(singles[k])
        same = len(g) == len(w) and np.array_equal(g, w)
        print("det", det, "member", k, len(g), len(w), "equal", same)
        if not same and len(g) == len(w):
            bad = np.flatnonzero((g["i"] != w["i"]) | (g["j"] != w["j"]) | (g["dist"] != w["dist"]) | (g["kind"] != w["kind"]))
            print("   n_bad", len(bad), [(tuple(g[b]), tuple(w[b])) for b in bad[:4]])
        p, wf = canon(packed[k]), canon(singles[k][singles[k]["kind"] != 0])
        if not (len(p) == len(wf) and np.array_equal(p, wf)):
            print("   packed differs", len(p), len(wf))
