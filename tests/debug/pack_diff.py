import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import arpeggia_amd as aa, synth
ctx = aa.Context(0)
structs = [aa.load_model(str(synth.DATA / "1ubq.pdb")), aa.load_model(str(synth.DATA / "6bft.pdb"))]
structs += [aa.Structure.from_records(synth.gen_stress(n_res=120 + 40 * k, seed=60 + k)) for k in range(4)]
two = synth.gen_stress(n_res=90, seed=77, n_models=2)
structs.append(aa.Structure.from_records(two))
far = synth.gen_stress(n_res=80, seed=78); far["x"] += 5.0e4
structs.append(aa.Structure.from_records(far))
empty = {k: v[:0] for k, v in two.items()}
structs.append(aa.Structure.from_records(empty))
for sel in (list(range(len(structs))), list(range(len(structs))), [0, 1, 8], [8, 1]):
    views = [structs[k].view("/") for k in sel]
    singles = [ctx.atomic_contacts(v) for v in views]
    got = aa.atomic_contacts_batch([ctx], views, aa.default_params())
    canon = lambda p: p[np.lexsort((p["j"], p["i"]))]
    for k in range(len(views)):
        g, w = canon(got[k]), canon(singles[k])
        print(sel, k, len(g), len(w), "ij", np.array_equal(g["i"], w["i"]) and np.array_equal(g["j"], w["j"]) if len(g) == len(w) else None,
              "dist", np.array_equal(g["dist"], w["dist"]) if len(g) == len(w) else None, "kind", int((g["kind"] != w["kind"]).sum()) if len(g) == len(w) else None)
        if len(g) == len(w) and (g["kind"] != w["kind"]).any():
            bad = np.flatnonzero(g["kind"] != w["kind"])[:5]
            print("   ", [(int(g["i"][b]), int(g["j"][b]), float(g["dist"][b]), hex(int(g["kind"][b])), hex(int(w["kind"][b]))) for b in bad])
