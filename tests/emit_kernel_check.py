"""Parity of ONE emit kernel variant against the oracle, in a process of its own (arp_debug_set("emit_kernel", ...) is process-wide).
Run by tests/test_gpu_parity.py::test_alternative_emit_kernels; usage: python tests/emit_kernel_check.py gather
(ARP_TEST_STRIP_ROWS=N in the environment: on cell rows forced into y strips of N rows)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402
import oracle_binding as ob  # noqa: E402
import synth  # noqa: E402


def canon(p):
    return p[np.lexsort((p["j"], p["i"]))]


def check(ctx, prod, orc, what, **kw):
    want = canon(orc.atomic_contacts("/", 0.1, 6.5))
    for only in (False, True):
        got = canon(ctx.atomic_contacts(prod.view("/"), aa.default_params(0.1, 6.5, contacts_only=only, **kw)))
        w = want[want["kind"] != 0] if only else want
        assert len(got) == len(w), f"{what} (contacts_only={only}): {len(got)} pairs vs oracle {len(w)}"
        assert np.array_equal(got["i"], w["i"].astype(np.uint32)) and np.array_equal(got["j"], w["j"].astype(np.uint32)) and np.array_equal(got["kind"], w["kind"]), what
        assert np.array_equal(got["dist"], w["dist"].astype(np.float32)), f"{what}: f32 distances not bit-identical"


def main():
    import os

    if os.environ.get("ARP_TEST_STRIP_ROWS"):  # (tests/conftest.py: the suite on forced y strips -- this process has to be told itself)
        aa.debug_set("strip_rows", int(os.environ["ARP_TEST_STRIP_ROWS"]))
    aa.debug_set("emit_kernel", {"default": 0, "gather": 1}[sys.argv[1] if len(sys.argv) > 1 else "gather"])
    ctx = aa.Context(0)
    for name in ("1ubq", "6bft"):
        path = str(synth.DATA / f"{name}.pdb")
        check(ctx, aa.load_model(path), ob.Structure.load(path), name)
    rec = synth.gen_stress(n_res=400, seed=3, hydrogens=True)     # hydrogens: the deferred probe pass behind this kernel
    check(ctx, aa.Structure.from_records(rec), ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False), "stress")
    rec = synth.gen_s1(250000, seed=0xA11CE5EED00 + 77)           # 3907 tasks: the whole-task kernel, not the split one
    check(ctx, aa.Structure.from_records(rec, hierarchy=True), ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True), "S1 250k")
    print("ok")


if __name__ == "__main__":
    main()
