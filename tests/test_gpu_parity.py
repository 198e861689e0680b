"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): bit-exact contact-pair indices and interaction-type flags; distances within 1e-5 A.
Run with `pytest -m gpu` on an MI355X box.  Reference citations are to y1zhou/arpeggia v0.8.0.
"""
import ctypes as C
import os

import numpy as np
import pytest

import arpeggia_amd as aa
import oracle_binding as ob
import synth
from arpeggia_amd import _lib
from conftest import GOLDEN

pytestmark = pytest.mark.gpu

DIST_TOL = 1e-5  # Angstrom (north_star: distances within 1e-5 A)
ANGLE_TOL_DEG = 5.7e-3  # degrees = 1e-4 rad (north_star: angles within 1e-4 rad)


@pytest.fixture(scope="module")
def ctx():
    assert aa.device_count() >= 1, "no gfx950 device: the product has no CPU fallback"
    return aa.Context(0)


def canon(p):
    return p[np.lexsort((p["j"], p["i"]))]


def assert_pairs_equal(got, want, what=""):
    """got: product pairs (PAIR_DTYPE); want: oracle pairs (i, j, dist f64, kind)."""
    assert len(got) == len(want), f"{what}: {len(got)} pairs vs oracle {len(want)}"
    g, w = canon(got), canon(want)
    assert np.array_equal(g["i"], w["i"].astype(np.uint32)) and np.array_equal(g["j"], w["j"].astype(np.uint32)), f"{what}: pair indices differ"
    bad = np.flatnonzero(g["kind"] != w["kind"])
    assert len(bad) == 0, f"{what}: {len(bad)} kind mismatches, first: i={g['i'][bad[0]]} j={g['j'][bad[0]]} got={g['kind'][bad[0]]:#x} want={w['kind'][bad[0]]:#x} d={w['dist'][bad[0]]}"
    assert np.abs(g["dist"].astype(np.float64) - w["dist"]).max(initial=0.0) <= DIST_TOL, f"{what}: distance tolerance"
    # the table stores (f32) distance (mod.rs:148): the narrowing of the correctly rounded f64 distance is reproduced bit for bit
    exact = float((g["dist"] == w["dist"].astype(np.float32)).mean()) if len(g) else 1.0
    assert exact == 1.0, f"{what}: f32 distances not bit-identical ({exact})"
    return exact


def run_both(ctx, prod, orc, groups="/", vdw_comp=0.1, cutoff=6.5):
    """Both emitters against the oracle: the default single-pass one is returned, the ordered one is checked here."""
    want = orc.atomic_contacts(groups, vdw_comp, cutoff)
    ordered = ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff, deterministic=True))
    assert_pairs_equal(ordered, want, "ordered emitter")
    got = ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff))
    return got, want


# ---------------------------------------------------------------------------------------------- the two reference files
@pytest.mark.parametrize("name,n_pairs", [("1ubq", 9128), ("6bft", 124047)])
def test_reference_files_atomic_parity(ctx, name, n_pairs):
    path = str(synth.DATA / f"{name}.pdb")
    prod, orc = aa.load_model(path), ob.Structure.load(path)
    got, want = run_both(ctx, prod, orc)
    assert len(want) == n_pairs
    exact = assert_pairs_equal(got, want, name)
    assert exact == 1.0, f"f32 distances not bit-identical: {exact}"


@pytest.mark.parametrize("groups", ["A/", "A,B/C,G", "H,L/H,L,A", "/G", "A,B,C,G,H,L/A,B,C,G,H,L", "G/G"])
def test_6bft_chain_groups(ctx, groups):
    path = str(synth.DATA / "6bft.pdb")
    prod, orc = aa.load_model(path), ob.Structure.load(path)
    got, want = run_both(ctx, prod, orc, groups)
    assert len(want) > 0
    assert_pairs_equal(got, want, groups)


@pytest.mark.parametrize("vdw_comp,cutoff", [(0.1, 4.0), (0.0, 6.5), (0.25, 5.0), (0.1, 12.0), (1.0, 3.0), (0.1, 0.5), (0.1, 100.0)])
def test_6bft_parameters(ctx, vdw_comp, cutoff):
    path = str(synth.DATA / "6bft.pdb")
    prod, orc = aa.load_model(path), ob.Structure.load(path)
    if cutoff > 50:
        path = str(synth.DATA / "1ubq.pdb")  # all-pairs regime: keep it small
        prod, orc = aa.load_model(path), ob.Structure.load(path)
    got, want = run_both(ctx, prod, orc, "/", vdw_comp, cutoff)
    assert_pairs_equal(got, want, f"c={vdw_comp} d={cutoff}")


# ---------------------------------------------------------------------------------------------- rule coverage
@pytest.mark.parametrize("kw,groups", [
    (dict(n_res=400, seed=7), "/"),
    (dict(n_res=300, seed=8), "A,B/B,C,D"),
    (dict(n_res=200, seed=9, n_models=3), "/"),
    (dict(n_res=200, seed=10, altlocs=True), "A/"),
    (dict(n_res=300, seed=11, hydrogens=False), "/"),
])
def test_stress_structures_cover_every_atomic_rule(ctx, kw, groups):
    rec = synth.gen_stress(**kw)
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    got, want = run_both(ctx, prod, orc, groups)
    assert_pairs_equal(got, want, str(kw))
    if kw.get("hydrogens", True) and kw["n_res"] >= 400 and groups == "/":
        seen = np.bitwise_or.reduce(want["kind"])
        for name in ("StericClash", "CovalentBond", "Disulfide", "VanDerWaalsContact", "IonicBond", "HydrogenBond", "WeakHydrogenBond",
                     "PolarContact", "WeakPolarContact", "IonicRepulsion", "SaltBridge", "HydrophobicContact"):
            assert seen & (1 << ob.INTERACTIONS.index(name)), f"stress input never produced {name}"


@pytest.mark.parametrize("vdw_comp", [-0.1, -0.6])
def test_negative_vdw_comp_keeps_the_first_match_order(ctx, vdw_comp):
    """vdw.rs:32-43 with a negative compensation: cov - c > cov + c, so the three bounds are not nested; a distance between them is a
    StericClash (first match), never a CovalentBond.  Every emitter against the oracle."""
    rec = synth.gen_stress(n_res=400, seed=7)
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    got, want = run_both(ctx, prod, orc, "/", vdw_comp, 6.5)
    assert_pairs_equal(got, want, f"c={vdw_comp}")
    only = ctx.atomic_contacts(prod.view("/"), aa.default_params(vdw_comp, 6.5, contacts_only=True))
    assert_pairs_equal(only, want[want["kind"] != 0], f"contacts only, c={vdw_comp}")
    assert (want["kind"] & (1 << ob.INTERACTIONS.index("StericClash"))).any()


def test_radii_with_vdw_below_cov_agree_between_the_emitters(ctx):
    """Caller radii whose van-der-Waals sum lies below the covalent sum (the API accepts them; the oracle has no radius knob): the
    single-pass emitter (distance levels counted, k_emit), the ordered one (first-match booleans, classify_fast) and the probe variant
    (contacts-only ordered fill, classify<true>) must agree record for record."""
    rec = synth.gen_stress(n_res=300, seed=12)
    prod = aa.Structure.from_records(rec)
    view = prod.view("/")
    base = aa.default_params(0.1, 6.5)
    for k in range(16):
        base.vdw_radius[k] = 0.4 + 0.05 * k   # far below the covalent radii of the classes that occur
    def with_flags(**kw):
        p = aa.default_params(0.1, 6.5, **kw)
        for k in range(16):
            p.vdw_radius[k] = base.vdw_radius[k]
        return p
    emit = canon(ctx.atomic_contacts(view, with_flags()))
    ordered = canon(ctx.atomic_contacts(view, with_flags(deterministic=True)))
    probes = canon(ctx.atomic_contacts(view, with_flags(deterministic=True, contacts_only=True)))
    assert len(emit) == len(ordered) and np.array_equal(emit["i"], ordered["i"]) and np.array_equal(emit["j"], ordered["j"])
    assert np.array_equal(emit["kind"], ordered["kind"])
    kept = emit[emit["kind"] != 0]
    assert len(kept) == len(probes) and np.array_equal(kept["kind"], probes["kind"])
    assert not (emit["kind"] & (1 << ob.INTERACTIONS.index("VanDerWaalsContact"))).any()  # the vdW band is empty: it lies inside the covalent one


def test_more_than_65535_chains(ctx):
    """API v2: chain ranks are 32-bit.  70 000 one-residue chains whose ids do not sort in file order (the reference keys on the chain id
    string and orders `cx > cy` byte-wise, complex.rs:124-129): the pair list against the oracle with both emitters and contacts only, the
    chain-group forms that need ranks on both sides, and the device table's chain columns / sort with 17-bit chain keys."""
    rec = synth.gen_many_chains(70000)
    assert len(np.unique(rec["chain"])) == 70000
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    soa = prod.soa("/")
    assert soa["chain_rank"].dtype == np.uint32 and int(soa["chain_rank"].max()) == 69999
    got, want = run_both(ctx, prod, orc)
    assert len(want) > 500_000
    assert_pairs_equal(got, want, "70 000 chains")
    only = ctx.atomic_contacts(prod.view("/"), aa.default_params(contacts_only=True))
    assert_pairs_equal(only, want[want["kind"] != 0], "70 000 chains, contacts only")
    # six chains against all the others (the L/R bits, the cross-chain rule with ranks beyond 2^16).  Expected = the all-against-all pairs with
    # exactly one atom in the six chains, that atom first (no hydrogens here: a pair's kinds do not depend on its orientation) -- a second
    # pass of the oracle, which compares chain id strings per pair, would take another half minute.
    some = np.unique(rec["chain"])[::11000][:6]
    groups = ",".join(c.decode() for c in some) + "/"
    in_l = np.isin(rec["chain"], some)
    li, lj = in_l[want["i"]], in_l[want["j"]]
    pick = li != lj
    want_g = want[pick].copy()
    flip = lj[pick]
    want_g["i"], want_g["j"] = np.where(flip, want["j"][pick], want["i"][pick]), np.where(flip, want["i"][pick], want["j"][pick])
    assert len(want_g) > 100
    for det in (False, True):
        assert_pairs_equal(ctx.atomic_contacts(prod.view(groups), aa.default_params(deterministic=det)), want_g, f"{groups} det={det}")
    # the table (the oracle's own table takes ten minutes on this many chains): one row per set bit of the oracle's pair kinds -- the
    # residues are too small for ring rows -- with the chain columns in the reference's sort order (mod.rs:120-134: model, from_chain, to_chain)
    cols = ctx.get_contacts(prod, "/", 0.1, 6.5)
    n_rows = int(sum(bin(int(k)).count("1") for k in np.unique(want["kind"]) for _ in range(int((want["kind"] == k).sum()))))
    assert len(cols["model"]) == n_rows > 100_000
    key = np.char.add(cols["from_chain"].astype("S8"), cols["to_chain"].astype("S8"))
    assert (key[1:] >= key[:-1]).all() and np.unique(cols["from_chain"])[-1] > b"K065536" and np.unique(cols["to_chain"])[-1] > b"K065536"  # (ranks beyond 16 bits in both columns)


def test_cys_without_cb_is_an_error_like_the_reference_panic(ctx):
    # vdw.rs:55-58: cb1 = residue.atoms().find(CB).unwrap()
    rec = synth.gen_stress(n_res=60, seed=21, hydrogens=False)
    # two cysteines' SG atoms 2.05 A apart, CB removed
    n = 4
    extra = {k: v[:n].copy() for k, v in rec.items()}
    extra["name"][:] = [b"SG", b"CA", b"SG", b"CA"]
    extra["resn"][:] = b"CYS"
    extra["element"][:] = [b"S", b"C", b"S", b"C"]
    extra["chain"][:] = [b"Y", b"Y", b"Z", b"Z"]
    extra["resi"][:] = [1, 1, 5, 5]
    extra["x"][:] = [500.0, 501.5, 502.05, 503.5]; extra["y"][:] = 500.0; extra["z"][:] = 500.0
    extra["serial"][:] = np.arange(90001, 90001 + n)
    rec2 = {k: np.concatenate([rec[k], extra[k]]) for k in rec}
    prod = aa.Structure.from_records(rec2)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec2, flat=False), flat=False)
    with pytest.raises(ob.OracleError):
        orc.atomic_contacts()
    with pytest.raises(aa.ArpeggiaError) as e:
        ctx.atomic_contacts(prod.view("/"))
    assert e.value.status == _lib.ARP_ERR_BAD_INPUT


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("ARP_FUZZ_CASES", "24"))))  # ARP_FUZZ_CASES=300 for a soak run
def test_randomised_structures_and_parameters(ctx, case):
    """Random sizes, densities, chain groups, cutoffs and compensation factors: every combination goes through both emitters,
    the contacts-only filter and the oracle (covers the probe hand-off of the hot kernel on hydrogen-rich and hydrogen-free input)."""
    rng = np.random.default_rng(9000 + case)
    n_res = int(rng.integers(40, 700))
    n_chains = int(rng.integers(1, 6))
    kw = dict(n_res=n_res, seed=500 + case, box=float(rng.uniform(14.0, 60.0)), hydrogens=bool(rng.integers(0, 2)), n_models=int(rng.integers(1, 3)),
              n_chains=n_chains, altlocs=bool(rng.integers(0, 4) == 0))
    chains = [chr(ord("A") + c) for c in range(n_chains)]
    pick = lambda: ",".join(sorted(rng.choice(chains, size=int(rng.integers(1, n_chains + 1)), replace=False)))
    groups = ["/", pick() + "/", "/" + pick(), pick() + "/" + pick()][int(rng.integers(0, 4))]
    vdw_comp, cutoff = float(rng.choice([0.0, 0.05, 0.1, 0.3])), float(rng.choice([3.0, 4.4, 4.5, 5.0, 6.5, 8.0]))
    rec = synth.gen_stress(**kw)
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    try:
        want = orc.atomic_contacts(groups, vdw_comp, cutoff)
    except ob.OracleError:  # a CYS SG..SG pair without CB (the reference panics): the product must refuse too
        with pytest.raises(aa.ArpeggiaError):
            ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff))
        return
    what = f"{kw} groups={groups} c={vdw_comp} d={cutoff}"
    for det in (False, True):
        assert_pairs_equal(ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff, deterministic=det)), want, what)
        only = ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff, deterministic=det, contacts_only=True))
        assert_pairs_equal(only, want[want["kind"] != 0], what + " contacts-only")


@pytest.mark.parametrize("case", range(int(__import__("os").environ.get("ARP_FUZZ_MID_CASES", "6"))))  # ARP_FUZZ_MID_CASES=60 for a soak run
def test_randomised_mid_size_structures_through_both_kernel_families(ctx, case):
    """The randomised structures above stay below 20 480 atoms (the 4-wave kernels).  These are large enough for the 12-wave kernels with the
    four-way task split -- with and without hydrogens, several chains and models, random chain groups and parameters -- and go through the plain
    and the residue-rule kernels (round 5), all candidates and contacts only, against the oracle."""
    rng = np.random.default_rng(7100 + case)
    n_res = int(rng.integers(2800, 6000))
    hydrogens = bool(rng.integers(0, 2))
    n_chains = int(rng.integers(1, 6))
    kw = dict(n_res=n_res if not hydrogens else n_res // 2 + 1400, seed=900 + case, box=float(rng.uniform(60.0, 110.0)), hydrogens=hydrogens,
              n_models=int(rng.integers(1, 3)), n_chains=n_chains, altlocs=bool(rng.integers(0, 4) == 0))
    chains = [chr(ord("A") + c) for c in range(n_chains)]
    pick = lambda: ",".join(sorted(rng.choice(chains, size=int(rng.integers(1, n_chains + 1)), replace=False)))
    groups = ["/", "/", pick() + "/", pick() + "/" + pick()][int(rng.integers(0, 4))]
    vdw_comp, cutoff = float(rng.choice([0.0, 0.1, 0.3])), float(rng.choice([4.5, 5.0, 6.5, 8.0]))
    rec = synth.gen_stress(**kw)
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    try:
        want = orc.atomic_contacts(groups, vdw_comp, cutoff)
    except ob.OracleError:  # a CYS SG..SG pair without CB (the reference panics): the product must refuse too
        with pytest.raises(aa.ArpeggiaError):
            ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff))
        return
    what = f"{kw} groups={groups} c={vdw_comp} d={cutoff}"
    assert prod.n_atoms > 20480, what
    for runs in (True, False):
        for only in (False, True):
            got = ctx.atomic_contacts(prod.view(groups), aa.default_params(vdw_comp, cutoff, contacts_only=only, residue_runs=runs))
            assert_pairs_equal(got, want[want["kind"] != 0] if only else want, f"{what} residue_runs={runs} only={only}")


def test_hydrogen_rich_medium_structure(ctx):
    # 5000 residues with explicit hydrogens (~63k atoms): the deferred probe pass carries a large share of the pairs
    rec = synth.gen_stress(n_res=5000, seed=77, box=28.0 * (5000 / 400.0) ** (1.0 / 3.0))
    prod = aa.Structure.from_records(rec)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    got, want = run_both(ctx, prod, orc)
    assert len(want) > 1_000_000
    assert_pairs_equal(got, want, "hydrogen-rich 5000 residues")
    seen = np.bitwise_or.reduce(want["kind"])
    for name in ("HydrogenBond", "WeakHydrogenBond", "SaltBridge", "Disulfide"):
        assert seen & (1 << ob.INTERACTIONS.index(name)), name


# ---------------------------------------------------------------------------------------------- synthetic clouds
@pytest.mark.parametrize("gen,n", [("s2", 20000), ("s1", 20000), ("s2", 100000), ("s1", 100000)])
def test_synthetic_clouds_vs_oracle(ctx, gen, n):
    rec = getattr(synth, f"gen_{gen}")(n)
    prod = aa.Structure.from_records(rec, hierarchy=True)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    got, want = run_both(ctx, prod, orc)
    assert len(want) > 10 * n
    assert_pairs_equal(got, want, f"{gen} {n}")


def test_residue_rule_kernels_emit_the_same_lists(ctx):
    """Round 5: k_emit<.., RES> drops prefilter survivors of the home atom's own residue and of its sequence neighbours (complex.rs:108-113)
    BEFORE the exact phase, on one 32-bit residue word per atom (chain rank << 20 | ordinal).  The early exit must never change the list:
    asked for (ARP_FLAG_RESIDUE_RUNS), ruled out (ARP_FLAG_NO_RESIDUE_RUNS) and chosen by the engine's memo, against the oracle, with all
    candidates and contacts only, on the sizes that select the 12-wave kernels (task split 4 and 1), with hydrogens (the deferred probes),
    and on inputs whose ordinals / chain ranks do not fit the word (the rule is then switched off on the device)."""
    cases = []
    for n in (30000, 330000):  # 12-wave kernels with the four-way task split / without
        rec = synth.gen_s1(n)
        soa = aa.Structure.from_records(rec, hierarchy=True).soa("/")
        want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()
        cases.append((f"s1 {n}", soa, want))
    rec = synth.gen_stress(n_res=5000, seed=77, box=28.0 * (5000 / 400.0) ** (1.0 / 3.0))
    cases.append(("hydrogen-rich", aa.Structure.from_records(rec).soa("/"), ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False).atomic_contacts()))
    rec = synth.gen_s2(40000)  # one-atom residues, ordinals 2 i: nothing to reject
    cases.append(("s2 40000", aa.Structure.from_records(rec, hierarchy=True).soa("/"), ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()))
    for what, soa, want in cases:
        for runs in (True, False, None):
            for only in (False, True):
                got = ctx.atomic_contacts(soa, aa.default_params(contacts_only=only, residue_runs=runs))
                assert_pairs_equal(got, want[want["kind"] != 0] if only else want, f"{what} residue_runs={runs} only={only}")
    # the word's limits: ordinals up to 2^20 - 3 fit (and the last residue of a chain is still 3 words from the first of the next); one more and
    # k_place flags the input, which switches the early exit off -- the lists stay the same either way
    what, soa, want = cases[0]
    top = int(soa["res_ord"].max())
    for shift in ((1 << 20) - 3 - top, (1 << 20) - 2 - top, 1 << 30):
        moved = dict(soa); moved["res_ord"] = (soa["res_ord"] + np.uint32(shift)).astype(np.uint32)
        assert_pairs_equal(ctx.atomic_contacts(moved, aa.default_params(residue_runs=True)), want, f"{what} ordinals + {shift}")
    for first in (2047 - int(soa["chain_rank"].max()), 2048, 1 << 31):  # chain ranks: 11 bits fit
        moved = dict(soa); moved["chain_rank"] = (soa["chain_rank"] + np.uint32(max(first, 0))).astype(np.uint32)
        assert_pairs_equal(ctx.atomic_contacts(moved, aa.default_params(residue_runs=True)), want, f"{what} chain ranks + {first}")
    # the memo: a fresh context runs the plain kernels first, samples the input in k_place, and picks the residue-rule kernels from the second call on
    fresh = aa.Context(0)
    for k in range(3):
        assert_pairs_equal(fresh.atomic_contacts(soa), want, f"{what} memo call {k}")


@pytest.mark.parametrize("rows", [2, 8, 32])
def test_cell_rows_in_y_strips_emit_the_same_lists(rows):
    """Round 5: above ~2.5 x 10^6 atoms the cell rows of a single-model input are ordered in y strips (arp_internal.h grid_row: the emit kernel's
    gathers then find the next layer's rows in the L2).  arp_debug_set("strip_rows", N) forces strips of N rows on inputs of any size, so the order is
    checked here against the oracle where the oracle finishes: every emit kernel family (hole-free 4-wave, 12-wave with and without the task
    split, residue-rule, staged), the alternative single-pass kernel's twin k_pairs through the ordered two-pass emitter, hydrogens (deferred
    probes address records by slot), the device table's ring-atom search (table_dev.hip walks the rows itself) and the SAP neighbour sum."""
    aa.debug_set("strip_rows", rows)  # (for parameter blocks built from now on: the fresh context below)
    try:
        ctx = aa.Context(0)
        for gen, n in (("gen_s2", 6000), ("gen_s1", 30000), ("gen_s2", 100000), ("gen_s1", 330000)):
            rec = getattr(synth, gen)(n)
            soa = aa.Structure.from_records(rec, hierarchy=True).soa("/")
            want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()
            for k in range(2):  # (the second call: residue-rule kernels and the staged hole-free sequence, by the context's memos)
                assert_pairs_equal(ctx.atomic_contacts(soa), want, f"{gen} {n} strips of {rows}, call {k}")
            assert_pairs_equal(ctx.atomic_contacts(soa, aa.default_params(contacts_only=True)), want[want["kind"] != 0], f"{gen} {n} strips of {rows}, contacts only")
            if n <= 30000:
                got = ctx.atomic_contacts(soa, aa.default_params(deterministic=True))
                assert_pairs_equal(got, want, f"{gen} {n} strips of {rows}, ordered")
        rec = synth.gen_stress(n_res=2600, seed=5)
        want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False).atomic_contacts()
        assert_pairs_equal(ctx.atomic_contacts(aa.Structure.from_records(rec).soa("/")), want, f"hydrogen-rich, strips of {rows}")
        for name, n_rows in (("1ubq", 532), ("6bft", 7236)):
            table = ctx.get_contacts(aa.load_model(str(synth.DATA / f"{name}.pdb")), "/", 0.1, 6.5)
            assert len(table["model"]) == n_rows
            _lines_close(_table_lines(table), (GOLDEN / f"{name}_contacts.csv").read_text().splitlines()[1:])
        rec = synth.gen_s1(40000)
        rng = np.random.default_rng(3)
        side = ~np.isin(rec["name"], [b"N", b"CA", b"C", b"O", b"OXT"])
        w = rng.uniform(0.0, 1.0, len(side)).astype(np.float32)
        got = aa.sap_neighbor_sum(ctx, rec["x"], rec["y"], rec["z"], side, w, 5.0)
        want_sap = ob.sap_neighbor_sum(rec["x"], rec["y"], rec["z"], side, w, 5.0)
        assert np.abs(got - want_sap).max() <= 2e-5 * max(1.0, float(np.abs(want_sap).max()))
        if rows == 8:  # the alternative single-pass kernel (k_pairs<kEmit>, the route of inputs beyond 2^24 slots) on strips, in a process of its own
            import subprocess
            import sys

            r = subprocess.run([sys.executable, str(synth.DATA.parent / "emit_kernel_check.py"), "gather"], capture_output=True, text=True, timeout=300,
                               env=dict(os.environ, ARP_TEST_STRIP_ROWS="8"))
            assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]
    finally:
        aa.debug_set("strip_rows", int(os.environ.get("ARP_TEST_STRIP_ROWS", "0")))  # (conftest.py: the whole suite may be running on strips)


# ---------------------------------------------------------------------------------------------- edge cases
def _mini(xyz, names=None, resn=None, elems=None, chains=None, resi=None):
    n = len(xyz)
    xyz = np.asarray(xyz, dtype=np.float64).reshape(n, 3)
    return {
        "x": xyz[:, 0].copy(), "y": xyz[:, 1].copy(), "z": xyz[:, 2].copy(), "occupancy": np.ones(n),
        "serial": np.arange(1, n + 1, dtype=np.int32), "resi": np.asarray(resi if resi is not None else 10 * np.arange(1, n + 1), dtype=np.int32),
        "model_serial": np.zeros(n, dtype=np.int32),
        "name": np.asarray(names if names is not None else [b"CA"] * n, dtype="S8"), "resn": np.asarray(resn if resn is not None else [b"ALA"] * n, dtype="S8"),
        "chain": np.asarray(chains if chains is not None else [b"A"] * n, dtype="S8"), "altloc": np.zeros(n, dtype="S4"), "icode": np.zeros(n, dtype="S4"),
        "element": np.asarray(elems if elems is not None else [b"C"] * n, dtype="S4"),
    }


def _both_from(rec):
    return aa.Structure.from_records(rec), ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)


def test_empty_and_tiny_inputs(ctx):
    zero = {k: np.zeros(0, dtype=d) for k, d in (("x", "f8"), ("y", "f8"), ("z", "f8"), ("attr", "u4"), ("res_ord", "u4"), ("chain_rank", "u4"), ("model", "u4"))}
    assert len(ctx.atomic_contacts(zero)) == 0
    prod, orc = _both_from(_mini([[1, 2, 3]]))
    got, want = run_both(ctx, prod, orc)
    assert len(got) == len(want) == 0
    # residues 0 and 1 of one chain are sequence neighbours; 0 and 2 are not (complex.rs:113)
    prod, orc = _both_from(_mini([[0, 0, 0], [3, 0, 0], [0, 3, 0]]))
    got, want = run_both(ctx, prod, orc)
    assert_pairs_equal(got, want)
    assert len(got) == 1 and (got["i"][0], got["j"][0]) == (0, 2)


def test_cutoff_is_inclusive(ctx):
    # rstar locate_within_distance: d^2 <= r^2 (complex.rs:204)
    c = 6.5
    xyz = [[0, 0, 0], [50, 0, 0], [c, 0, 0], [50, c, 0], [100, 0, 0], [100, 0, np.nextafter(c, 10)]]
    prod, orc = _both_from(_mini(xyz, chains=[b"A", b"A", b"B", b"B", b"A", b"B"]))
    got, want = run_both(ctx, prod, orc)
    assert_pairs_equal(got, want)
    assert sorted(zip(got["i"].tolist(), got["j"].tolist())) == [(0, 2), (1, 3)]


def test_all_hydrogen_and_coincident_atoms(ctx):
    prod, orc = _both_from(_mini([[0, 0, 0], [1, 0, 0], [0, 1, 0]], names=[b"H1", b"H2", b"H3"], elems=[b"H"] * 3))
    got, want = run_both(ctx, prod, orc)
    assert len(got) == len(want) == 0
    # 300 atoms on one point -> one cell with 300 members, every non-adjacent pair is a StericClash
    prod, orc = _both_from(_mini(np.full((300, 3), 7.25)))
    got, want = run_both(ctx, prod, orc)
    assert_pairs_equal(got, want)
    assert len(got) == 299 * 298 // 2 and (got["kind"] == 1).all()
    # 560 atoms on one point: 156 k pairs from nine tasks -- more than the context's first guess for a buffer (65 536 records), so the hole-free
    # sequence of small inputs has to report the size (the host derives the overflow from the record counter) and the call is repeated
    prod, orc = _both_from(_mini(np.full((560, 3), -3.5)))
    got, want = run_both(aa.Context(0), prod, orc)
    assert_pairs_equal(got, want)
    assert len(got) == 559 * 558 // 2 > 65536


def test_sparse_huge_extent_and_large_coordinates(ctx):
    rng = np.random.default_rng(5)
    blobs = np.concatenate([rng.uniform(0, 12, size=(150, 3)) + o for o in ([0, 0, 0], [9000, -9000, 4000], [-9999, 9999, -9999], [5000, 5000, 5000])])
    prod, orc = _both_from(_mini(np.round(blobs, 3)))
    got, want = run_both(ctx, prod, orc)
    assert len(want) > 1000
    assert_pairs_equal(got, want)
    # 30 000 atoms in 200 far-apart blobs: the grid takes all the cells the workspace has (a few hundred thousand), several times what the
    # one-block cell scan holds in its registers per pass (grid.inl k_scan_one: 65 536) -- the launcher picked it by the atom count
    offs = rng.uniform(-5000, 5000, size=(200, 3))
    blobs = np.concatenate([rng.uniform(0, 14, size=(150, 3)) + o for o in offs])
    prod, orc = _both_from(_mini(np.round(blobs, 3)))
    got, want = run_both(ctx, prod, orc)
    assert len(want) > 100000
    assert_pairs_equal(got, want)


def test_bad_inputs_are_errors(ctx):
    soa = aa.load_model(str(synth.DATA / "1ubq.pdb")).soa("/")
    soa["x"][17] = np.nan
    with pytest.raises(aa.ArpeggiaError) as e:
        ctx.atomic_contacts(soa)
    assert e.value.status == _lib.ARP_ERR_BAD_INPUT


# ---------------------------------------------------------------------------------------------- size-independent properties
def test_full_size_properties_1e6(ctx):
    n = 1_000_000
    rec = synth.gen_s2(n)
    prod = aa.Structure.from_records(rec, hierarchy=True)
    soa = prod.soa("/")
    a = ctx.atomic_contacts(soa)
    assert len(a) > 25 * n
    # (1) the ordered emitter is deterministic byte for byte; the default single-pass emitter returns the same set
    det = aa.default_params(deterministic=True)
    b = ctx.atomic_contacts(soa, det)
    assert np.array_equal(b, ctx.atomic_contacts(soa, det))
    assert np.array_equal(canon(a), canon(b))
    # (2) permutation invariance: shuffling the atoms changes indices only
    perm = np.random.default_rng(1).permutation(n)
    inv = np.empty(n, dtype=np.uint32); inv[perm] = np.arange(n, dtype=np.uint32)
    shuffled = {k: (v[perm] if len(v) == n else v) for k, v in soa.items()}
    shuffled["res_id"] = np.arange(n, dtype=np.uint32)  # one residue per atom; tables are per residue
    for k in ("res_cb", "res_sg"):
        shuffled[k] = np.full(n, 0xFFFFFFFF, dtype=np.uint32)
    shuffled["res_h_ptr"] = np.zeros(n + 1, dtype=np.uint32)
    c = ctx.atomic_contacts(shuffled)
    c2 = c.copy(); c2["i"] = perm[c["i"]]; c2["j"] = perm[c["j"]]
    assert np.array_equal(canon(c2), canon(a))
    # (3) a smaller cutoff yields the subset with identical flags (every rule threshold is <= 4.5 A)
    d = ctx.atomic_contacts(soa, aa.default_params(0.1, 5.0))
    ca = canon(a); cd = canon(d)
    keep = ca["dist"].astype(np.float64) <= 5.0 - 1e-4
    key = lambda p: p["i"].astype(np.uint64) << np.uint64(32) | p["j"].astype(np.uint64)
    sel = np.isin(key(ca), key(cd))
    assert (sel | ~keep).all()
    assert np.array_equal(ca[sel], cd)
    # (4) every flag bit implies its distance bound; candidates are within the cutoff
    dist = a["dist"].astype(np.float64)
    assert dist.max() <= 6.5 + 1e-5
    for name, lim in (("HydrophobicContact", 4.5), ("IonicBond", 4.0), ("IonicRepulsion", 4.0), ("PolarContact", 3.5), ("WeakPolarContact", 3.5)):
        m = (a["kind"] >> ob.INTERACTIONS.index(name)) & 1 == 1
        assert dist[m].max(initial=0.0) <= lim + 1e-5
    # (5) against the oracle at full size (seconds on one core)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    assert_pairs_equal(a, orc.atomic_contacts(), "s2 1e6")


def test_automatic_y_strips_at_full_size_match_the_oracle():
    """A size at which grid_setup switches the cell rows to y strips BY ITSELF (grid.inl: more than 36 864 atoms per layer of a single-model
    input): 3.3 x 10^6 S2 atoms -> ~9.4 x 10^7 records, against the oracle's list through the record count and the order-independent 64-bit hash
    the bench line uses (bench.record_hash_numpy: sorting 10^8 records twice would take minutes), and against the same call in layer order
    (strip_rows = 1) -- which must emit the same set."""
    import bench

    n = 3_300_000
    rec = synth.gen_s2(n)
    ext = [float(rec[k].max() - rec[k].min()) for k in ("x", "y", "z")]
    nz = int(ext[2] / (6.5 * (1.0 + 1e-6))) + 1
    assert n / nz > 36864, "this input no longer selects the strips: grid.inl kStripLayerAtoms"
    soa = aa.Structure.from_records(rec, hierarchy=True).soa("/")
    got = aa.Context(0).atomic_contacts(soa)
    h_got = bench.record_hash_numpy(got["i"], got["j"], got["dist"], got["kind"])
    want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()
    assert len(got) == len(want) > 25 * n
    assert h_got == bench.record_hash_numpy(want["i"], want["j"], want["dist"].astype(np.float32), want["kind"])
    del want
    aa.debug_set("strip_rows", 1)
    try:
        flat = aa.Context(0).atomic_contacts(soa)
    finally:
        aa.debug_set("strip_rows", int(os.environ.get("ARP_TEST_STRIP_ROWS", "0")))
    assert len(flat) == len(got) and bench.record_hash_numpy(flat["i"], flat["j"], flat["dist"], flat["kind"]) == h_got


def test_s1_cloud_1e6_vs_oracle(ctx):
    """The other half of BASELINE config 4 ("the headline run reports both"): the chemistry-faithful S1 cloud at 10^6 atoms, every pair against
    the oracle -- indices and flags bit-exact, f32 distances bit-identical -- for both emitters and for the contacts-only filter."""
    rec = synth.gen_s1(1_000_000, seed=0xA11CE5EED00 + 3)
    prod = aa.Structure.from_records(rec, hierarchy=True)
    soa = prod.soa("/")
    want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()
    assert len(want) > 15_000_000
    got = ctx.atomic_contacts(soa, aa.default_params(residue_runs=False))
    assert_pairs_equal(got, want, "s1 1e6")
    got = ctx.atomic_contacts(soa, aa.default_params(residue_runs=True))  # (round 5: the residue rule applied before the exact phase)
    assert_pairs_equal(got, want, "s1 1e6, residue-rule kernels")
    only = ctx.atomic_contacts(soa, aa.default_params(contacts_only=True))
    assert_pairs_equal(only, want[want["kind"] != 0], "s1 1e6 contacts only")
    del got, only
    assert_pairs_equal(ctx.atomic_contacts(soa, aa.default_params(deterministic=True)), want, "s1 1e6 ordered")


# ---------------------------------------------------------------------------------------------- enqueue / batch forms
def test_enqueue_with_resident_buffers_and_capacity_error(ctx):
    torch = pytest.importorskip("torch")
    prod = aa.load_model(str(synth.DATA / "6bft.pdb"))
    soa = prod.soa("/")
    want = ctx.atomic_contacts(soa)
    dev = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa.items()}
    keep = []
    atoms = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    prm = aa.default_params()
    c2 = aa.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    out = torch.empty((len(want), 4), dtype=torch.int32, device="cuda")
    for _ in range(3):
        c2.enqueue(atoms, prm, out.data_ptr(), len(want))
        assert c2.result() == len(want)
    got = out.cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1)
    assert np.array_equal(canon(got), canon(want))
    det = aa.default_params(deterministic=True)
    c2.enqueue(atoms, det, out.data_ptr(), len(want))
    assert c2.result() == len(want)
    got2 = out.cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1)
    c2.enqueue(atoms, det, out.data_ptr(), len(want))
    assert c2.result() == len(want)
    assert np.array_equal(got2, out.cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1)) and np.array_equal(canon(got2), canon(want))
    small = torch.zeros((1000, 4), dtype=torch.int32, device="cuda")
    guard = small.clone()
    c2.enqueue(atoms, prm, small.data_ptr(), 900)
    with pytest.raises(aa.ArpeggiaError) as e:
        c2.result()
    assert e.value.status == _lib.ARP_ERR_CAPACITY and str(len(want)) in str(e.value)
    assert torch.equal(small[900:], guard[900:])  # nothing written past the capacity
    prof = None
    c2.profile(True)
    c2.enqueue(atoms, prm, out.data_ptr(), len(want)); c2.result()
    prof = c2.profile_read()
    assert "pairs_emit" in prof and all(v >= 0 for v in prof.values())
    # 6bft runs the hole-free sequence of small inputs; the same two checks on the chunked sequence with its fix-up (30 000 atoms): a buffer of
    # exactly P records suffices (the holes' overflow lives in the engine's scratch until they are closed), and a short one reports P
    soa_m = aa.Structure.from_records(synth.gen_s2(30000, seed=3), hierarchy=True).soa("/")
    want_m = ctx.atomic_contacts(soa_m)
    dev_m = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa_m.items()}
    atoms_m = aa.atoms_from_arrays(dev_m, location=_lib.ARP_MEM_DEVICE, keep=keep)
    out_m = torch.empty((len(want_m), 4), dtype=torch.int32, device="cuda")
    c2.enqueue(atoms_m, prm, out_m.data_ptr(), len(want_m))
    assert c2.result() == len(want_m)
    assert np.array_equal(canon(out_m.cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1)), canon(want_m))
    c2.enqueue(atoms_m, prm, small.data_ptr(), 900)
    with pytest.raises(aa.ArpeggiaError) as e:
        c2.result()
    assert e.value.status == _lib.ARP_ERR_CAPACITY and str(len(want_m)) in str(e.value)
    assert torch.equal(small[900:], guard[900:])


def test_batch_of_structures(ctx):
    recs = [synth.gen_s1(1500 + 400 * k, seed=100 + k) for k in range(5)]
    structs = [aa.Structure.from_records(r, hierarchy=True) for r in recs]
    views = [s.view("/") for s in structs]
    singles = [ctx.atomic_contacts(v) for v in views]
    arr = (C.POINTER(_lib.arp_atoms) * len(views))(*[C.pointer(v) for v in views])
    outs = (_lib.arp_pairs * len(views))()
    ctxs = (C.c_void_p * 1)(ctx._h)
    prm = aa.default_params()
    st = _lib.lib.arp_contacts_atomic_batch(ctxs, 1, arr, len(views), C.byref(prm), outs)
    assert st == 0, _lib.lib.arp_last_error()
    for k in range(len(views)):
        buf = (C.c_char * (outs[k].n * 16)).from_address(outs[k].data)
        got = np.frombuffer(buf, dtype=aa.PAIR_DTYPE).copy()
        _lib.lib.arp_pairs_free(C.byref(outs[k]))
        assert np.array_equal(canon(got), canon(singles[k]))
    # the members' lists were views into pooled pinned blocks: all freed now, so the pool holds idle blocks that can be given back
    assert _lib.lib.arp_release_host_pool() > 0 and _lib.lib.arp_release_host_pool() == 0
    again = aa.atomic_contacts_batch([ctx], views, prm)
    assert all(np.array_equal(canon(again[k]), canon(singles[k])) for k in range(len(views)))


def test_packed_batch_equals_single_calls(ctx):
    """Packing renumbers models and offsets residue tables; the split result must equal every single call, and the
    single calls are oracle-checked (6bft, 1ubq, stress clouds with hydrogens, a two-model NMR-like structure)."""
    structs = [aa.load_model(str(synth.DATA / "1ubq.pdb")), aa.load_model(str(synth.DATA / "6bft.pdb"))]
    structs += [aa.Structure.from_records(synth.gen_stress(n_res=120 + 40 * k, seed=60 + k)) for k in range(4)]
    two = synth.gen_stress(n_res=90, seed=77, n_models=2)
    structs.append(aa.Structure.from_records(two))
    far = synth.gen_stress(n_res=80, seed=78)
    far["x"] += 5.0e4  # a member that would blow up the shared grid starts its own pack
    structs.append(aa.Structure.from_records(far))
    empty = {k: v[:0] for k, v in two.items()}
    structs.append(aa.Structure.from_records(empty))
    views = [s.view("/") for s in structs]
    singles = [ctx.atomic_contacts(v) for v in views]
    assert len(singles[-1]) == 0 and len(singles[0]) == 9128
    # packs are formed for contacts-only lists (full lists are copy-bound and go one structure at a time)
    for det in (False, True):
        got = aa.atomic_contacts_batch([ctx], views, aa.default_params(deterministic=det))
        packed = aa.atomic_contacts_batch([ctx], views, aa.default_params(deterministic=det, contacts_only=True))
        assert len(got) == len(packed) == len(views)
        for k in range(len(views)):
            assert np.array_equal(canon(got[k]), canon(singles[k])), k
            assert np.array_equal(canon(packed[k]), canon(singles[k][singles[k]["kind"] != 0])), k
    # the oracle on one packed member, to pin the whole chain
    want = ob.Structure.load(str(synth.DATA / "6bft.pdb")).atomic_contacts()
    assert_pairs_equal(got[1], want, "6bft from a batch")
    assert_pairs_equal(packed[1], want[want["kind"] != 0], "6bft from a pack")


@pytest.mark.parametrize("source", ["6bft", "stress", "s1"])
def test_contacts_only_is_the_kind_filter_of_the_full_list(ctx, source):
    """ARP_FLAG_CONTACTS_ONLY drops kind == 0 candidates on the device (the rows get_atomic_contacts never makes,
    complex.rs:208-297); both emitters, the size query and the packed batch agree with filtering the oracle's list."""
    if source == "6bft":
        path = str(synth.DATA / "6bft.pdb")
        prod, orc = aa.load_model(path), ob.Structure.load(path)
    elif source == "stress":
        rec = synth.gen_stress(n_res=400, seed=7)
        prod = aa.Structure.from_records(rec)
        orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
    else:
        rec = synth.gen_s1(60000)
        prod = aa.Structure.from_records(rec, hierarchy=True)
        orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    want = orc.atomic_contacts()
    want = want[want["kind"] != 0]
    assert 0 < len(want)
    view = prod.view("/")
    for det in (False, True):
        prm = aa.default_params(deterministic=det, contacts_only=True)
        got = ctx.atomic_contacts(view, prm)
        assert_pairs_equal(got, want, f"{source} contacts-only det={det}")
    assert_pairs_equal(aa.atomic_contacts_batch([ctx], [view, view], aa.default_params(contacts_only=True))[1], want, "packed contacts-only")
    torch = pytest.importorskip("torch")
    soa = prod.soa("/")
    dev = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa.items()}
    keep = []
    atoms = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    c2 = aa.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    for det in (False, True):
        prm = aa.default_params(deterministic=det, contacts_only=True)
        assert c2.count(atoms, prm) == len(want)
        out = torch.empty((len(want), 4), dtype=torch.int32, device="cuda")
        c2.enqueue(atoms, prm, out.data_ptr(), len(want))
        assert c2.result() == len(want)
        assert_pairs_equal(out.cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1), want, f"{source} resident contacts-only det={det}")


def test_packed_batch_of_thousands_of_tiny_structures(ctx):
    """45 000 members of 3-9 residues each (packs of ~20 000: the 2^20-atom limit): a few dozen records per member, so that one wave's range of
    a pack's joint list (a couple of hundred records) mentions more members than its table holds (batch.inl kSplitSlots) -- the per-64-records
    part of the split -- and many members come back empty.  Every member against its single call, both modes; the single calls of the
    distinct members against the oracle."""
    base = [synth.gen_stress(n_res=3 + k % 7, seed=300 + k, hydrogens=bool(k % 2), n_chains=1 + k % 2) for k in range(40)]
    structs = [aa.Structure.from_records(r) for r in base]
    order = np.random.default_rng(3).integers(0, len(structs), 45000)
    views = [structs[k].view("/") for k in order]
    for only in (False, True):
        prm = aa.default_params(contacts_only=only)
        singles = [ctx.atomic_contacts(structs[k].view("/"), prm) for k in range(len(structs))]
        for k in range(len(structs)):
            want = ob.Structure.from_atoms(synth.records_to_oracle(base[k], flat=False), flat=False).atomic_contacts()
            assert_pairs_equal(singles[k], want[want["kind"] != 0] if only else want, f"tiny member {k} only={only}")
        got = aa.atomic_contacts_batch([ctx], views, prm)
        assert len(got) == len(views)
        assert sum(len(g) for g in got) == sum(len(singles[k]) for k in order)
        for j, k in enumerate(order):
            assert np.array_equal(canon(got[j]), canon(singles[k])), (j, int(k), only)


def test_packed_batch_reports_the_failing_structure(ctx):
    ok = synth.gen_stress(n_res=60, seed=21, hydrogens=False)
    bad = {k: v[:4].copy() for k, v in ok.items()}
    bad["name"][:] = [b"SG", b"CA", b"SG", b"CA"]
    bad["resn"][:] = b"CYS"
    bad["element"][:] = [b"S", b"C", b"S", b"C"]
    bad["chain"][:] = [b"Y", b"Y", b"Z", b"Z"]
    bad["resi"][:] = [1, 1, 5, 5]
    bad["x"][:] = [0.0, 1.5, 2.05, 3.5]; bad["y"][:] = 0.0; bad["z"][:] = 0.0
    structs = [aa.Structure.from_records(r) for r in (ok, bad, ok)]
    for only in (True, False):  # packed (contacts-only) and one-by-one
        prm = aa.default_params(contacts_only=only)
        with pytest.raises(aa.ArpeggiaError) as e:
            aa.atomic_contacts_batch([ctx], [s.view("/") for s in structs], prm)
        assert e.value.status == _lib.ARP_ERR_BAD_INPUT and "CB" in str(e.value)
        assert len(aa.atomic_contacts_batch([ctx], [structs[0].view("/"), structs[2].view("/")], prm)) == 2


def test_packed_batch_with_model_ordinals_beyond_the_pack_tables(ctx):
    """API v2's 32-bit model ordinals can exceed the 65 536 per-model entries of a pack's tables, and the host only learns of it after
    the stream has run.  A member with a sparse ordinal (70 000, 2^32 - 1) must leave every kernel of the pack in bounds (k_pack_fix gives an
    overfull pack's atoms model 0; the tables' indices are clamped), the call must report THAT member's input error, and the others' lists must
    still be right afterwards."""
    recs = [synth.gen_s1(1200 + 300 * k, seed=400 + k) for k in range(4)]
    soas = [aa.Structure.from_records(r, hierarchy=True).soa("/") for r in recs]
    singles = [ctx.atomic_contacts(s, aa.default_params(contacts_only=True)) for s in soas]
    ctx = aa.Context(0)  # a fresh, small workspace: whether a sparse ordinal is an error depends on the cells the workspace holds (70 001 slabs fit a large one)
    for sparse in (70000, 0xFFFFFFFF):
        bad = dict(soas[1]); bad["model"] = soas[1]["model"].copy(); bad["model"][len(bad["model"]) // 2:] = sparse
        with pytest.raises(aa.ArpeggiaError) as e:
            aa.atomic_contacts_batch([ctx], [soas[0], bad, soas[2], soas[3]], aa.default_params(contacts_only=True))
        assert e.value.status == _lib.ARP_ERR_BAD_INPUT and "model" in str(e.value)
        got = aa.atomic_contacts_batch([ctx], soas, aa.default_params(contacts_only=True))
        assert all(np.array_equal(canon(g), canon(w)) for g, w in zip(got, singles))
    # many members whose models add up past the tables: 40 members x 2000 models each (dense, legal one by one) -> the pack is overfull, the
    # members run one at a time, every list is right
    many = []
    for k in range(40):
        m = dict(soas[k % 4]); n = len(m["x"])
        m["model"] = (np.arange(n, dtype=np.uint32) * np.uint32(2000) // np.uint32(n)).astype(np.uint32)
        many.append(m)
    want = [ctx.atomic_contacts(m, aa.default_params(contacts_only=True)) for m in many[:4]]
    got = aa.atomic_contacts_batch([ctx], many, aa.default_params(contacts_only=True))
    assert all(np.array_equal(canon(got[k]), canon(want[k % 4])) for k in range(40))


@pytest.mark.parametrize("kernel", ["gather"])
def test_alternative_emit_kernels(kernel):
    """The one alternative single-pass emit kernel (k_pairs<kEmit>: the route of inputs beyond 2^24 slots; arp_debug_set("emit_kernel", 1), a
    process-wide switch) stays parity-green, in a process of its own."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, str(synth.DATA.parent / "emit_kernel_check.py"), kernel], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------- BASELINE config 5: a batch of ~5k-atom structures
def _config5_members(n_members, seed=5):
    """atoms ~ N(5000, 500^2) clipped to [3000, 7000], each structure built by S1 from the 1ubq template."""
    rng = np.random.default_rng(seed)
    sizes = np.clip(np.rint(rng.normal(5000.0, 500.0, n_members)), 3000, 7000).astype(int)
    recs = [synth.gen_s1(int(n), seed=0xA11CE5EED00 + 5 + 17 * k) for k, n in enumerate(sizes)]
    for k, r in enumerate(recs):  # unrelated files sit anywhere in space: every member of a pack gets its own grid origin
        r["x"] = np.round(r["x"] + 1000.0 * (k % 7), 3); r["y"] = np.round(r["y"] - 731.0 * (k % 5), 3); r["z"] = np.round(r["z"] + 97.0 * (k % 3), 3)
    return recs


def test_config5_batch_every_member_matches_the_oracle(ctx):
    recs = _config5_members(72)
    assert 3000 <= min(len(r["x"]) for r in recs) and max(len(r["x"]) for r in recs) <= 7000
    structs = [aa.Structure.from_records(r, hierarchy=True) for r in recs]
    views = [s.view("/") for s in structs]
    wants = [ob.Structure.from_atoms(synth.records_to_oracle(r, flat=True), flat=True).atomic_contacts() for r in recs]
    assert all(len(w) > 30000 for w in wants)
    ctx2 = aa.Context(0)  # a second context on the same device stands in for a second GPU: the longest-first deal over devices
    for det in (False, True):
        for only in (False, True):
            prm = aa.default_params(deterministic=det, contacts_only=only, residue_runs=None if det else only)  # (the pack through both kernel families)
            for contexts in ([ctx], [ctx, ctx2]):
                got = aa.atomic_contacts_batch(contexts, views, prm)          # packed: shared launches, split on the device
                assert len(got) == len(views)
                for k, w in enumerate(wants):
                    assert_pairs_equal(got[k], w[w["kind"] != 0] if only else w, f"member {k} det={det} only={only} ctxs={len(contexts)}")
            if det:  # the ordered emitter is byte-identical run to run, through a pack too
                one, two = aa.atomic_contacts_batch([ctx], views, prm), aa.atomic_contacts_batch([ctx], views, prm)
                assert all(np.array_equal(a, b) for a, b in zip(one, two))
    # unpacked: the same members one call at a time
    for k in (0, 17, 71):
        for det in (False, True):
            assert_pairs_equal(ctx.atomic_contacts(views[k], aa.default_params(deterministic=det)), wants[k], f"single call member {k}")


def test_lds_reads_past_the_allocation_return_zero():
    """The hardware contract behind k_emit's prefilter runs and the SAP sum: a lane past its window end keeps reading 16-byte
    records at immediate offsets -- at worst beyond the block's LDS allocation -- and only its result bits are dropped.  What that needs is that
    an LDS read past the allocation never faults.  tests/lds_oob/lds_oob.hip reads from just past a 4 KB allocation up to 2 GB beyond it: no fault,
    the fill pattern in range, 0 from 64 KB past the end on.  (Just past the end the reads return stale words of the allocation granule's
    padding, not 0 -- round 5 measured it; the kernels' comments said "reads as zero" until then.  Nothing depends on the value.)"""
    import subprocess

    src = synth.DATA.parent / "lds_oob" / "lds_oob.hip"
    so = src.with_name("liblds_oob.so")
    if not so.exists():
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "--offload-arch=gfx950", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    lib.lds_oob_check.restype = C.c_int
    assert lib.lds_oob_check(1) == 0


# ---------------------------------------------------------------------------------------------- robustness (round-1 review)
def test_sparse_model_ids_are_an_input_error_not_a_hang():
    # a model ordinal the workspace cannot hold (65535 on a ten-atom input: two grid layers per model, 73k cells in a fresh
    # context) used to spin the device-side grid sizing forever; a context whose workspace has grown can hold it and just works
    c = aa.Context(0)
    soa = aa.Structure.from_records(_mini(np.arange(30, dtype=np.float64).reshape(10, 3))).soa("/")
    dense = c.atomic_contacts(soa)
    soa["model"][:] = 65535
    for det in (False, True):
        with pytest.raises(aa.ArpeggiaError) as e:
            c.atomic_contacts(soa, aa.default_params(deterministic=det))
        assert e.value.status == _lib.ARP_ERR_BAD_INPUT and "model" in str(e.value)
    soa["model"][:] = 0
    assert np.array_equal(canon(c.atomic_contacts(soa)), canon(dense))  # the context is still usable


def test_negative_cutoff_searches_the_same_sphere(ctx):
    # the reference only ever uses dist_cutoff^2 (complex.rs:191,303)
    prod = aa.load_model(str(synth.DATA / "6bft.pdb"))
    plus = ctx.atomic_contacts(prod.view("/"), aa.default_params(0.1, 6.5))
    for det in (False, True):
        minus = ctx.atomic_contacts(prod.view("/"), aa.default_params(0.1, -6.5, deterministic=det))
        assert np.array_equal(canon(minus), canon(plus))
    assert len(ctx.get_contacts(prod, "/", 0.1, -6.5)["model"]) == len(ctx.get_contacts(prod, "/", 0.1, 6.5)["model"]) == 7236


def test_ordered_paths_on_a_fresh_context_and_after_a_workspace_regrow():
    """Round 1 recorded one GPU fault on a work-in-progress build: the first kernel of the first (ordered) call wrote through a
    workspace member that was not allocated yet.  The workspace now checks its members on the host; this runs the ordered and the
    contacts-only ordered path as the very first calls of a context and again right after the workspace has been regrown."""
    small, big = aa.load_model(str(synth.DATA / "1ubq.pdb")), aa.load_model(str(synth.DATA / "6bft.pdb"))
    want_small = ob.Structure.load(str(synth.DATA / "1ubq.pdb")).atomic_contacts()
    want_big = ob.Structure.load(str(synth.DATA / "6bft.pdb")).atomic_contacts()
    for first_only in (False, True):
        c = aa.Context(0)
        prm = aa.default_params(deterministic=True, contacts_only=first_only)
        assert_pairs_equal(c.atomic_contacts(small.view("/"), prm), want_small[want_small["kind"] != 0] if first_only else want_small, "first call")
        for only in (True, False):  # 6bft needs a larger workspace than 1ubq: every buffer is reallocated
            prm = aa.default_params(deterministic=True, contacts_only=only)
            assert_pairs_equal(c.atomic_contacts(big.view("/"), prm), want_big[want_big["kind"] != 0] if only else want_big, "after regrow")
        assert_pairs_equal(c.atomic_contacts(big.view("/")), want_big, "single pass after regrow")


@pytest.mark.parametrize("n_res", [300, 2600])  # (2600 residues with hydrogens: ~32 k atoms, the range of the scratch-staged hole-free sequence of round 5)
def test_deferred_pass_memo_survives_new_content_in_the_same_buffers(n_res):
    """The engine skips the launch of the probe pass when the previous call on the same (x pointer, n) deferred nothing; the fix-up checks
    on the device that nothing was deferred THIS time and the host repeats the call with the pass otherwise.  Here the caller's device
    buffers keep their addresses while their CONTENT alternates between a cloud that defers nothing and a hydrogen-rich structure of the
    same size whose HydrogenBond / Disulfide kinds only the probe pass can decide -- every call must equal the oracle, whatever the memo
    guessed (enqueue/result and the one-call form; all candidates and contacts only)."""
    torch = pytest.importorskip("torch")
    rec_b = synth.gen_stress(n_res=n_res, seed=17, box=28.0 * (n_res / 400.0) ** (1.0 / 3.0))
    prod_b = aa.Structure.from_records(rec_b)
    soa_b = prod_b.soa("/")
    n = len(soa_b["x"])
    rec_a = synth.gen_s2(n, seed=5)
    prod_a = aa.Structure.from_records(rec_a, hierarchy=True)
    soa_a = prod_a.soa("/")
    assert len(soa_a["x"]) == n
    want_a = ob.Structure.from_atoms(synth.records_to_oracle(rec_a, flat=True), flat=True).atomic_contacts()
    want_b = ob.Structure.from_atoms(synth.records_to_oracle(rec_b, flat=False), flat=False).atomic_contacts()
    probe_kinds = (1 << aa.INTERACTIONS.index("HydrogenBond")) | (1 << aa.INTERACTIONS.index("Disulfide"))
    assert (want_b["kind"] & probe_kinds).any() and not (want_a["kind"] & probe_kinds).any()

    def dev_of(v):
        return torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda()

    per_atom = ("x", "y", "z", "attr", "res_ord", "chain_rank", "model", "res_id")
    shared = {k: torch.empty_like(dev_of(soa_b[k])) for k in per_atom}
    keep = []
    atoms, content = {}, {}
    for name, soa in (("a", soa_a), ("b", soa_b)):
        dev = dict(shared)
        dev.update({k: dev_of(v) for k, v in soa.items() if k not in per_atom})
        content[name] = {k: dev_of(soa[k]) for k in per_atom}
        atoms[name] = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    assert atoms["a"].x == atoms["b"].x and atoms["a"].n == atoms["b"].n
    want = {"a": want_a, "b": want_b}
    c = aa.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    out = torch.empty((max(len(want_a), len(want_b)), 4), dtype=torch.int32, device="cuda")
    for only in (False, True):
        prm = aa.default_params(contacts_only=only)
        for name in ("a", "a", "b", "b", "a", "b", "a", "a"):
            for k in per_atom:
                shared[k].copy_(content[name][k])
            torch.cuda.synchronize()
            w = want[name] if not only else want[name][want[name]["kind"] != 0]
            c.enqueue(atoms[name], prm, out.data_ptr(), out.shape[0])
            got_n = c.result()
            assert got_n == len(w), f"{name} only={only}: {got_n} pairs vs oracle {len(w)}"
            assert_pairs_equal(out[:got_n].cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1), w, f"memo {name} only={only} (enqueue)")
            assert_pairs_equal(c.atomic_contacts(atoms[name], prm), w, f"memo {name} only={only} (one call)")


@pytest.mark.parametrize("gen,n", [("s2", 30000), ("s1", 30000), ("s2", 100000), ("s1", 100000), ("s1", 131000)])
def test_staged_hole_free_sequence_on_resident_inputs(gen, n):
    """Round 5: between 20 480 and ~131 k atoms a REPEATED call on the same device arrays (the engine's memo: the previous call deferred nothing) runs
    k_emit<.., STAGE>: every wave stages its records in a region of its own in the scratch block, the block takes exact places at its end, and no
    fix-up kernel runs.  First call (chunks + fix-up) and the staged calls after it must all equal the oracle: all candidates and contacts only, plain
    and residue-rule kernels; a buffer that is too small must report the count it needs (the staged copy drops what does not fit, it never spills
    into the scratch block that holds the other waves' regions) and a retry with enough room must be right."""
    torch = pytest.importorskip("torch")
    rec = getattr(synth, f"gen_{gen}")(n)
    soa = aa.Structure.from_records(rec, hierarchy=True).soa("/")
    want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts()
    dev = {k: torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda() for k, v in soa.items()}
    keep = []
    atoms = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    c = aa.Context(0, stream=torch.cuda.current_stream().cuda_stream)
    out = torch.empty((len(want) + 64, 4), dtype=torch.int32, device="cuda")
    for only in (False, True):
        w = want[want["kind"] != 0] if only else want
        for runs in (False, True):
            prm = aa.default_params(contacts_only=only, residue_runs=runs)
            for call in range(3):
                out.zero_()
                c.enqueue(atoms, prm, out.data_ptr(), out.shape[0])
                got_n = c.result()
                assert got_n == len(w), f"{gen} {n} only={only} runs={runs} call {call}: {got_n} vs {len(w)}"
                assert_pairs_equal(out[:got_n].cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1), w, f"{gen} {n} only={only} runs={runs} call {call}")
            # too small a buffer on the staged route: the needed count comes back, nothing is written past the capacity
            small = len(w) // 2
            guard = out[small:small + 64].clone()
            c.enqueue(atoms, prm, out.data_ptr(), small)
            with pytest.raises(aa.ArpeggiaError) as e:
                c.result()
            assert e.value.status == _lib.ARP_ERR_CAPACITY and str(len(w)) in str(e.value)
            assert torch.equal(out[small:small + 64], guard)
            c.enqueue(atoms, prm, out.data_ptr(), out.shape[0])
            assert c.result() == len(w)
            assert_pairs_equal(out[:len(w)].cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1), w, f"{gen} {n} only={only} runs={runs} after a capacity error")


@pytest.mark.parametrize("n_res", [300, 1800])
def test_no_speculation_flag_makes_the_enqueued_list_final_on_the_stream(n_res):
    """include/arpeggia_amd.h, arp_contacts_atomic_enqueue: `out` is defined only after arp_contacts_atomic_result returns ARP_OK -- unless
    ARP_FLAG_NO_SPECULATION is set, which makes the enqueue launch the probe pass whatever the memo says.  The caller's buffers first hold a
    cloud that defers nothing (so the memo says "skip the probe pass"), then a hydrogen-rich structure of the same size at the same addresses.
    A device-side copy of `out`, ordered on the SAME stream between enqueue and result, must equal the oracle under the flag; without it the
    copy still shows the kind-0 placeholders of the records only the probe pass can decide (the speculation the header warns about).
    1800 residues (22 k atoms) run the chunked sequence with its probe pass; 300 (3.7 k atoms) the hole-free sequence of small inputs, whose
    probes run inside the emit kernel: nothing is speculated there and the copy is final with or without the flag."""
    torch = pytest.importorskip("torch")
    rec_b = synth.gen_stress(n_res=n_res, seed=17)
    soa_b = aa.Structure.from_records(rec_b).soa("/")
    n = len(soa_b["x"])
    rec_a = synth.gen_s2(n, seed=5)
    soa_a = aa.Structure.from_records(rec_a, hierarchy=True).soa("/")
    want_b = ob.Structure.from_atoms(synth.records_to_oracle(rec_b, flat=False), flat=False).atomic_contacts()
    probe_kinds = (1 << aa.INTERACTIONS.index("HydrogenBond")) | (1 << aa.INTERACTIONS.index("Disulfide"))
    assert (want_b["kind"] & probe_kinds).any()

    def dev_of(v):
        return torch.from_numpy(v.view(np.int32) if v.dtype == np.uint32 else v).cuda()

    per_atom = ("x", "y", "z", "attr", "res_ord", "chain_rank", "model", "res_id")
    shared = {k: torch.empty_like(dev_of(soa_b[k])) for k in per_atom}
    keep, atoms, content = [], {}, {}
    for name, soa in (("a", soa_a), ("b", soa_b)):
        dev = dict(shared)
        dev.update({k: dev_of(v) for k, v in soa.items() if k not in per_atom})
        content[name] = {k: dev_of(soa[k]) for k in per_atom}
        atoms[name] = aa.atoms_from_arrays(dev, location=_lib.ARP_MEM_DEVICE, keep=keep)
    stream = torch.cuda.current_stream()
    c = aa.Context(0, stream=stream.cuda_stream)
    out = torch.zeros((len(want_b) + 4096, 4), dtype=torch.int32, device="cuda")

    def fill(name):
        for k in per_atom:
            shared[k].copy_(content[name][k])
        torch.cuda.synchronize()

    for flag in (True, False):
        prm = aa.default_params(no_speculation=flag)
        fill("a")
        c.enqueue(atoms["a"], aa.default_params(), out.data_ptr(), out.shape[0])
        c.result()  # the memo now says: these arrays defer nothing
        fill("b")
        c.enqueue(atoms["b"], prm, out.data_ptr(), out.shape[0])
        snap = out.clone()  # ordered on the context's stream, BEFORE the result call
        got_n = c.result()
        assert got_n == len(want_b)
        assert_pairs_equal(out[:got_n].cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1), want_b, f"after result, flag={flag}")
        early = snap[:got_n].cpu().numpy().view(aa.PAIR_DTYPE).reshape(-1)
        if flag:
            assert_pairs_equal(early, want_b, "device-side copy taken between enqueue and result, ARP_FLAG_NO_SPECULATION")
        elif n >= 20480:  # the speculation is real: the copy holds placeholders where the final list holds probe-decided kinds
            e, w = canon(early), canon(want_b)
            assert np.array_equal(e["i"], w["i"].astype(np.uint32)) and np.array_equal(e["j"], w["j"].astype(np.uint32))
            assert ((w["kind"] & probe_kinds) != 0)[e["kind"] == 0].any() and not (e["kind"] & probe_kinds).any()
        else:
            assert_pairs_equal(early, want_b, "device-side copy taken between enqueue and result, small input, no flag")


def test_deferred_list_overflow_grows_and_repeats():
    # hydrogen-rich structure: thousands of candidates need a probe; a 65536-entry list (every block holds a partly used 512-entry chunk) overflows and is grown 4x per retry
    rec = synth.gen_stress(n_res=600, seed=91)
    prod = aa.Structure.from_records(rec)
    want = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False).atomic_contacts()
    aa.debug_set("defer_entries", 65536)  # (the workspaces allocated from now on: the fresh context below)
    try:
        c = aa.Context(0)
        for det in (False, True):
            assert_pairs_equal(c.atomic_contacts(prod.view("/"), aa.default_params(deterministic=det)), want, f"grown list det={det}")
    finally:
        aa.debug_set("defer_entries", 0)
    with pytest.raises(aa.ArpeggiaError):
        aa.debug_set("no_such_switch", 1)


# ---------------------------------------------------------------------------------------------- SAP neighbour sum
def test_sap_neighbor_sum_matches_the_restatement(ctx):
    """src/sap.rs:155-204: f32 sum of hydrophobicity x relative side-chain SASA over the side-chain atoms within 5 A, self included.
    The reference accumulates in R*-tree order, the oracle in index order, the device in slot order: f32 sums agree to rounding."""
    rng = np.random.default_rng(12)
    for path_or_n in ("6bft", 40000, 100000):
        if path_or_n == "6bft":
            rec = synth.read_pdb_records(synth.DATA / "6bft.pdb")
        else:
            rec = synth.gen_s1(path_or_n)
        n = len(rec["x"])
        backbone = np.isin(rec["name"], [b"N", b"CA", b"C", b"O", b"OXT"])
        side = (~backbone) & (rec["resn"] != b"HOH") & (rec["element"] != b"H")
        sasa = rng.uniform(0.0, 60.0, n).astype(np.float32)
        sasa[rng.random(n) < 0.3] = 0.0  # buried atoms
        names = {nm: nm.decode() for nm in np.unique(rec["resn"])}
        w = np.array([aa.sap_weight(names[r], float(a)) for r, a in zip(rec["resn"], sasa)], dtype=np.float32)
        w_orc = np.array([ob.sap_weight(names[r], float(a)) for r, a in zip(rec["resn"], sasa)], dtype=np.float32)
        assert np.array_equal(w, w_orc) and (w != 0).any() and aa.sap_weight("HOH", 10.0) == 0.0 and aa.sap_weight("GLY", 99.0) == 0.0
        got = aa.sap_neighbor_sum(ctx, rec["x"], rec["y"], rec["z"], side, w, 5.0)
        want = ob.sap_neighbor_sum(rec["x"], rec["y"], rec["z"], side, w, 5.0)
        assert (got[~side] == 0).all() and np.abs(want[side]).max() > 1.0
        assert np.abs(got - want).max() <= 2e-5 * max(1.0, float(np.abs(want).max())), float(np.abs(got - want).max())
        again = aa.sap_neighbor_sum(ctx, rec["x"], rec["y"], rec["z"], side, w, 5.0)
        assert np.array_equal(got, again)  # slot order is a function of the input: bit-reproducible
    # the radius is inclusive and squared in f32 (sap.rs:183)
    x = np.array([0.0, 5.0, 0.0, float(np.nextafter(np.float64(5.0), 6.0))])
    y = np.array([0.0, 0.0, 50.0, 50.0])
    z0 = np.zeros(4)
    one, all_sc = np.ones(4, dtype=np.float32), np.ones(4, dtype=np.uint8)
    got = aa.sap_neighbor_sum(ctx, x, y, z0, all_sc, one, 5.0)
    assert got.tolist() == ob.sap_neighbor_sum(x, y, z0, all_sc, one, 5.0).tolist() == [2.0, 2.0, 1.0, 1.0]


# ---------------------------------------------------------------------------------------------- the table (get_contacts)
def _table_lines(cols):
    out = []
    names = _lib.INTERACTIONS
    g = lambda v: repr(float(np.float32(v)))
    for k in range(len(cols["model"])):
        d = lambda c: cols[c][k].decode()
        sc = (g(cols["sc_centroid_dist"][k]), g(cols["sc_dihedral"][k]), g(cols["sc_centroid_angle"][k])) if cols["sc_valid"][k] else ("", "", "")
        out.append(",".join([str(cols["model"][k]), names[cols["interaction"][k]], g(cols["distance"][k]),
                             d("from_chain"), d("from_resn"), str(cols["from_resi"][k]), d("from_insertion"), d("from_altloc"), d("from_atomn"), str(cols["from_atomi"][k]),
                             d("to_chain"), d("to_resn"), str(cols["to_resi"][k]), d("to_insertion"), d("to_altloc"), d("to_atomn"), str(cols["to_atomi"][k]), *sc]))
    return out


def _lines_close(got, want):
    assert len(got) == len(want)
    for a, b in zip(got, want):
        fa, fb = a.split(","), b.split(",")
        assert fa[:2] == fb[:2] and fa[3:17] == fb[3:17], (a, b)
        for x, y, tol in ((fa[2], fb[2], DIST_TOL), (fa[17], fb[17], DIST_TOL), (fa[18], fb[18], ANGLE_TOL_DEG), (fa[19], fb[19], ANGLE_TOL_DEG)):
            assert (x == "") == (y == ""), (a, b)
            if x:
                assert abs(float(x) - float(y)) <= tol, (a, b)


def test_first_table_call_then_immediate_free_and_first_calls_from_several_threads():
    """the first get_contacts call on a structure starts a job that fills the entity book while the device works; freeing the
    structure right after that call (the table still alive: it owns its columns) and first calls on several structures from several threads
    (each with its own context) must both be safe."""
    import threading

    path = str(synth.DATA / "6bft.pdb")
    c = aa.Context(0)
    for _ in range(3):
        s = aa.load_model(path)
        cols = c.get_contacts(s, "/", 0.1, 6.5)
        del s  # arp_structure_free right behind the first call
        assert len(cols["model"]) == 7236 and cols["from_chain"][0] != b""
    results, errors = [None] * 4, []

    def work(k):
        try:
            ctx_k = aa.Context(0)
            st = aa.load_model(path if k % 2 == 0 else str(synth.DATA / "1ubq.pdb"))
            results[k] = len(ctx_k.get_contacts(st, "/", 0.1, 6.5)["model"])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors
    assert results == [7236, 532, 7236, 532]


def test_device_planes_phe4_of_1ubq(ctx):
    """the ring plane of PHE 4 as fitted ON THE DEVICE against the numbers of the reference's own unit test
    (residues.rs:355-372; the normal is defined up to sign, nalgebra's SVD returns the negated vector)."""
    s = aa.load_model(str(synth.DATA / "1ubq.pdb"))
    nres = int(_lib.lib.arp_structure_n_residues(s._h))
    planes, valid = np.zeros((nres, 12)), np.zeros(nres, dtype=np.uint8)
    st = _lib.lib.arp_structure_fit_planes(ctx._h, s._h, planes.ctypes.data_as(C.POINTER(C.c_double)), valid.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert st == 0, _lib.lib.arp_last_error()
    assert nres == 134 and int((valid & 1).sum()) == 4 and int(((valid & 2) != 0).sum()) == 68  # 4 rings, 68 sc planes
    phe4 = planes[3]  # MET1 GLN2 ILE3 PHE4
    assert valid[3] & 1
    assert np.allclose(phe4[0:3], [24.96883333, 34.687, 6.16233333], atol=1e-6)
    want = np.array([0.53253994, -0.82736044, -0.17853828])
    assert min(np.abs(phe4[3:6] - want).max(), np.abs(phe4[3:6] + want).max()) < 1e-6
    # every plane against the oracle's fit (Hestenes SVD there, symmetric Jacobi on the device)
    orc = ob.Structure.load(str(synth.DATA / "1ubq.pdb"))
    for pl in orc.planes("ring"):
        r = int(pl["res_idx"])
        assert np.allclose(planes[r, 0:3], pl["c"], atol=1e-9)
        assert min(np.abs(planes[r, 3:6] - pl["n"]).max(), np.abs(planes[r, 3:6] + pl["n"]).max()) < 1e-9


def table_cases():
    return {"6bft": aa.load_model(str(synth.DATA / "6bft.pdb")), "stress_altlocs": aa.Structure.from_records(synth.gen_stress(n_res=300, seed=5, altlocs=True)),
            "two_models": aa.Structure.from_records(synth.gen_stress(n_res=120, seed=9, n_models=2))}


def test_device_table_equals_the_host_assembly(ctx):
    """The device table (plane fits, ring rows, sort, sc statistics as kernels) against the round-1 host assembly of the same pair
    list, row for row: 6bft (17 CationPi + 43 pi rows), a stress structure with altlocs and insertion-free ties, two models.  The host
    assembly is not part of the product library: it lives in a test-only build (arpeggia_amd/build.py build_host_table_library) that a
    child process loads."""
    import json
    import os
    import subprocess
    import sys
    from arpeggia_amd import build as B

    lib = B.build_host_table_library()
    env = dict(os.environ, ARPEGGIA_AMD_LIB=str(lib))  # (the child switches the host assembly on: arp_debug_set("table_host", 1))
    r = subprocess.run([sys.executable, str(synth.DATA.parent / "hosttable" / "dump_table.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    host = json.loads(r.stdout.strip().splitlines()[-1])
    for name, s in table_cases().items():
        for groups in ("/", "A,B/"):
            dev_rows = _table_lines(ctx.get_contacts(s, groups, 0.1, 6.5))
            _lines_close(dev_rows, host[name][groups])
            if name == "6bft" and groups == "/":
                kinds = [ln.split(",")[1] for ln in dev_rows]
                assert len(dev_rows) == 7236 and kinds.count("CationPi") == 17 and sum(x.startswith("Pi") for x in kinds) == 43


@pytest.mark.parametrize("name,rows", [("1ubq", 532), ("6bft", 7236)])
def test_table_matches_golden(ctx, name, rows):
    # python/tests/test_arpeggia.py:32-35: 1ubq -> 532 rows; golden CSVs are restatement-derived (tests/golden/make_golden.py)
    s = aa.load_model(str(synth.DATA / f"{name}.pdb"))
    cols = ctx.get_contacts(s, "/", 0.1, 6.5)
    assert len(cols["model"]) == rows
    gold = (GOLDEN / f"{name}_contacts.csv").read_text().splitlines()[1:]
    _lines_close(_table_lines(cols), gold)


def test_contacts_drop_in_surface(ctx):
    # python/tests/test_arpeggia.py:28-71
    df = aa.contacts(str(synth.DATA / "1ubq.pdb"), groups="/", vdw_comp=0.1, dist_cutoff=6.5)
    height = df.height if hasattr(df, "height") else df.num_rows
    width = df.width if hasattr(df, "width") else df.num_columns
    assert height == 532 and width == 20
    import pyarrow as pa

    cols = df.column_names if isinstance(df, pa.Table) else df.columns
    assert list(cols) == [c for c, _ in aa.TABLE_COLUMNS]

    if isinstance(df, pa.Table):
        assert df.schema.field("model").type == pa.uint32() and df.schema.field("distance").type == pa.float32()
        assert df.schema.field("from_resi").type == pa.int32() and df.schema.field("interaction").type == pa.string()
        assert df.schema.field("sc_dihedral").type == pa.float32()
        assert min(df.column("distance").to_pylist()) >= 0
    # ignore_zero_occupancy is a no-op on 1ubq (test_arpeggia.py:85-112)
    df2 = aa.contacts(str(synth.DATA / "1ubq.pdb"), ignore_zero_occupancy=True)
    assert (df2.height if hasattr(df2, "height") else df2.num_rows) == 532


@pytest.mark.parametrize("name", ["1ubq", "6bft"])
def test_arrow_export_equals_the_column_accessors(ctx, name):
    """arp_table_export_arrow (what contacts() returns) against arp_table_column (what the golden comparison reads)."""
    import pyarrow as pa

    s = aa.load_model(str(synth.DATA / f"{name}.pdb"))
    cols = ctx.get_contacts(s, "/", 0.1, 6.5)
    df = aa.get_contacts(s)
    tab = df if isinstance(df, pa.Table) else df.to_arrow()
    assert tab.num_rows == len(cols["model"]) and tab.num_columns == 20
    for cname, kind in aa.TABLE_COLUMNS:
        got = tab.column(cname).to_pylist()
        if cname == "interaction":
            want = [_lib.INTERACTIONS[k] for k in cols[cname]]
        elif kind == "str":
            want = [v.decode() for v in cols[cname]]
        elif cname.startswith("sc_"):
            want = [float(v) if ok else None for v, ok in zip(cols[cname], cols["sc_valid"])]
            assert tab.column(cname).null_count == int((~cols["sc_valid"]).sum())
        else:
            want = cols[cname].tolist()
        assert got == want, cname
    tab.validate(full=True)


def test_table_on_stress_structure_matches_oracle_table(ctx, tmp_path):
    rec = synth.gen_stress(n_res=250, seed=31, n_chains=3)
    p = tmp_path / "stress.pdb"
    synth.write_pdb(rec, p)
    s, o = aa.load_model(p), ob.Structure.load(p)
    for groups in ("/", "A/B,C"):
        cols = ctx.get_contacts(s, groups, 0.1, 6.5)
        want = ob.rows_to_csv_lines(o.get_contacts(groups, 0.1, 6.5))
        _lines_close(_table_lines(cols), want)


@pytest.mark.parametrize("n_res", [20, 70, 130, 200])
def test_small_tables_in_one_launch_match_the_oracle(ctx, tmp_path, n_res):
    """Round 5: tables of up to 2048 contact pairs are counted, expanded, keyed, sorted (a bitonic network over one workgroup: 1, 2 or 4 sort words per
    thread), tie-fixed and finished by ONE kernel (k_table_small) that writes into the host's landing buffer.  Graded sizes around the words-per-thread
    steps, with hydrogens, altlocs and two chain groups, row for row against the oracle's table; and the same structure twice in a row (the
    landing buffer and the counters are reused)."""
    rec = synth.gen_stress(n_res=n_res, seed=500 + n_res, n_chains=2)
    p = tmp_path / "small.pdb"
    synth.write_pdb(rec, p)
    s, o = aa.load_model(p), ob.Structure.load(p)
    for groups in ("/", "A/B"):
        want = ob.rows_to_csv_lines(o.get_contacts(groups, 0.1, 6.5))
        for _ in range(2):
            _lines_close(_table_lines(ctx.get_contacts(s, groups, 0.1, 6.5)), want)
    assert len(want) > 0


def test_table_with_long_runs_of_tied_sort_keys(ctx, tmp_path):
    """Rows whose ten sort keys tie are ordered by (from_insertion, to_insertion, distance).  Here 26 residues at a time share their residue
    NUMBER (insertion codes A..Z) and every atom of the file carries serial 5, so all rows between such residues with the same interaction tie:
    runs far beyond the 64 rows the device's tie pass resolves in place, i.e. the table goes through the second, long-way sort -- and the
    insertion codes decide the order.  A milder variant (serials kept) keeps the runs short: the in-place pass."""
    base = synth.gen_stress(n_res=104, seed=77, n_chains=1, hydrogens=False)
    key = np.char.add(np.char.add(base["chain"].astype("U8"), "|"), base["resi"].astype("U12"))
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    res_of = np.argsort(np.argsort(first))[inv]
    for same_serial in (True, False):
        rec = {k: v.copy() for k, v in base.items()}
        rec["resi"] = (10 + res_of // 26).astype(rec["resi"].dtype)
        rec["icode"] = np.array([bytes([65 + int(r) % 26]) for r in res_of], dtype=rec["icode"].dtype)
        if same_serial:
            rec["serial"][:] = 5
        p = tmp_path / f"ties_{int(same_serial)}.pdb"
        synth.write_pdb(rec, p)
        s, o = aa.load_model(p), ob.Structure.load(p)
        want = ob.rows_to_csv_lines(o.get_contacts("/", 0.1, 6.5))
        assert len(want) > 2000
        _lines_close(_table_lines(ctx.get_contacts(s, "/", 0.1, 6.5)), want)


def test_disk_files_with_altlocs_insertion_codes_and_two_models_end_to_end(ctx, tmp_path):
    """a file with alternate locations, insertion codes and two MODEL records, read from DISK as PDB and as mmCIF (quoted
    values, a multi-line text field, wrapped rows, label_* numbering that differs from the author numbering), through contacts() to the table,
    against the oracle's table of the PDB file."""
    rec = synth.with_insertion_codes(synth.gen_stress(n_res=160, seed=43, n_models=2, altlocs=True, n_chains=3))
    pdb, cif = tmp_path / "alt.pdb", tmp_path / "alt.cif"
    synth.write_pdb(rec, pdb)
    synth.write_mmcif(rec, cif, fancy=True)
    orc = ob.Structure.load(pdb)
    for groups in ("/", "A/B,C"):
        want = ob.rows_to_csv_lines(orc.get_contacts(groups, 0.1, 6.5))
        assert len(want) > 500 and any(",A," in ln for ln in want)
        for path in (pdb, cif):
            s = aa.load_model(path)
            _lines_close(_table_lines(ctx.get_contacts(s, groups, 0.1, 6.5)), want)
    df = aa.contacts(str(cif))
    assert df.num_rows == len(ob.rows_to_csv_lines(orc.get_contacts("/", 0.1, 6.5)))
    assert set(df.column("from_insertion").to_pylist()) >= {"", "A"} and set(df.column("from_altloc").to_pylist()) >= {"", "A", "B"}
    assert set(df.column("model").to_pylist()) == {1, 2}


def test_table_multi_model_and_host_threads(ctx, tmp_path):
    """Two models (the plane tables visit all chains of all models under every model serial, complex.rs:447-449) against the
    oracle's table, and the same table for 1 and 8 host threads."""
    rec = synth.gen_stress(n_res=150, seed=33, n_models=2, n_chains=2)
    p = tmp_path / "two_models.pdb"
    synth.write_pdb(rec, p)
    s, o = aa.load_model(p), ob.Structure.load(p)
    want = ob.rows_to_csv_lines(o.get_contacts("/", 0.1, 6.5))
    tables = []
    try:
        for threads in (1, 8):
            _lib.lib.arp_set_num_threads(threads)
            assert _lib.lib.arp_get_num_threads() == threads
            cols = ctx.get_contacts(s, "/", 0.1, 6.5)
            _lines_close(_table_lines(cols), want)
            tables.append(_table_lines(cols))
    finally:
        _lib.lib.arp_set_num_threads(1)
    assert tables[0] == tables[1]


def test_cli_contacts_end_to_end(ctx, tmp_path):
    # cli/contacts.rs:54-138: <output>/<filename>.<format>, 532 rows + header for 1ubq
    from arpeggia_amd.__main__ import main

    rc = main(["contacts", "-i", str(synth.DATA / "1ubq.pdb"), "-o", str(tmp_path / "out"), "-f", "ubq", "-t", "csv"])
    assert rc == 0
    lines = (tmp_path / "out" / "ubq.csv").read_text().splitlines()
    assert len(lines) == 533 and lines[0].replace('"', "").split(",") == [c for c, _ in aa.TABLE_COLUMNS]
    assert main(["contacts", "-i", str(tmp_path / "missing.pdb"), "-o", str(tmp_path / "out")]) == 1


def test_contacts_batch_over_files(ctx, tmp_path):
    # many files through worker threads == the same files one by one, in input order
    files = [str(synth.DATA / "1ubq.pdb"), str(synth.DATA / "6bft.pdb")] * 6
    p = tmp_path / "stress.pdb"
    synth.write_pdb(synth.gen_stress(n_res=120, seed=91, n_chains=2), p)
    files.append(str(p))
    one_by_one = [aa.contacts(f) for f in files[:2]] + [aa.contacts(files[-1])]
    got = aa.contacts_batch(files, num_workers=4)
    assert len(got) == len(files)
    rows = lambda t: t.num_rows if hasattr(t, "num_rows") else t.height
    assert [rows(t) for t in got[:2]] == [532, 7236]
    for k, t in enumerate(got):
        ref = one_by_one[2] if k == len(files) - 1 else one_by_one[k % 2]
        assert t.equals(ref), k
    assert aa.contacts_batch([]) == []


def test_no_ring_structure_is_an_error(ctx):
    # complex.rs:50,480-482: panics when the model has no HIS/PHE/TYR/TRP ring
    prod, orc = _both_from(_mini([[0, 0, 0], [3, 0, 0], [0, 3, 0]]))
    with pytest.raises(ob.OracleError) as eo:
        orc.get_contacts()
    assert eo.value.code == ob.ORC_ERR_NO_RINGS
    with pytest.raises(aa.ArpeggiaError) as e:
        ctx.get_contacts(prod)
    assert e.value.status == _lib.ARP_ERR_NO_RINGS


def test_compiled_c_consumer_runs_the_integration_sequence(c_consumer, ubq_path):
    """INTEGRATION.md section 3 from a plain C program: load 1ubq -> arp_get_contacts -> columns -> Arrow C Data export -> release.  532 rows is
    the count the reference's own test pins (python/tests/test_arpeggia.py:35), 20 columns its width (:67)."""
    import subprocess

    r = subprocess.run([c_consumer, ubq_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[-1].startswith("532 rows"), r.stdout
    cols = lines[0].split()[1:]
    assert len(cols) == 20 and cols[0] == "model" and "interaction" in cols and "sc_centroid_angle" in cols


def test_handwritten_mmcif_table_equals_the_pdb_twin_and_the_oracle(ctx):
    """End to end from the hand-written deposition-layout mmCIF (tests/data/hand7.cif; see tests/test_host_cpu.py): its contact table equals
    the table of the PDB twin, which equals the oracle's table row for row."""
    want = ob.rows_to_csv_lines(ob.Structure.load(str(synth.DATA / "hand7.pdb")).get_contacts("/", 0.1, 6.5))
    assert len(want) == 8 and {ln.split(",")[0] for ln in want} == {"1", "2"}  # (a heptapeptide: four rows in each of its two models)
    for ext in ("cif", "pdb"):
        _lines_close(_table_lines(ctx.get_contacts(aa.load_model(str(synth.DATA / f"hand7.{ext}")), "/", 0.1, 6.5)), want)
