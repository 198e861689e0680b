"""Pin the CPU oracle against every fact the reference's own tests hold for the contacts path.

CPU-only.  Reference citations are relative to the reference checkout (y1zhou/arpeggia v0.8.0).
"""
import numpy as np
import pytest

import oracle_binding as ob
from conftest import GOLDEN

EXPECTED_COLUMNS = [
    "model", "interaction", "distance",
    "from_chain", "from_resn", "from_resi", "from_insertion", "from_altloc", "from_atomn", "from_atomi",
    "to_chain", "to_resn", "to_resi", "to_insertion", "to_altloc", "to_atomn", "to_atomi",
    "sc_centroid_dist", "sc_dihedral", "sc_centroid_angle",
]


@pytest.fixture(scope="module")
def ubq(ubq_path):
    return ob.Structure.load(ubq_path)


@pytest.fixture(scope="module")
def bft(bft_path):
    return ob.Structure.load(bft_path)


def test_1ubq_532_rows(ubq):
    # python/tests/test_arpeggia.py:32-35: contacts(1ubq, "/", 0.1, 6.5).height == 532
    rows = ubq.get_contacts("/", 0.1, 6.5)
    assert len(rows) == 532
    # :70-71 distances are floats >= 0
    assert (rows["distance"] >= 0).all()
    kinds = {ob.INTERACTIONS[k]: int(c) for k, c in zip(*np.unique(rows["interaction"], return_counts=True))}
    # restatement-derived split; the total is what the reference pins
    assert kinds == {"VanDerWaalsContact": 199, "HydrophobicContact": 167, "PolarContact": 107, "WeakPolarContact": 55, "IonicBond": 4}


def test_table_has_20_columns(ubq):
    # python/tests/test_arpeggia.py:38-67: 20 named columns
    lines = ob.rows_to_csv_lines(ubq.get_contacts("/", 0.1, 6.5))
    header = (GOLDEN / "1ubq_contacts.csv").read_text().splitlines()[0].split(",")
    assert header == EXPECTED_COLUMNS
    assert all(len(l.split(",")) == 20 for l in lines)


def test_golden_csv_matches_oracle(ubq, bft):
    for s, name, n in ((ubq, "1ubq", 532), (bft, "6bft", 7236)):
        gold = (GOLDEN / f"{name}_contacts.csv").read_text().splitlines()[1:]
        assert len(gold) == n
        assert ob.rows_to_csv_lines(s.get_contacts("/", 0.1, 6.5)) == gold


def test_1ubq_atoms_and_waters(ubq):
    # python/tests/test_arpeggia.py:122 (602 protein atoms); chains.rs:21-40 (58 waters stay in chain A after residue 76)
    a = ubq.atoms
    assert len(a) == 660
    assert (a["res_resn"] != b"HOH").sum() == 602
    assert (a["res_resn"] == b"HOH").sum() == 58
    assert set(a["chain"]) == {b"A"}
    prot_ord = a["res_ord"][a["res_resn"] != b"HOH"]
    water_ord = a["res_ord"][a["res_resn"] == b"HOH"]
    assert prot_ord.min() == 0 and prot_ord.max() == 75
    assert sorted(set(water_ord)) == list(range(76, 76 + 58))
    assert (a["model_serial"] == 0).all()  # aromatic.rs:86-90: model serial 0 without MODEL records


def test_zero_occupancy_strip_is_noop_on_1ubq(ubq_path):
    # utils.rs:230-247, python/tests/test_arpeggia.py:85-112
    a = ob.Structure.load(ubq_path, ignore_zero_occupancy=False)
    b = ob.Structure.load(ubq_path, ignore_zero_occupancy=True)
    assert len(a.atoms) == len(b.atoms)
    assert len(a.get_contacts()) == len(b.get_contacts()) == 532


def test_phe4_ring_plane(ubq):
    # residues.rs:334-395: PHE4 ring centre / normal at 1e-6 relative; normal sign is arbitrary (SVD)
    rings = ubq.planes("ring")
    assert len(rings) == 4  # PHE4, PHE45, TYR59, HIS68
    f4 = rings[(rings["resn"] == b"PHE") & (rings["resi"] == 4)][0]
    np.testing.assert_allclose(f4["c"], [24.96883333, 34.687, 6.16233333], rtol=1e-6)
    n_ref = np.array([0.53253994, -0.82736044, -0.17853828])
    n = f4["n"] * np.sign(f4["n"] @ n_ref)
    np.testing.assert_allclose(n, n_ref, rtol=1e-6)
    # normal is orthogonal to the ring atoms (:387-394)
    a = ubq.atoms
    sel = (a["res_resn"] == b"PHE") & (a["resi"] == 4) & np.isin(a["name"], [b"CG", b"CD1", b"CD2", b"CE1", b"CE2", b"CZ"])
    xyz = np.stack([a["x"][sel], a["y"][sel], a["z"][sel]], 1) - f4["c"]
    assert sel.sum() == 6
    assert np.abs(xyz @ f4["n"]).mean() < 0.02
    # MET1: no ring atoms, 3 sc-plane atoms -> has an sc plane (residues.rs:345-348)
    sc = ubq.planes("sc")
    assert ((sc["resn"] == b"MET") & (sc["resi"] == 1)).sum() == 1
    assert len(sc) == 68


def test_plane_identities():
    # residues.rs:306-332
    plane_x = ([0, 0, 0], [0, 0, 1])
    point = [0, 1, 1]
    parallel = (point, [0, 0, -1])
    d, dih, ang = ob.plane_metrics(*plane_x, *parallel)
    assert abs(d - 2 ** 0.5) < 1e-6 and abs(ang - 45.0) < 1e-6 and dih < 1e-6
    _, _, ang2 = ob.plane_metrics(*parallel, *plane_x)
    assert abs(ang2 - 45.0) < 1e-6
    perp = (point, [1, 0, 0])
    _, dih, _ = ob.plane_metrics(*plane_x, *perp)
    assert abs(dih - 90.0) < 1e-6
    _, _, ang3 = ob.plane_metrics(*perp, *plane_x)
    assert abs(ang3 - 90.0) < 1e-6


def _ring(bft, chain, resi):
    rings = bft.planes("ring")
    return rings[(rings["chain"] == chain) & (rings["resi"] == resi)][0]


def _atom(bft, chain, resi, name):
    a = bft.atoms
    return a[(a["chain"] == chain) & (a["resi"] == resi) & (a["name"] == name)][0]


def test_6bft_cation_pi_cases(bft):
    rows = bft.get_contacts("/", 0.1, 6.5)
    cp = rows[rows["interaction"] == ob.INTERACTIONS.index("CationPi")]

    def has(ring_chain, ring_resi, chain, resi, atomn):
        f, t = cp["from"], cp["to"]
        return bool(((f["chain"] == ring_chain) & (f["resi"] == ring_resi) & (f["atomn"] == b"Ring") &
                     (t["chain"] == chain) & (t["resi"] == resi) & (t["atomn"] == atomn)).any())

    # aromatic.rs:72-99: TYR A102 ring .. ARG G82 NE => CationPi
    assert has(b"A", 102, b"G", 82, b"NE")
    ring, ne = _ring(bft, b"A", 102), _atom(bft, b"G", 82, b"NE")
    d, _, theta = ob.plane_metrics(ring["c"], ring["n"], [ne["x"], ne["y"], ne["z"]], [0, 0, 1])
    assert abs(d - 3.4419) < 1e-3 and abs(theta - 17.44) < 1e-2  # SURVEY.md Appendix C probe values
    # aromatic.rs:101-128: TRP A108 ring .. LYS G84 NZ => None
    assert not has(b"A", 108, b"G", 84, b"NZ")
    ring, nz = _ring(bft, b"A", 108), _atom(bft, b"G", 84, b"NZ")
    d, _, _ = ob.plane_metrics(ring["c"], ring["n"], [nz["x"], nz["y"], nz["z"]], [0, 0, 1])
    assert d > 4.5
    # ring rows carry atomi == 0 (complex.rs:334-342)
    assert (cp["from"]["atomi"] == 0).all()


def test_parse_groups_cases():
    # utils.rs:174-212
    chains = ["A", "B", "C", "D"]
    assert ob.parse_groups(chains, "A,B/C,D") == ({"A", "B"}, {"C", "D"})
    assert ob.parse_groups(chains, "A/C,D") == ({"A"}, {"C", "D"})
    assert ob.parse_groups(chains, "/C,D") == ({"A", "B"}, {"C", "D"})
    assert ob.parse_groups(chains, "C/") == ({"C"}, {"A", "B", "D"})
    assert ob.parse_groups(chains, "/") == (set(chains), set(chains))


def test_parse_groups_panics():
    # utils.rs:214-228: the two should_panic messages
    with pytest.raises(ob.OracleError, match="Invalid chain groups format! Use '/' for all-to-all comparisons."):
        ob.parse_groups(["A", "B", "C", "D"], "")
    with pytest.raises(ob.OracleError, match="Empty chain groups!"):
        ob.parse_groups(["A", "B", "C"], "A,B,C/")


def test_grid_search_equals_brute_force(ubq, bft):
    for s in (ubq, bft):
        for cutoff in (6.5, 4.0):
            g = s.atomic_contacts("/", 0.1, cutoff, brute=False)
            b = s.atomic_contacts("/", 0.1, cutoff, brute=True)
            key = lambda p: np.lexsort((p["j"], p["i"]))
            g, b = g[key(g)], b[key(b)]
            assert np.array_equal(g, b)


def test_candidate_pair_counts(ubq, bft):
    # SURVEY.md 8a probe values (restatement-derived)
    assert len(ubq.atomic_contacts()) == 9128
    assert len(bft.atomic_contacts()) == 124047


def test_angle_dihedral_helpers():
    assert abs(ob.angle([1, 0, 0], [0, 0, 0], [0, 1, 0]) - 90.0) < 1e-12
    assert abs(ob.angle([1, 0, 0], [0, 0, 0], [-1, 0, 0]) - 180.0) < 1e-12
    # unsigned dihedral in [0, 180]
    assert abs(ob.dihedral([1, 0, 0], [0, 0, 0], [0, 0, 1], [0, 1, 1]) - 90.0) < 1e-12
    assert abs(ob.dihedral([1, 0, 0], [0, 0, 0], [0, 0, 1], [0, -1, 1]) - 90.0) < 1e-12
    assert abs(ob.dihedral([1, 0, 0], [0, 0, 0], [0, 0, 1], [1, 0, 1]) - 0.0) < 1e-6


def test_sap_weight_reference_facts():
    """The facts the reference's own tests hold about the SAP tables (src/sap.rs:362-376 test_hydrophobicity_values, :379-403
    test_max_asa_values), asserted on the oracle's restatement AND on the product's arp_sap_weight (a host function: no GPU needed).
    weight(resn, sasa) = hydrophobicity(resn) * clamp(sasa / max_sc_asa(resn), 0, 1)  (sap.rs:198-209), so a saturating sasa reads the
    hydrophobicity and a small one the maximum side-chain ASA."""
    import arpeggia_amd as aa

    amino_acids = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLU", "GLN", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"]
    for weight in (ob.sap_weight, aa.sap_weight):
        hyd = {a: weight(a, 1e6) for a in amino_acids}           # sasa far above any maximum: the ratio clamps to 1
        assert abs(hyd["GLY"] - 0.0) < 1e-6                        # sap.rs:364-365 glycine is the reference point
        assert all(weight("GLY", s) == 0.0 for s in (0.0, 1.0, 50.0))
        assert hyd["PHE"] > 0.4 and hyd["PHE"] == max(hyd.values())   # :367-369 "most hydrophobic"
        assert hyd["ARG"] < -0.4 and hyd["ARG"] == min(hyd.values())  # :371-373 "most hydrophilic"
        assert weight("XXX", 50.0) == 0.0                          # :375-376 unknown residue -> None -> contributes nothing (sap.rs:198-209)
        max_asa = {}
        for a in amino_acids:                                      # :381-394 every standard amino acid has a positive maximum side-chain ASA
            if hyd[a] == 0.0:
                continue                                           # (glycine's cannot be read through a zero weight)
            w = weight(a, 1.0)
            assert w != 0.0 and (w > 0) == (hyd[a] > 0), a
            max_asa[a] = hyd[a] / w                                # 1 A^2 of SASA is below every maximum: ratio = 1 / max
            assert 1.0 < max_asa[a] < 300.0, (a, max_asa[a])
        assert len(max_asa) == 19
        assert max_asa["TRP"] > max_asa["ALA"] and max_asa["TRP"] > 90.0   # :397-403 larger residues, larger maxima (TRP vs the small ones)
        assert weight("PHE", -5.0) == 0.0                          # the ratio is clamped at 0 too
