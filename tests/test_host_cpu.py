"""CPU-only checks of the product's host side: the C-ABI library loads and exports what include/arpeggia_amd.h declares,
the structure ingest agrees with the oracle's independent reader, and the once-per-atom attribute words agree with the
reference's per-pair string rules as restated by the oracle.  No compute call is made (there is no GPU here)."""
import re

import numpy as np
import pytest

import arpeggia_amd as aa
import oracle_binding as ob
import synth
from arpeggia_amd import _lib
from conftest import ROOT


def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "arpeggia_amd.h").read_text()
    declared = sorted(set(re.findall(r"\b(arp_[a-z_0-9]+)\s*\(", header)))
    assert len(declared) >= 25
    import ctypes

    L = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib.arp_api_version() == 2


def test_struct_layouts_match_header(c_consumer):
    """Field by field: the offsets and sizes a C compiler gives the header's structs (printed by the compiled consumer, whose own
    _Static_asserts pin them to the numbers INTEGRATION.md's #[repr(C)] block assumes) against the ctypes mirror in _lib.py."""
    import ctypes as C
    import json
    import subprocess

    r = subprocess.run([c_consumer, "--abi"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    abi = json.loads(r.stdout)
    assert abi["api_version"] == _lib.lib.arp_api_version() == 2
    mirrors = {"arp_atoms": _lib.arp_atoms, "arp_params": _lib.arp_params, "arp_pair": _lib.arp_pair, "arp_pairs": _lib.arp_pairs, "arp_records": _lib.arp_records}
    assert set(abi["structs"]) == set(mirrors)
    for name, T in mirrors.items():
        c = abi["structs"][name]
        assert C.sizeof(T) == c["sizeof"], name
        assert [f[0] for f in T._fields_] == list(c["fields"]), f"{name}: field order"
        for fname, _ in [(f[0], f[1]) for f in T._fields_]:
            d = getattr(T, fname)
            assert [d.offset, d.size] == c["fields"][fname], f"{name}.{fname}"
    assert aa.PAIR_DTYPE.itemsize == 16 and [aa.PAIR_DTYPE.fields[k][1] for k in ("i", "j", "dist", "kind")] == [0, 4, 8, 12]


def test_interaction_vocabulary():
    # structs.rs:6-51 order; Display == variant name (structs.rs:151-157)
    names = [_lib.lib.arp_interaction_name(k).decode() for k in range(19)]
    assert names == _lib.INTERACTIONS == ob.INTERACTIONS


@pytest.mark.skipif(aa.device_count() > 0, reason="only meaningful without a GPU")
def test_no_cpu_fallback_without_device():
    with pytest.raises(aa.ArpeggiaError) as e:
        aa.Context(0)
    assert e.value.status == _lib.ARP_ERR_NO_DEVICE
    with pytest.raises(aa.ArpeggiaError):
        aa.contacts(str(ROOT / "tests" / "data" / "1ubq.pdb"))


def test_parse_groups_cases_and_panic_strings():
    # utils.rs:174-228
    chains = ["A", "B", "C", "D"]
    assert aa.parse_groups(chains, "A,B/C,D") == ({"A", "B"}, {"C", "D"})
    assert aa.parse_groups(chains, "A/C,D") == ({"A"}, {"C", "D"})
    assert aa.parse_groups(chains, "/C,D") == ({"A", "B"}, {"C", "D"})
    assert aa.parse_groups(chains, "C/") == ({"C"}, {"A", "B", "D"})
    assert aa.parse_groups(chains, "/") == (set(chains), set(chains))
    with pytest.raises(aa.ArpeggiaError, match="Invalid chain groups format! Use '/' for all-to-all comparisons."):
        aa.parse_groups(chains, "")
    with pytest.raises(aa.ArpeggiaError, match="Empty chain groups!"):
        aa.parse_groups(["A", "B", "C"], "A,B,C/")


def _compare_structure(prod: aa.Structure, orc: ob.Structure, groups="/"):
    soa = prod.soa(groups)
    oa = orc.atoms
    assert prod.n_atoms == len(oa)
    for k in ("x", "y", "z"):
        assert np.array_equal(soa[k], oa[k])
    assert np.array_equal(soa["res_ord"], oa["res_ord"].astype(np.uint32))
    assert np.array_equal(soa["res_id"], oa["res_idx"].astype(np.uint32))
    assert np.array_equal(prod.strings("chain"), oa["chain"])
    assert np.array_equal(prod.strings("resn"), oa["res_resn"])
    assert np.array_equal(prod.strings("atomn"), oa["name"])
    assert np.array_equal(prod.strings("altloc"), oa["altloc"])
    assert np.array_equal(prod.strings("insertion"), oa["icode"])
    assert np.array_equal(prod.ints("resi"), oa["resi"])
    assert np.array_equal(prod.ints("atomi"), oa["serial"])
    assert np.array_equal(prod.ints("model"), oa["model_serial"])
    # attribute word vs the oracle's string predicates
    cls = ob.atom_classes(oa)
    A = _lib.ATTR
    pairs = [("DONOR", ob.lib().__class__ and 1), ]
    mapping = {"DONOR": 1, "ACCEPTOR": 2, "WEAK_DONOR": 4, "POS": 8, "NEG": 16, "HYDROPHOBIC": 32, "CYS_SG": 64, "H": 128, "POS_RESN": 256}
    for name, obit in mapping.items():
        got = (soa["attr"] & A[name]) != 0
        want = (cls & obit) != 0
        assert np.array_equal(got, want), name
    # chain rank == rank under byte-wise order
    ids = sorted(set(oa["chain"]))
    assert np.array_equal(soa["chain_rank"], np.array([ids.index(c) for c in oa["chain"]], dtype=np.uint32))
    # element class radii
    prm = aa.default_params()
    import ctypes as C

    for e in set(oa["elem"]):
        k = _lib.lib.arp_element_class(e)
        cov, vdw = C.c_double(), C.c_double()
        assert ob.lib().orc_radii(e, C.byref(cov), C.byref(vdw))
        assert prm.cov_radius[k] == cov.value and prm.vdw_radius[k] == vdw.value
        sel = oa["elem"] == e
        assert ((soa["attr"][sel] & 0xF) == k).all()
    return soa, oa


def test_ingest_matches_oracle_on_test_files(ubq_path, bft_path):
    for path in (ubq_path, bft_path):
        soa, oa = _compare_structure(aa.load_model(path), ob.Structure.load(path))
        assert (soa["attr"] & _lib.ATTR["LIGAND"]).all() and (soa["attr"] & _lib.ATTR["RECEPTOR"]).all()


def test_ingest_matches_oracle_on_stress_records(tmp_path):
    for kw in (dict(n_res=120, seed=3), dict(n_res=80, seed=4, n_models=2), dict(n_res=80, seed=5, altlocs=True)):
        rec = synth.gen_stress(**kw)
        prod = aa.Structure.from_records(rec)
        orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=False), flat=False)
        soa, oa = _compare_structure(prod, orc, "A,B/B,C,D")
        # hydrogen CSR and first CB / SG tables agree with a direct scan in hierarchy order
        assert len(soa["res_h_idx"]) == (oa["elem"] == b"H").sum()
        for r in range(len(soa["res_cb"])):
            hs = soa["res_h_idx"][soa["res_h_ptr"][r]:soa["res_h_ptr"][r + 1]]
            assert (oa["res_idx"][hs] == r).all() and (oa["elem"][hs] == b"H").all()
            for col, nm in (("res_cb", b"CB"), ("res_sg", b"SG")):
                idx = soa[col][r]
                members = np.flatnonzero((oa["res_idx"] == r) & (oa["name"] == nm))
                assert (idx == 0xFFFFFFFF) == (len(members) == 0)
                if len(members):
                    assert idx in members
        # file-level round trip through the PDB writer/reader (single-model, no altloc case only needs < 100k atoms)
        if kw.get("n_models", 1) == 1:
            p = tmp_path / "s.pdb"
            synth.write_pdb(rec, p)
            _compare_structure(aa.load_model(p), ob.Structure.load(p), "/")


def test_flat_records_path():
    rec = synth.gen_s2(5000, seed=11)
    prod = aa.Structure.from_records(rec, hierarchy=True)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    soa, oa = _compare_structure(prod, orc)
    assert np.array_equal(soa["res_ord"], 2 * np.arange(5000, dtype=np.uint32))
    rec = synth.gen_s1(3000, seed=12)
    prod = aa.Structure.from_records(rec, hierarchy=True)
    orc = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    _compare_structure(prod, orc)


def test_load_model_filters_non_protein(tmp_path, ubq_path):
    # utils.rs:60: only the 20 amino acids + HOH survive; ordinals are positions among the survivors
    lines = [l for l in open(ubq_path) if l.startswith(("ATOM", "HETATM"))]
    # turn residue 10 into a ligand-like residue name
    lines = [l[:17] + "LIG" + l[20:] if l[22:26].strip() == "10" and l.startswith("ATOM") else l for l in lines]
    p = tmp_path / "x.pdb"
    p.write_text("".join(lines))
    s, o = aa.load_model(p), ob.Structure.load(p)
    soa, oa = _compare_structure(s, o)
    assert b"LIG" not in set(s.strings("resn"))
    # residue 11 now directly follows residue 9
    resi = s.ints("resi")
    assert soa["res_ord"][resi == 11][0] == soa["res_ord"][resi == 9][0] + 1


def test_mmcif_reader(tmp_path, ubq_path):
    rec = synth.read_pdb_records(ubq_path)
    lines = ["data_test", "#", "loop_"]
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_seq_id",
            "pdbx_PDB_ins_code", "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "auth_seq_id", "auth_comp_id",
            "auth_asym_id", "auth_atom_id", "pdbx_PDB_model_num"]
    lines += [f"_atom_site.{c}" for c in cols]
    for k in range(len(rec["x"])):
        nm = rec["name"][k].decode()
        lines.append(" ".join([
            "HETATM" if rec["resn"][k] == b"HOH" else "ATOM", str(rec["serial"][k]), rec["element"][k].decode(), f'"{nm}"' if "'" in nm else nm, ".",
            rec["resn"][k].decode(), "A", str(rec["resi"][k]), "?", f"{rec['x'][k]:.3f}", f"{rec['y'][k]:.3f}", f"{rec['z'][k]:.3f}", "1.00", "10.00",
            str(rec["resi"][k]), rec["resn"][k].decode(), "A", nm, "1"]))
    lines.append("#")
    p = tmp_path / "x.cif"
    p.write_text("\n".join(lines) + "\n")
    a, b = aa.load_model(p), aa.load_model(ubq_path)
    sa, sb = a.soa(), b.soa()
    for k in ("x", "y", "z", "attr", "res_ord", "res_id", "chain_rank"):
        assert np.array_equal(sa[k], sb[k]), k
    assert (a.ints("model") == 1).all()


def test_mmcif_lexer_handles_quotes_text_fields_wrapped_rows_and_author_numbering(tmp_path):
    """utils.rs:53-57 reads mmCIF through pdbtbx: a real CIF lexer is needed (quoted values, semicolon text fields, rows wrapped over lines),
    and the chain / residue number come from the auth_* columns when present.  The same synthetic structure -- altlocs, insertion codes, two
    models -- written as PDB and as mmCIF must load into identical SoA columns and identity strings."""
    rec = synth.with_insertion_codes(synth.gen_stress(n_res=90, seed=41, n_models=2, altlocs=True, hydrogens=True))
    assert (rec["icode"] == b"A").sum() > 0 and (rec["altloc"] == b"B").sum() > 0
    pdb, cif, plain = tmp_path / "s.pdb", tmp_path / "s.cif", tmp_path / "plain.cif"
    synth.write_pdb(rec, pdb)
    synth.write_mmcif(rec, cif, fancy=True)
    synth.write_mmcif(rec, plain, fancy=False)
    a, b, c = aa.load_model(pdb), aa.load_model(cif), aa.load_model(plain)
    assert a.n_atoms == b.n_atoms == c.n_atoms == len(rec["x"])
    for other in (b, c):
        sa, sb = a.soa(), other.soa()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]), k
        for col in ("chain", "resn", "atomn", "insertion", "altloc", "element"):
            assert np.array_equal(a.strings(col), other.strings(col)), col
        for col in ("resi", "atomi", "model"):
            assert np.array_equal(a.ints(col), other.ints(col)), col
    assert set(a.strings("insertion")) == {b"", b"A"}


def test_cli_flags_and_defaults_match_the_reference():
    # src/cli/contacts.rs:9-52: -i -o required; -g "/", -f "contacts", -t csv, -c 0.1, -d 6.5, -j 1, --ignore-zero-occupancy false
    from arpeggia_amd.__main__ import FORMATS, build_parser

    a = build_parser().parse_args(["contacts", "-i", "m.pdb", "-o", "out"])
    assert (a.groups, a.filename, a.output_format, a.vdw_comp, a.dist_cutoff, a.num_threads, a.ignore_zero_occupancy) == ("/", "contacts", "csv", 0.1, 6.5, 1, False)
    b = build_parser().parse_args(["contacts", "--input", "m.cif", "--output", "o", "-g", "A,B/C", "-f", "x", "-t", "Parquet", "-c", "0.2", "-d", "5", "-j", "0",
                                   "--ignore-zero-occupancy"])
    assert (b.groups, b.filename, b.output_format, b.vdw_comp, b.dist_cutoff, b.num_threads, b.ignore_zero_occupancy) == ("A,B/C", "x", "parquet", 0.2, 5.0, 0, True)
    assert FORMATS == ("csv", "parquet", "json", "ndjson")  # utils.rs:159-167
    with pytest.raises(SystemExit):
        build_parser().parse_args(["contacts", "-o", "out"])


def test_cli_writers(tmp_path):
    import json

    import pyarrow as pa
    import pyarrow.parquet as pq
    from arpeggia_amd.__main__ import write_table

    t = pa.table({"model": pa.array([0, 0], pa.uint32()), "interaction": ["VanDerWaalsContact", "PolarContact"], "distance": pa.array([3.5, 2.9], pa.float32()),
                  "sc_dihedral": pa.array([12.5, None], pa.float32())})
    for fmt in ("csv", "parquet", "json", "ndjson"):
        write_table(t, tmp_path / f"c.{fmt}", fmt)
    assert (tmp_path / "c.csv").read_text().splitlines()[0].replace('"', "") == "model,interaction,distance,sc_dihedral"
    assert pq.read_table(tmp_path / "c.parquet").equals(t)
    assert json.loads((tmp_path / "c.json").read_text())[1]["sc_dihedral"] is None
    assert [json.loads(line)["interaction"] for line in (tmp_path / "c.ndjson").read_text().splitlines()] == ["VanDerWaalsContact", "PolarContact"]


def test_handwritten_mmcif_in_deposition_layout_loads_like_its_pdb_twin():
    """tests/data/hand7.cif is written by hand in the layout of a wwPDB deposition -- key-value categories, a semicolon text field that
    contains `loop_` / `data_` / `_atom_site.id` as plain text, a quoted value with an embedded quote, both quote kinds around atom names,
    `label_alt_id '.'`, `pdbx_PDB_ins_code '?'`, `label_seq_id '.'` on the waters, rows wrapped over two lines, a comment between two rows,
    `pdbx_PDB_model_num` 1 and 2, and auth_* != label_* (chain X vs A / B, residues 101-106A vs 1-7) -- next to its PDB-format twin
    hand7.pdb (MODEL / ENDMDL, altLoc, iCode, TER).  Nothing here comes from synth.write_mmcif.  Both must load to the same model, the one
    the oracle's independent PDB reader builds (load_model, src/utils.rs:51-63: pdbtbx picks the author chain and numbering)."""
    cif, pdb = aa.load_model(ROOT / "tests" / "data" / "hand7.cif"), aa.load_model(ROOT / "tests" / "data" / "hand7.pdb")
    assert cif.n_atoms == pdb.n_atoms == 126
    for col in ("chain", "resn", "atomn", "altloc", "insertion", "element"):
        assert np.array_equal(cif.strings(col), pdb.strings(col)), col
    for col in ("resi", "atomi", "model"):
        assert np.array_equal(cif.ints(col), pdb.ints(col)), col
    sa, sb = cif.soa("/"), pdb.soa("/")
    assert sorted(sa) == sorted(sb) and all(np.array_equal(sa[k], sb[k]) for k in sa)
    # what the file says, literally
    assert set(cif.strings("chain")) == {b"X"}                                  # auth_asym_id, not label_asym_id A / B
    resi, ins, alt, model = cif.ints("resi"), cif.strings("insertion"), cif.strings("altloc"), cif.ints("model")
    assert sorted(set(resi)) == [101, 102, 103, 104, 105, 106, 201, 202, 203]   # auth_seq_id, not label_seq_id 1..7
    thr = cif.strings("resn") == b"THR"
    assert (resi[thr] == 106).all() and (ins[thr] == b"A").all() and (ins[~thr] == b"").all()
    nz = (cif.strings("atomn") == b"NZ")
    assert sorted(alt[nz & (model == 1)]) == [b"A", b"B"] and (alt[~nz] == b"").all()
    assert sorted(set(model)) == [1, 2] and (model == 1).sum() == (model == 2).sum() == 63
    assert np.allclose(sa["y"][model == 2] - sa["y"][model == 1], 0.25)
    assert abs(sa["x"][0] - 27.340) < 1e-12 and abs(sa["z"][5] - 5.134) < 1e-12   # the wrapped row (atom 6) kept its coordinates
    # and the oracle's own reader agrees on the PDB twin: hierarchy, ordinals, attribute words
    _compare_structure(pdb, ob.Structure.load(str(ROOT / "tests" / "data" / "hand7.pdb")), "/")
    _compare_structure(cif, ob.Structure.load(str(ROOT / "tests" / "data" / "hand7.pdb")), "/")
