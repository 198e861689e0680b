"""Synthetic inputs for parity tests and bench.py.  numpy/scipy only; no oracle, no product imports.

A "records" dict is the interchange format: per-atom columns that both the product (arpeggia_amd.Structure.from_records)
and the oracle (tests/oracle_binding via records_to_oracle) accept.

  gen_s2      kernel stress at protein-interior density: jittered lattice, rho = 0.05 / A^3, atom kinds drawn i.i.d. from
              6bft's atoms, one chain, every atom its own residue with res_ord = 2 * index (nothing is sequence-adjacent).
  gen_s1      chemistry-faithful: rigid copies of 1ubq on an fcc lattice (30 A spacing), one chain per copy.
  gen_stress  rule coverage: whole 6bft residues thrown at random into a small box with hydrogens added to donors, so
              clashes, covalent/disulfide bands, hydrogen bonds, salt bridges and weak hydrogen bonds all occur.
"""
from __future__ import annotations

from pathlib import Path

import numpy as np

DATA = Path(__file__).resolve().parent / "data"
RHO = 0.050  # atoms / A^3 (protein interiors 0.043-0.057)


def read_pdb_records(path) -> dict:
    """Plain fixed-column reader for the two test files (no altlocs, no MODEL records)."""
    cols = {k: [] for k in ("x", "y", "z", "occupancy", "serial", "resi", "name", "resn", "chain", "altloc", "icode", "element", "model_serial")}
    for line in Path(path).read_text().splitlines():
        if not (line.startswith("ATOM  ") or line.startswith("HETATM")):
            continue
        cols["serial"].append(int(line[6:11]))
        cols["name"].append(line[12:16].strip().upper())
        cols["altloc"].append(line[16:17].strip())
        cols["resn"].append(line[17:20].strip().upper())
        cols["chain"].append(line[21:22].strip())
        cols["resi"].append(int(line[22:26]))
        cols["icode"].append(line[26:27].strip())
        cols["x"].append(float(line[30:38])); cols["y"].append(float(line[38:46])); cols["z"].append(float(line[46:54]))
        cols["occupancy"].append(float(line[54:60]) if line[54:60].strip() else 1.0)
        cols["element"].append(line[76:78].strip().upper())
        cols["model_serial"].append(0)
    return _finish(cols)


def _finish(cols: dict) -> dict:
    dt = {"x": "<f8", "y": "<f8", "z": "<f8", "occupancy": "<f8", "serial": "<i4", "resi": "<i4", "model_serial": "<i4",
          "name": "S8", "resn": "S8", "chain": "S8", "altloc": "S4", "icode": "S4", "element": "S4", "res_ord": "<u4", "res_id": "<u4"}
    out = {}
    for k, v in cols.items():
        if k in ("name", "resn", "chain", "altloc", "icode", "element") and len(v) and isinstance(v[0], str):
            v = [s.encode() for s in v]
        out[k] = np.asarray(v, dtype=dt[k])
    return out


STANDARD = {b"ALA", b"ARG", b"ASN", b"ASP", b"CYS", b"GLN", b"GLU", b"GLY", b"HIS", b"ILE", b"LEU", b"LYS", b"MET", b"PHE", b"PRO",
            b"SER", b"THR", b"TRP", b"TYR", b"VAL", b"HOH"}


def records_to_oracle(rec: dict, flat: bool):
    """records -> OrcAtom array (tests/oracle_binding.ATOM_DTYPE).  flat=True needs res_ord/res_id in rec."""
    import oracle_binding as ob

    n = len(rec["x"])
    a = np.zeros(n, dtype=ob.ATOM_DTYPE)
    a["x"], a["y"], a["z"] = rec["x"], rec["y"], rec["z"]
    a["occ"] = rec.get("occupancy", np.ones(n))
    a["serial"], a["resi"] = rec["serial"], rec["resi"]
    a["model_serial"] = rec.get("model_serial", np.zeros(n, dtype=np.int32))
    a["name"], a["resn"], a["chain"], a["elem"] = rec["name"], rec["resn"], rec["chain"], rec["element"]
    if "altloc" in rec:
        a["altloc"] = rec["altloc"]
    if "icode" in rec:
        a["icode"] = rec["icode"]
    if flat:
        a["res_ord"], a["res_idx"] = rec["res_ord"], rec["res_id"]
        a["res_resn"] = rec["resn"]
        ms = a["model_serial"]
        change = np.concatenate([[0], (ms[1:] != ms[:-1]).astype(np.int32)])
        a["model_idx"] = np.cumsum(change)
        a["chain_idx"] = _chain_index(a)
    return a


def _chain_index(a):
    keys = np.char.add(np.char.add(a["model_idx"].astype("U12"), "|"), a["chain"].astype("U8"))
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(np.argsort(first))  # order of first appearance
    return order[inv].astype(np.int32)


def _random_rotations(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1),
    ], 1)


# ------------------------------------------------------------------------------------------------ S2
def gen_s2(n_atoms: int, seed: int = 0xA11CE5EED00 + 4, rho: float = RHO, jitter: float = 0.15) -> dict:
    rng = np.random.default_rng(seed)
    a = rho ** (-1.0 / 3.0)
    radius = (3.0 * n_atoms / (4.0 * np.pi * rho)) ** (1.0 / 3.0)
    m = int(np.ceil(radius / a)) + 2
    g = np.arange(-m, m + 1, dtype=np.float64) * a
    # lattice points inside a slightly larger ball, keep the n_atoms closest to the centre
    gx, gy, gz = np.meshgrid(g, g, g, indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1)
    r2 = (pts ** 2).sum(1)
    keep = np.argsort(r2, kind="stable")[:n_atoms]
    keep.sort()
    pts = pts[keep] + rng.uniform(-jitter, jitter, size=(n_atoms, 3))  # min separation a - 2*jitter*sqrt(1) >= 2.41 A
    pts = np.round(pts + 500.0, 3)  # positive, PDB-like 3 decimals
    tmpl = read_pdb_records(DATA / "6bft.pdb")
    kind = rng.integers(0, len(tmpl["x"]), size=n_atoms)
    idx = np.arange(n_atoms)
    return {
        "x": pts[:, 0].copy(), "y": pts[:, 1].copy(), "z": pts[:, 2].copy(), "occupancy": np.ones(n_atoms),
        "serial": (idx + 1).astype(np.int32), "resi": (idx + 1).astype(np.int32), "model_serial": np.zeros(n_atoms, dtype=np.int32),
        "name": tmpl["name"][kind], "resn": tmpl["resn"][kind], "chain": np.full(n_atoms, b"A", dtype="S8"),
        "altloc": np.zeros(n_atoms, dtype="S4"), "icode": np.zeros(n_atoms, dtype="S4"), "element": tmpl["element"][kind],
        "res_ord": (2 * idx).astype(np.uint32), "res_id": idx.astype(np.uint32),
    }


# ------------------------------------------------------------------------------------------------ S1
def _fcc_sites(n_sites: int, spacing: float):
    a = spacing * np.sqrt(2.0)  # conventional cell edge for nearest-neighbour distance `spacing`
    m = int(np.ceil((n_sites / 4.0) ** (1.0 / 3.0))) + 2
    base = np.array([[0, 0, 0], [0.5, 0.5, 0], [0.5, 0, 0.5], [0, 0.5, 0.5]])
    g = np.arange(-m, m + 1)
    cells = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    sites = (cells[:, None, :] + base[None, :, :]).reshape(-1, 3) * a
    order = np.argsort((sites ** 2).sum(1), kind="stable")
    return sites[order[:n_sites]]


def gen_s1(n_atoms: int, seed: int = 0xA11CE5EED00 + 3, spacing: float = 30.0, clearance: float = 2.2, tries: int = 12) -> dict:
    from scipy.spatial import cKDTree

    rng = np.random.default_rng(seed)
    tmpl = read_pdb_records(DATA / "1ubq.pdb")
    nt = len(tmpl["x"])
    txyz = np.stack([tmpl["x"], tmpl["y"], tmpl["z"]], 1)
    txyz = txyz - txyz.mean(0)
    # residue ordinals of the template (order of appearance; waters follow the protein in chain A)
    rkey = tmpl["resi"].astype(np.int64) * 2 + (tmpl["resn"] == b"HOH")
    _, first, inv = np.unique(rkey, return_index=True, return_inverse=True)
    t_ord = np.argsort(np.argsort(first))[inv]
    n_res_t = int(t_ord.max()) + 1
    n_copies = -(-n_atoms // nt)
    sites = _fcc_sites(n_copies, spacing)
    site_tree = cKDTree(sites)
    placed = [None] * n_copies
    for k in range(n_copies):
        neigh = [j for j in site_tree.query_ball_point(sites[k], spacing * 1.5) if j < k and placed[j] is not None]
        others = np.concatenate([placed[j] for j in neigh]) if neigh else None
        best, best_bad = None, None
        for _ in range(tries):
            R = _random_rotations(rng, 1)[0]
            xyz = txyz @ R.T + sites[k] + rng.uniform(-1.5, 1.5, size=3)
            if others is None:
                best = xyz
                break
            d, _ = cKDTree(xyz).query(others, k=1, distance_upper_bound=clearance)
            bad = int(np.isfinite(d).sum())
            if best is None or bad < best_bad:
                best, best_bad = xyz, bad
            if bad == 0:
                break
        placed[k] = best
    xyz = np.concatenate(placed)
    copy = np.repeat(np.arange(n_copies), nt)
    # truncate the last copy at a residue boundary
    n_keep = n_atoms
    t_ord_all = np.tile(t_ord, n_copies)
    while n_keep < len(xyz) and n_keep > 0 and copy[n_keep] == copy[n_keep - 1] and t_ord_all[n_keep] == t_ord_all[n_keep - 1]:
        n_keep -= 1
    sl = slice(0, n_keep)
    xyz = np.round(xyz[sl] - xyz[sl].min(0) + 10.0, 3)
    copy = copy[sl]
    tile = lambda v: np.tile(v, n_copies)[sl]
    chains = np.array([f"C{c:06d}".encode() for c in range(n_copies)], dtype="S8")[copy]
    return {
        "x": xyz[:, 0].copy(), "y": xyz[:, 1].copy(), "z": xyz[:, 2].copy(), "occupancy": np.ones(n_keep),
        "serial": np.arange(1, n_keep + 1, dtype=np.int32), "resi": tile(tmpl["resi"]), "model_serial": np.zeros(n_keep, dtype=np.int32),
        "name": tile(tmpl["name"]), "resn": tile(tmpl["resn"]), "chain": chains,
        "altloc": np.zeros(n_keep, dtype="S4"), "icode": np.zeros(n_keep, dtype="S4"), "element": tile(tmpl["element"]),
        "res_ord": t_ord_all[sl].astype(np.uint32), "res_id": (copy * n_res_t + t_ord_all[sl]).astype(np.uint32),
    }


# ------------------------------------------------------------------------------------------------ stress
def gen_many_chains(n_chains: int = 70000, seed: int = 31, spacing: float = 5.8, max_atoms: int = 6) -> dict:
    """One residue per chain, more chains than a 16-bit rank holds (arpeggia_amd.h API v2; the reference keys on the chain id string,
    complex.rs:19-21): residues of 1ubq (waters excluded) at random orientations on a jittered cubic lattice, every one with a chain id of
    its own -- ids that do NOT sort in file order, so that rank and file order differ.  Hierarchy-free records (the builders derive it)."""
    rng = np.random.default_rng(seed)
    tmpl = read_pdb_records(DATA / "1ubq.pdb")
    keep = tmpl["resn"] != b"HOH"
    tmpl = {k: v[keep] for k, v in tmpl.items()}
    resi = tmpl["resi"]
    starts = np.flatnonzero(np.concatenate([[True], resi[1:] != resi[:-1]]))
    ends = np.concatenate([starts[1:], [len(resi)]])
    small = (ends - starts) <= max_atoms  # (the oracle compares chain id strings per pair: keep the pair count of the test in check)
    phe = int(np.flatnonzero(tmpl["resn"][starts] == b"PHE")[0])  # ... plus ONE aromatic residue: without any ring the table path errors
    small[phe] = True                                              # like the reference panics (complex.rs:50,480-482)
    phe_t = int(small[:phe].sum())
    starts, ends = starts[small], ends[small]
    n_t = len(starts)
    side = int(np.ceil(n_chains ** (1.0 / 3.0)))
    g = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 3)[:n_chains]
    centre = g * spacing + rng.uniform(-0.8, 0.8, size=(n_chains, 3))
    which = rng.integers(0, n_t, n_chains)
    which[which == phe_t] = (phe_t + 1) % n_t
    which[0] = phe_t  # (chain 0 is the one phenylalanine)
    rots = _random_rotations(rng, n_chains)
    cols = {k: [] for k in ("x", "y", "z", "occupancy", "serial", "resi", "name", "resn", "chain", "altloc", "icode", "element", "model_serial")}
    ids = rng.permutation(n_chains)  # chain c is called K<ids[c]>: byte-wise order != file order
    xyz_all, meta = [], []
    for t in range(n_t):  # vectorised per template residue
        sel = np.flatnonzero(which == t)
        if not len(sel):
            continue
        sl = slice(starts[t], ends[t])
        base = np.stack([tmpl["x"][sl], tmpl["y"][sl], tmpl["z"][sl]], 1)
        base = base - base.mean(0)
        pos = np.einsum("cij,aj->cai", rots[sel], base) + centre[sel][:, None, :]
        xyz_all.append((sel, pos, sl))
    order = []  # file order = chain order, atoms of a residue together
    per_chain = {}
    for sel, pos, sl in xyz_all:
        for k, c in enumerate(sel):
            per_chain[int(c)] = (pos[k], sl)
    serial = 1
    for c in range(n_chains):
        pos, sl = per_chain[c]
        na = pos.shape[0]
        cols["x"].extend(np.round(pos[:, 0], 3)); cols["y"].extend(np.round(pos[:, 1], 3)); cols["z"].extend(np.round(pos[:, 2], 3))
        cols["occupancy"].extend([1.0] * na); cols["serial"].extend(range(serial, serial + na)); serial += na
        cols["resi"].extend([1] * na); cols["name"].extend(tmpl["name"][sl]); cols["resn"].extend(tmpl["resn"][sl])
        cols["chain"].extend([b"K%06d" % int(ids[c])] * na); cols["altloc"].extend([b""] * na); cols["icode"].extend([b""] * na)
        cols["element"].extend(tmpl["element"][sl]); cols["model_serial"].extend([0] * na)
    return _finish(cols)


def gen_stress(n_res: int = 400, seed: int = 7, box: float = 28.0, hydrogens: bool = True, n_models: int = 1, n_chains: int = 4,
               altlocs: bool = False) -> dict:
    """Whole residues of 6bft at random poses in a small box: overlaps give clashes / covalent-band pairs, CYS pairs give
    disulfide candidates, and explicit hydrogens (1.0 A from each N/O/S donor and carbon, random direction) exercise the
    angle-dependent branches.  Goes through the pdbtbx-like hierarchy builder (no res_ord / res_id columns)."""
    rng = np.random.default_rng(seed)
    tmpl = read_pdb_records(DATA / "6bft.pdb")
    key = np.char.add(np.char.add(tmpl["chain"].astype("U8"), "|"), tmpl["resi"].astype("U12"))
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    res_of = np.argsort(np.argsort(first))[inv]
    n_t = int(res_of.max()) + 1
    starts = np.searchsorted(res_of, np.arange(n_t))
    ends = np.searchsorted(res_of, np.arange(n_t), side="right")
    is_cys = np.array([tmpl["resn"][starts[r]] == b"CYS" for r in range(n_t)])
    cols = {k: [] for k in ("x", "y", "z", "occupancy", "serial", "resi", "name", "resn", "chain", "altloc", "icode", "element", "model_serial")}
    serial = 1
    chain_names = [chr(ord("A") + c) for c in range(n_chains)]
    for m in range(n_models):
        per_chain_resi = [0] * n_chains
        for r in range(n_res):
            # every 6th residue is a cysteine so that SG..SG pairs in the covalent band exist
            cand = np.flatnonzero(is_cys) if r % 6 == 0 else np.arange(n_t)
            t = int(rng.choice(cand))
            sl = slice(starts[t], ends[t])
            xyz = np.stack([tmpl["x"][sl], tmpl["y"][sl], tmpl["z"][sl]], 1)
            xyz = (xyz - xyz.mean(0)) @ _random_rotations(rng, 1)[0].T + rng.uniform(0, box, size=3) + 100.0 * np.array([1, 2, 3])
            if r % 6 == 0 and r >= 6 and rng.random() < 0.7:
                # drag this cysteine's SG next to the previous cysteine's SG (2.0-2.2 A) -> covalent band, both dihedral outcomes
                prev = getattr(gen_stress, "_last_sg", None)
                names = tmpl["name"][sl]
                if prev is not None and (names == b"SG").any():
                    sg = xyz[np.flatnonzero(names == b"SG")[0]]
                    v = rng.normal(size=3); v /= np.linalg.norm(v)
                    xyz = xyz + (prev + v * rng.uniform(2.0, 2.2) - sg)
            c = int(rng.integers(0, n_chains))
            per_chain_resi[c] += 1
            names = tmpl["name"][sl]
            if (names == b"SG").any() and tmpl["resn"][starts[t]] == b"CYS":
                gen_stress._last_sg = xyz[np.flatnonzero(names == b"SG")[0]].copy()
            alt = ""
            for k in range(xyz.shape[0]):
                el = tmpl["element"][sl][k]
                this_alt = alt
                if altlocs and names[k] not in (b"N", b"CA", b"C", b"O") and r % 5 == 1:
                    this_alt = "A"
                for rep in range(2 if (this_alt == "A") else 1):
                    p = xyz[k] + (0.0 if rep == 0 else rng.normal(scale=0.4, size=3))
                    cols["x"].append(round(float(p[0]), 3)); cols["y"].append(round(float(p[1]), 3)); cols["z"].append(round(float(p[2]), 3))
                    cols["occupancy"].append(1.0 if rep == 0 else 0.5); cols["serial"].append(serial); serial += 1
                    cols["resi"].append(per_chain_resi[c]); cols["name"].append(names[k].decode()); cols["resn"].append(tmpl["resn"][sl][k].decode())
                    cols["chain"].append(chain_names[c]); cols["altloc"].append(this_alt if rep == 0 else "B"); cols["icode"].append("")
                    cols["element"].append(el.decode()); cols["model_serial"].append(m + 1 if n_models > 1 else 0)
                if hydrogens and el in (b"N", b"O", b"S", b"C"):
                    nh = 1 if el != b"C" else int(rng.integers(0, 2))
                    for h in range(nh):
                        v = rng.normal(size=3); v /= np.linalg.norm(v)
                        p = xyz[k] + v * 1.0
                        cols["x"].append(round(float(p[0]), 3)); cols["y"].append(round(float(p[1]), 3)); cols["z"].append(round(float(p[2]), 3))
                        cols["occupancy"].append(1.0); cols["serial"].append(serial); serial += 1
                        cols["resi"].append(per_chain_resi[c]); cols["name"].append(f"H{names[k].decode()[:2]}{h}"); cols["resn"].append(tmpl["resn"][sl][k].decode())
                        cols["chain"].append(chain_names[c]); cols["altloc"].append(""); cols["icode"].append("")
                        cols["element"].append("H"); cols["model_serial"].append(m + 1 if n_models > 1 else 0)
    gen_stress._last_sg = None
    return _finish(cols)


def write_pdb(rec: dict, path):
    """Minimal PDB writer (< 100k atoms) so that file-level entry points can be exercised on synthetic records."""
    lines = []
    last_model = None
    for k in range(len(rec["x"])):
        ms = int(rec["model_serial"][k])
        if ms != 0 and ms != last_model:
            if last_model is not None:
                lines.append("ENDMDL")
            lines.append(f"MODEL     {ms:4d}")
            last_model = ms
        name = rec["name"][k].decode()
        nm = f" {name:<3s}" if len(name) < 4 else name
        lines.append(
            f"ATOM  {int(rec['serial'][k]) % 100000:5d} {nm}{rec['altloc'][k].decode() or ' ':1s}{rec['resn'][k].decode():>3s} "
            f"{rec['chain'][k].decode():1s}{int(rec['resi'][k]):4d}{rec['icode'][k].decode() or ' ':1s}   "
            f"{rec['x'][k]:8.3f}{rec['y'][k]:8.3f}{rec['z'][k]:8.3f}{rec['occupancy'][k]:6.2f}{0.0:6.2f}          {rec['element'][k].decode():>2s}"
        )
    if last_model is not None:
        lines.append("ENDMDL")
    lines.append("END")
    Path(path).write_text("\n".join(lines) + "\n")


def with_insertion_codes(rec: dict, every: int = 23) -> dict:
    """Insertion codes the way antibody numbering produces them: every `every`-th residue (in file order, per chain) takes the residue
    NUMBER of its predecessor and the code 'A' (100, 100A).  Returns a copy."""
    out = {k: v.copy() for k, v in rec.items()}
    key = np.char.add(np.char.add(np.char.add(out["model_serial"].astype("U12"), "|"), out["chain"].astype("U8")), np.char.add("|", out["resi"].astype("U12")))
    _, first, inv = np.unique(key, return_index=True, return_inverse=True)
    res_of = np.argsort(np.argsort(first))[inv]
    prev_resi = {}
    for r in range(1, int(res_of.max()) + 1):
        if r % every:
            continue
        cur, prv = np.flatnonzero(res_of == r), np.flatnonzero(res_of == r - 1)
        if out["chain"][cur[0]] != out["chain"][prv[0]] or out["model_serial"][cur[0]] != out["model_serial"][prv[0]] or out["icode"][prv[0]] != b"":
            continue
        out["resi"][cur] = out["resi"][prv[0]]
        out["icode"][cur] = b"A"
    return out


def write_mmcif(rec: dict, path, fancy: bool = True):
    """mmCIF writer for the ingest tests.  `fancy` uses what a minimal line-based reader cannot take: a multi-line semicolon text field,
    quoted values, rows wrapped over two lines, label_* columns that DIFFER from the author numbering (label_seq_id counts 1, 2, ...;
    label_asym_id is a letter per chain in order of appearance) next to the auth_* columns PDB files carry, '?' and '.' for absent values."""
    lines = ["data_synth", "#", "_entry.id synth", "_struct.title", ";A synthetic structure", "with a title over two lines", ";", "#",
             "loop_", "_entity.id", "_entity.pdbx_description", "1 'first entity'", '2 "second entity, isn\'t it"', "#", "loop_"]
    cols = ["group_PDB", "id", "type_symbol", "label_atom_id", "label_alt_id", "label_comp_id", "label_asym_id", "label_seq_id", "pdbx_PDB_ins_code",
            "Cartn_x", "Cartn_y", "Cartn_z", "occupancy", "B_iso_or_equiv", "auth_seq_id", "auth_asym_id", "pdbx_PDB_model_num"]
    lines += [f"_atom_site.{c}" for c in cols]
    label_asym, label_seq, last = {}, {}, None
    for k in range(len(rec["x"])):
        ch = (int(rec["model_serial"][k]), rec["chain"][k])
        if ch not in label_asym:
            label_asym[ch] = "L" + chr(ord("A") + len(label_asym) % 26) if fancy else rec["chain"][k].decode()
        rkey = (ch, int(rec["resi"][k]), rec["icode"][k])
        if rkey != last:
            label_seq[ch] = label_seq.get(ch, 0) + 1
            last = rkey
        nm = rec["name"][k].decode()
        name = f'"{nm}"' if fancy and k % 3 == 0 else (f"'{nm}'" if fancy and k % 3 == 1 else nm)
        f = ["ATOM", str(int(rec["serial"][k])), rec["element"][k].decode(), name, rec["altloc"][k].decode() or ".", rec["resn"][k].decode(), label_asym[ch],
             str(label_seq[ch]) if fancy else str(int(rec["resi"][k])), rec["icode"][k].decode() or "?",
             f"{rec['x'][k]:.3f}", f"{rec['y'][k]:.3f}", f"{rec['z'][k]:.3f}", f"{rec['occupancy'][k]:.2f}", "10.00", str(int(rec["resi"][k])), rec["chain"][k].decode(),
             str(int(rec["model_serial"][k]) or 1)]
        if fancy and k % 5 == 4:
            lines += [" ".join(f[:9]), "   " + " ".join(f[9:])]
        else:
            lines.append(" ".join(f))
    lines.append("#")
    Path(path).write_text("\n".join(lines) + "\n")
