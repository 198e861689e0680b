"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (arpeggia_amd) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
LIB_PATH = ORACLE_DIR / "liboracle.so"

ATOM_DTYPE = np.dtype(
    [
        ("x", "<f8"), ("y", "<f8"), ("z", "<f8"), ("occ", "<f8"),
        ("serial", "<i4"), ("resi", "<i4"), ("model_serial", "<i4"), ("hetero", "<i4"),
        ("name", "S8"), ("resn", "S8"), ("chain", "S8"), ("altloc", "S4"), ("icode", "S4"), ("elem", "S4"),
        ("model_idx", "<i4"), ("chain_idx", "<i4"), ("res_idx", "<i4"), ("res_ord", "<i4"),
        ("res_resn", "S8"),
    ],
    align=True,
)
PAIR_DTYPE = np.dtype([("i", "<i4"), ("j", "<i4"), ("dist", "<f8"), ("kind", "<u4"), ("pad", "<u4")], align=True)
ENTITY_DTYPE = np.dtype(
    [("chain", "S8"), ("resn", "S8"), ("insertion", "S4"), ("altloc", "S4"), ("atomn", "S8"), ("resi", "<i4"), ("atomi", "<i4")],
    align=True,
)
ROW_DTYPE = np.dtype(
    [
        ("model", "<u4"), ("interaction", "<i4"), ("distance", "<f8"), ("from", ENTITY_DTYPE), ("to", ENTITY_DTYPE),
        ("has_sc", "<i4"), ("sc_centroid_dist", "<f8"), ("sc_dihedral", "<f8"), ("sc_centroid_angle", "<f8"),
        ("from_atom", "<i4"), ("to_atom", "<i4"),
    ],
    align=True,
)
PLANE_DTYPE = np.dtype(
    [
        ("c", "<f8", (3,)), ("n", "<f8", (3,)),
        ("model_serial", "<i4"), ("resi", "<i4"), ("res_idx", "<i4"), ("res_ord", "<i4"),
        ("chain", "S8"), ("resn", "S8"), ("insertion", "S4"), ("altloc", "S4"),
    ],
    align=True,
)

INTERACTIONS = [
    "StericClash", "CovalentBond", "Disulfide", "VanDerWaalsContact", "IonicBond", "HydrogenBond",
    "WeakHydrogenBond", "PolarContact", "WeakPolarContact", "IonicRepulsion", "SaltBridge",
    "PiDisplacedStacking", "PiTStacking", "PiSandwichStacking", "PiParallelInPlaneStacking",
    "PiTiltedStacking", "PiLStacking", "CationPi", "HydrophobicContact",
]

ORC_OK, ORC_ERR_IO, ORC_ERR_BAD_GROUPS, ORC_ERR_EMPTY_GROUPS, ORC_ERR_NO_RINGS, ORC_ERR_BAD_INPUT = range(6)


class OracleError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code


def build(force: bool = False) -> Path:
    """Compile the oracle with gcc (seconds)."""
    src = ORACLE_DIR / "arp_oracle.c"
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, (ORACLE_DIR / "arp_oracle.h").stat().st_mtime):
        subprocess.run(["make", "-C", str(ORACLE_DIR), "liboracle.so"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(LIB_PATH))
        L.orc_last_error.restype = C.c_char_p
        L.orc_interaction_name.restype = C.c_char_p
        L.orc_load_model.restype = C.c_void_p
        L.orc_load_model.argtypes = [C.c_char_p, C.c_int]
        L.orc_from_atoms.restype = C.c_void_p
        L.orc_from_atoms.argtypes = [C.c_void_p, C.c_int32, C.c_int]
        L.orc_free_structure.argtypes = [C.c_void_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_n_atoms.argtypes = [C.c_void_p]
        L.orc_n_atoms.restype = C.c_int32
        L.orc_atoms.argtypes = [C.c_void_p]
        L.orc_atoms.restype = C.c_void_p
        L.orc_atomic_contacts.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_get_contacts.argtypes = [C.c_void_p, C.c_char_p, C.c_double, C.c_double, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
        L.orc_planes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int32)]
        L.orc_atom_classes.argtypes = [C.c_void_p]
        L.orc_atom_classes.restype = C.c_uint32
        L.orc_angle.restype = C.c_double
        L.orc_angle.argtypes = [C.POINTER(C.c_double)] * 3
        L.orc_dihedral.restype = C.c_double
        L.orc_dihedral.argtypes = [C.POINTER(C.c_double)] * 4
        L.orc_radii.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        assert L.orc_sizeof_atom() == ATOM_DTYPE.itemsize, (L.orc_sizeof_atom(), ATOM_DTYPE.itemsize)
        assert L.orc_sizeof_pair() == PAIR_DTYPE.itemsize
        assert L.orc_sizeof_row() == ROW_DTYPE.itemsize, (L.orc_sizeof_row(), ROW_DTYPE.itemsize)
        assert L.orc_sizeof_plane() == PLANE_DTYPE.itemsize, (L.orc_sizeof_plane(), PLANE_DTYPE.itemsize)
        _lib = L
    return _lib


def _copy_out(ptr, n, dtype):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (n * dtype.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


class Structure:
    """Owns an OrcStructure*."""

    def __init__(self, handle):
        if not handle:
            raise OracleError(-1, lib().orc_last_error().decode())
        self._h = handle

    @classmethod
    def load(cls, path, ignore_zero_occupancy=False):
        return cls(lib().orc_load_model(os.fsencode(str(path)), int(ignore_zero_occupancy)))

    @classmethod
    def from_atoms(cls, atoms: np.ndarray, flat: bool):
        atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
        return cls(lib().orc_from_atoms(atoms.ctypes.data, len(atoms), int(flat)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_free_structure(self._h)
            self._h = None

    @property
    def atoms(self) -> np.ndarray:
        n = lib().orc_n_atoms(self._h)
        return _copy_out(lib().orc_atoms(self._h), n, ATOM_DTYPE)

    def _check(self, rc):
        if rc != ORC_OK:
            raise OracleError(rc, lib().orc_last_error().decode())

    def atomic_contacts(self, groups="/", vdw_comp=0.1, dist_cutoff=6.5, brute=False) -> np.ndarray:
        p, n = C.c_void_p(), C.c_int64()
        self._check(lib().orc_atomic_contacts(self._h, groups.encode(), vdw_comp, dist_cutoff, int(brute), C.byref(p), C.byref(n)))
        out = _copy_out(p.value, n.value, PAIR_DTYPE)
        lib().orc_free(p)
        return out

    def get_contacts(self, groups="/", vdw_comp=0.1, dist_cutoff=6.5) -> np.ndarray:
        p, n = C.c_void_p(), C.c_int64()
        self._check(lib().orc_get_contacts(self._h, groups.encode(), vdw_comp, dist_cutoff, C.byref(p), C.byref(n)))
        out = _copy_out(p.value, n.value, ROW_DTYPE)
        lib().orc_free(p)
        return out

    def planes(self, which: str) -> np.ndarray:
        p, n = C.c_void_p(), C.c_int32()
        self._check(lib().orc_planes(self._h, {"ring": 0, "sc": 1}[which], C.byref(p), C.byref(n)))
        out = _copy_out(p.value, n.value, PLANE_DTYPE)
        lib().orc_free(p)
        return out


def sap_weight(resn: str, sasa: float) -> float:
    L = lib()
    L.orc_sap_weight.restype = C.c_float
    L.orc_sap_weight.argtypes = [C.c_char_p, C.c_float]
    return float(L.orc_sap_weight(resn.encode(), C.c_float(sasa)))


def sap_neighbor_sum(x, y, z, sidechain, weight, sap_radius: float) -> np.ndarray:
    """src/sap.rs:155-204 restated (brute force, index order, f32 accumulation)."""
    L = lib()
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    L.orc_sap_neighbor_sum.restype = None
    L.orc_sap_neighbor_sum.argtypes = [C.c_int64, dp, dp, dp, C.POINTER(C.c_uint8), fp, C.c_float, fp]
    x, y, z = (np.ascontiguousarray(v, dtype="<f8") for v in (x, y, z))
    m = np.ascontiguousarray(sidechain, dtype=np.uint8)
    w = np.ascontiguousarray(weight, dtype="<f4")
    out = np.zeros(len(x), dtype="<f4")
    L.orc_sap_neighbor_sum(len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp), z.ctypes.data_as(dp), m.ctypes.data_as(C.POINTER(C.c_uint8)), w.ctypes.data_as(fp),
                           C.c_float(sap_radius), out.ctypes.data_as(fp))
    return out


def atom_classes(atoms: np.ndarray) -> np.ndarray:
    atoms = np.ascontiguousarray(atoms, dtype=ATOM_DTYPE)
    base = atoms.ctypes.data
    L = lib()
    return np.array([L.orc_atom_classes(base + k * ATOM_DTYPE.itemsize) for k in range(len(atoms))], dtype=np.uint32)


def parse_groups(all_chains, groups):
    L = lib()
    arr = (C.c_char_p * len(all_chains))(*[c.encode() for c in all_chains])
    lig = C.create_string_buffer(4096)
    rec = C.create_string_buffer(4096)
    nl, nr = C.c_int(), C.c_int()
    rc = L.orc_parse_groups(arr, len(all_chains), groups.encode(), lig, 4096, C.byref(nl), rec, 4096, C.byref(nr))
    if rc != ORC_OK:
        raise OracleError(rc, L.orc_last_error().decode())

    def unpack(buf, n):
        return set(x.decode() for x in buf.raw.split(b"\0")[:n])

    return unpack(lig, nl.value), unpack(rec, nr.value)


def angle(a, b, c):
    arr = [(C.c_double * 3)(*v) for v in (a, b, c)]
    return lib().orc_angle(*arr)


def dihedral(a, b, c, d):
    arr = [(C.c_double * 3)(*v) for v in (a, b, c, d)]
    return lib().orc_dihedral(*arr)


def plane_metrics(c1, n1, c2, n2):
    arr = [(C.c_double * 3)(*v) for v in (c1, n1, c2, n2)]
    out = (C.c_double * 3)()
    lib().orc_plane_metrics(*arr, out)
    return tuple(out)


def rows_to_csv_lines(rows: np.ndarray):
    """Same text form as oracle/orc_dump (distance and sc_* narrowed to f32 like the reference table)."""
    out = []
    for r in rows:
        f, t = r["from"], r["to"]
        d = lambda b: b.decode()
        g = lambda v: repr(float(np.float32(v)))
        sc = (g(r["sc_centroid_dist"]), g(r["sc_dihedral"]), g(r["sc_centroid_angle"])) if r["has_sc"] else ("", "", "")
        out.append(
            ",".join(
                [
                    str(r["model"]), INTERACTIONS[r["interaction"]], g(r["distance"]),
                    d(f["chain"]), d(f["resn"]), str(f["resi"]), d(f["insertion"]), d(f["altloc"]), d(f["atomn"]), str(f["atomi"]),
                    d(t["chain"]), d(t["resn"]), str(t["resi"]), d(t["insertion"]), d(t["altloc"]), d(t["atomn"]), str(t["atomi"]),
                    *sc,
                ]
            )
        )
    return out
