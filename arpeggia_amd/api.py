"""Python surface of the MI355X-native contact engine.

Mirrors the reference's PyO3 module for the one path this repo replaces:
`arpeggia.contacts(input_file, groups="/", vdw_comp=0.1, dist_cutoff=6.5, ignore_zero_occupancy=False, num_threads=1)`
(reference: src/python.rs:31-56, stubs python/arpeggia/arpeggia.pyi:7-33) plus the Rust-level pieces it is made of
(`load_model` utils.rs:51, `parse_groups` utils.rs:71, `get_contacts` contacts/mod.rs:61).

Everything numeric happens in libarpeggia_amd.so on a gfx950 device; this file only marshals pointers.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import lib

PAIR_DTYPE = np.dtype([("i", "<u4"), ("j", "<u4"), ("dist", "<f4"), ("kind", "<u4")])

TABLE_COLUMNS = [  # mod.rs:140-181 + 209-211, in the reference's order with its dtypes
    ("model", "u4"), ("interaction", "str"), ("distance", "f4"),
    ("from_chain", "str"), ("from_resn", "str"), ("from_resi", "i4"), ("from_insertion", "str"), ("from_altloc", "str"),
    ("from_atomn", "str"), ("from_atomi", "i4"),
    ("to_chain", "str"), ("to_resn", "str"), ("to_resi", "i4"), ("to_insertion", "str"), ("to_altloc", "str"),
    ("to_atomn", "str"), ("to_atomi", "i4"),
    ("sc_centroid_dist", "f4"), ("sc_dihedral", "f4"), ("sc_centroid_angle", "f4"),
]


class ArpeggiaError(RuntimeError):
    """Raised where the reference panics (pyo3_runtime.PanicException) or a HIP call fails."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status = status


def _check(status: int):
    if status != _lib.ARP_OK:
        msg = lib.arp_last_error().decode() or lib.arp_strerror(status).decode()
        raise ArpeggiaError(status, msg)


def device_count() -> int:
    return int(lib.arp_device_count())


def default_params(vdw_comp: float = 0.1, dist_cutoff: float = 6.5, deterministic: bool = False, contacts_only: bool = False,
                   no_speculation: bool = False, residue_runs: bool | None = None) -> _lib.arp_params:
    """deterministic=True selects the two-pass ordered emitter (ARP_FLAG_DETERMINISTIC): identical bytes run to run.
    contacts_only=True drops candidates without any interaction on the device (ARP_FLAG_CONTACTS_ONLY).
    no_speculation=True (ARP_FLAG_NO_SPECULATION): Context.enqueue always launches the probe pass, so that work ordered on the stream behind
    the enqueue sees final records (include/arpeggia_amd.h).
    residue_runs: a hint about the input, never a change of the result -- True asks for the kernels that apply the reference's residue rule
    (complex.rs:108-113) before the exact phase (ARP_FLAG_RESIDUE_RUNS), False rules them out (ARP_FLAG_NO_RESIDUE_RUNS), None lets the
    engine choose from a sample of the previous call's atoms."""
    p = _lib.arp_params()
    lib.arp_default_params(C.byref(p))
    p.vdw_comp = vdw_comp
    p.dist_cutoff = dist_cutoff
    p.flags = ((_lib.ARP_FLAG_DETERMINISTIC if deterministic else 0) | (_lib.ARP_FLAG_CONTACTS_ONLY if contacts_only else 0)
               | (_lib.ARP_FLAG_NO_SPECULATION if no_speculation else 0)
               | (0 if residue_runs is None else (_lib.ARP_FLAG_RESIDUE_RUNS if residue_runs else _lib.ARP_FLAG_NO_RESIDUE_RUNS)))
    return p


def debug_set(key: str, value: int):
    """arp_debug_set: the library's diagnostic switches ("timing", "emit_kernel", "defer_entries", "table_host"; include/arpeggia_amd.h)."""
    _check(lib.arp_debug_set(key.encode(), int(value)))


def parse_groups(all_chains, groups: str):
    """utils.rs:71-115 on a throw-away structure with one atom per chain (keeps ONE implementation, in C++)."""
    chains = sorted(set(all_chains))
    n = len(chains)
    rec = {
        "x": np.zeros(n), "y": np.zeros(n), "z": np.arange(n, dtype=np.float64) * 100.0,
        "serial": np.arange(1, n + 1, dtype=np.int32), "resi": np.ones(n, dtype=np.int32),
        "name": np.array([b"CA"] * n, dtype="S8"), "resn": np.array([b"GLY"] * n, dtype="S8"),
        "chain": np.array([c.encode() for c in chains], dtype="S8"), "element": np.array([b"C"] * n, dtype="S4"),
    }
    s = Structure.from_records(rec)
    soa = s.soa(groups)
    lig = {chains[k] for k in range(n) if soa["attr"][k] & _lib.ATTR["LIGAND"]}
    recp = {chains[k] for k in range(n) if soa["attr"][k] & _lib.ATTR["RECEPTOR"]}
    return lig, recp


def _np_from(ptr, n, dtype):
    if not ptr or n == 0:
        return np.zeros(0, dtype=dtype)
    dt = np.dtype(dtype)
    buf = (C.c_char * (n * dt.itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt, count=n).copy()


class _PairsOwner:
    """Keeps a library-allocated pair list alive for the numpy view over it; frees it with the last reference."""

    def __init__(self, pairs):
        self.pairs = pairs

    def __del__(self):
        if self.pairs is not None and lib is not None:
            lib.arp_pairs_free(C.byref(self.pairs))
            self.pairs = None


def _adopt_pairs(out) -> np.ndarray:
    """Zero-copy numpy view of an arp_pairs in host memory (copying a 460 MB list costs 200 ms of page faults)."""
    n = int(out.n)
    if not out.data or n == 0:
        lib.arp_pairs_free(C.byref(out))
        return np.zeros(0, dtype=PAIR_DTYPE)
    owner = _PairsOwner(out)
    buf = (C.c_char * (n * PAIR_DTYPE.itemsize)).from_address(out.data)
    buf._owner = owner  # the ctypes buffer is the array's base: the owner lives exactly as long as the array (and its views)
    return np.frombuffer(buf, dtype=PAIR_DTYPE, count=n)


class Structure:
    """A parsed, filtered model: what `load_model` (utils.rs:51-63) returns in the reference."""

    def __init__(self, handle):
        self._h = handle
        self._keep = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.arp_structure_free(self._h)
            self._h = None

    @classmethod
    def load(cls, path, ignore_zero_occupancy: bool = False) -> "Structure":
        h = C.c_void_p()
        _check(lib.arp_structure_load(os.fsencode(str(path)), int(ignore_zero_occupancy), C.byref(h)))
        return cls(h)

    @classmethod
    def from_records(cls, rec: dict, hierarchy: bool = False) -> "Structure":
        """rec: columns x,y,z f64; serial,resi i32; name,resn,chain S8; element S4; optional occupancy f64, model_serial i32,
        altloc/icode S4; with hierarchy=True also res_ord,res_id u32 (synthetic inputs)."""
        n = len(rec["x"])
        keep = {}

        def col(name, dtype, required=True):
            if name not in rec or rec[name] is None:
                if required:
                    raise KeyError(name)
                return None
            a = np.ascontiguousarray(rec[name], dtype=dtype)
            assert len(a) == n, name
            keep[name] = a
            return a.ctypes.data

        r = _lib.arp_records()
        r.n = n
        r.x, r.y, r.z = col("x", "<f8"), col("y", "<f8"), col("z", "<f8")
        r.occupancy = col("occupancy", "<f8", False)
        r.serial, r.resi = col("serial", "<i4"), col("resi", "<i4")
        r.model_serial = col("model_serial", "<i4", False)
        r.name, r.resn, r.chain = col("name", "S8"), col("resn", "S8"), col("chain", "S8")
        r.altloc, r.icode = col("altloc", "S4", False), col("icode", "S4", False)
        r.element = col("element", "S4")
        r.res_ord, r.res_id = col("res_ord", "<u4", hierarchy), col("res_id", "<u4", hierarchy)
        h = C.c_void_p()
        _check(lib.arp_structure_from_records(C.byref(r), int(hierarchy), C.byref(h)))
        return cls(h)

    @property
    def n_atoms(self) -> int:
        return int(lib.arp_structure_n_atoms(self._h))

    def strings(self, column: str) -> np.ndarray:
        w = C.c_int32()
        p = lib.arp_structure_strings(self._h, column.encode(), C.byref(w))
        if not p:
            raise KeyError(column)
        return _np_from(p, self.n_atoms, f"S{w.value}")

    def ints(self, column: str) -> np.ndarray:
        p = lib.arp_structure_ints(self._h, column.encode())
        if not p:
            raise KeyError(column)
        return _np_from(p, self.n_atoms, "<i4")

    def view(self, groups: str = "/") -> _lib.arp_atoms:
        """Borrowed host SoA view (valid until the next view()/soa() call on this structure)."""
        v = _lib.arp_atoms()
        _check(lib.arp_structure_atoms(self._h, groups.encode(), C.byref(v)))
        return v

    def soa(self, groups: str = "/") -> dict:
        """Copy of the SoA columns the GPU path consumes (include/arpeggia_amd.h: arp_atoms)."""
        v = self.view(groups)
        n, nr = int(v.n), int(v.n_res)
        out = {
            "x": _np_from(v.x, n, "<f8"), "y": _np_from(v.y, n, "<f8"), "z": _np_from(v.z, n, "<f8"),
            "attr": _np_from(v.attr, n, "<u4"), "res_ord": _np_from(v.res_ord, n, "<u4"),
            "chain_rank": _np_from(v.chain_rank, n, "<u4"), "model": _np_from(v.model, n, "<u4"),
            "res_id": _np_from(v.res_id, n, "<u4"),
            "res_h_ptr": _np_from(v.res_h_ptr, nr + 1 if nr else 0, "<u4"),
            "res_cb": _np_from(v.res_cb, nr, "<u4"), "res_sg": _np_from(v.res_sg, nr, "<u4"),
        }
        nh = int(out["res_h_ptr"][-1]) if nr else 0
        out["res_h_idx"] = _np_from(v.res_h_idx, nh, "<u4")
        return out


def atoms_from_arrays(soa: dict, location: int = _lib.ARP_MEM_HOST, keep: list | None = None) -> _lib.arp_atoms:
    """Build an arp_atoms from numpy arrays (host) or from objects with .data_ptr() (device tensors)."""
    v = _lib.arp_atoms()

    def ptr(name, dtype):
        a = soa.get(name)
        if a is None:
            return None
        if hasattr(a, "data_ptr"):
            # a device tensor is handed over as a raw address: its element size and layout are all that can be checked here, and they must be
            # what arp_atoms declares (API v2: 32-bit chain ranks and models -- a v1-style int16 tensor would be read 4 bytes per atom)
            want = np.dtype(dtype).itemsize
            if a.element_size() != want or not a.is_contiguous():
                raise ArpeggiaError(_lib.ARP_ERR_BAD_INPUT, f"arp_atoms.{name}: needs a contiguous tensor of {want}-byte elements ({dtype}), "
                                    f"got element size {a.element_size()}, contiguous={a.is_contiguous()}")
            if keep is not None:
                keep.append(a)
            return a.data_ptr() if a.numel() else None
        a = np.ascontiguousarray(a, dtype=dtype)
        if keep is not None:
            keep.append(a)
        return a.ctypes.data if a.size else None

    x = soa["x"]
    v.n = int(x.numel() if hasattr(x, "numel") else len(x))
    v.x, v.y, v.z = ptr("x", "<f8"), ptr("y", "<f8"), ptr("z", "<f8")
    v.attr, v.res_ord = ptr("attr", "<u4"), ptr("res_ord", "<u4")
    v.chain_rank, v.model = ptr("chain_rank", "<u4"), ptr("model", "<u4")
    rcb = soa.get("res_cb")
    v.n_res = int((rcb.numel() if hasattr(rcb, "numel") else len(rcb))) if rcb is not None else 0
    if v.n_res:
        v.res_id = ptr("res_id", "<u4")
        v.res_h_ptr, v.res_h_idx = ptr("res_h_ptr", "<u4"), ptr("res_h_idx", "<u4")
        v.res_cb, v.res_sg = ptr("res_cb", "<u4"), ptr("res_sg", "<u4")
    v.location = location
    return v


class Context:
    """One engine context = one device + one stream + a reusable workspace (arp_context)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        _check(lib.arp_context_create(device, C.byref(h)))
        self._h = h
        self._keep = []
        if stream is not None:
            self.set_stream(stream)

    def __del__(self):
        if getattr(self, "_h", None):
            lib.arp_context_destroy(self._h)
            self._h = None

    def set_stream(self, hip_stream: int | None):
        _check(lib.arp_context_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def synchronize(self):
        _check(lib.arp_context_synchronize(self._h))

    def atomic_contacts(self, atoms, params: _lib.arp_params | None = None) -> np.ndarray:
        """Synchronous hot path (complex.rs:189-299).  atoms: Structure view / arp_atoms / dict of numpy arrays."""
        keep = []
        if isinstance(atoms, dict):
            atoms = atoms_from_arrays(atoms, keep=keep)
        params = params or default_params()
        out = _lib.arp_pairs()
        _check(lib.arp_contacts_atomic(self._h, C.byref(atoms), C.byref(params), _lib.ARP_MEM_HOST, C.byref(out)))
        return _adopt_pairs(out)

    def enqueue(self, atoms: _lib.arp_atoms, params: _lib.arp_params, out_ptr: int, capacity: int):
        """Asynchronous, allocation-free form on device-resident data (arp_contacts_atomic_enqueue)."""
        _check(lib.arp_contacts_atomic_enqueue(self._h, C.byref(atoms), C.byref(params), C.c_void_p(out_ptr), capacity))

    def count(self, atoms: _lib.arp_atoms, params: _lib.arp_params) -> int:
        """Number of classified pairs without writing any (sizes the buffer for enqueue)."""
        self.enqueue(atoms, params, 0, 0)
        n = C.c_uint64()
        st = lib.arp_contacts_atomic_result(self._h, C.byref(n))
        if st not in (_lib.ARP_OK, _lib.ARP_ERR_CAPACITY):
            _check(st)
        return int(n.value)

    def result(self) -> int:
        n = C.c_uint64()
        st = lib.arp_contacts_atomic_result(self._h, C.byref(n))
        if st == _lib.ARP_ERR_CAPACITY:
            raise ArpeggiaError(st, f"pair buffer too small: {n.value} pairs needed")
        _check(st)
        return int(n.value)

    def profile(self, on: bool):
        _check(lib.arp_profile_enable(self._h, int(on)))

    def profile_read(self) -> dict:
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        k = lib.arp_profile_read(self._h, names, ms, 32)
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    def get_contacts(self, structure: Structure, groups: str = "/", vdw_comp: float = 0.1, dist_cutoff: float = 6.5) -> dict:
        """`arpeggia::get_contacts` (mod.rs:61-137): the sorted 20-column table as a dict of numpy columns."""
        t = C.c_void_p()
        _check(lib.arp_get_contacts(self._h, structure._h, groups.encode(), vdw_comp, dist_cutoff, C.byref(t)))
        try:
            n = int(lib.arp_table_rows(t))
            cols = {}

            def col(name, dtype=None):
                w = C.c_int32()
                p = lib.arp_table_column(t, name.encode(), C.byref(w))
                if not p and n:
                    raise KeyError(name)
                return _np_from(p, n, dtype or f"S{w.value}")

            for name, kind in TABLE_COLUMNS:
                if name == "interaction":
                    cols[name] = col(name, "<i4")
                elif kind == "str":
                    cols[name] = col(name)
                else:
                    cols[name] = col(name, "<" + kind)
            cols["sc_valid"] = col("sc_valid", "u1").astype(bool)
            cols["from_atom"] = col("from_atom", "<i4")
            cols["to_atom"] = col("to_atom", "<i4")
            return cols
        finally:
            lib.arp_table_free(t)


def sap_weight(resn: str, sasa: float) -> float:
    """hydrophobicity(resn) * clamp(sasa / max side-chain SASA(resn), 0, 1) as src/sap.rs:41-101,198-209 forms it (f32)."""
    return float(lib.arp_sap_weight(str(resn).encode(), C.c_float(sasa)))


def sap_neighbor_sum(ctx: "Context", x, y, z, sidechain, weight, sap_radius: float = 5.0) -> np.ndarray:
    """The radius sum of the SAP score (src/sap.rs:155-204) on the contact engine's cell list: for every side-chain atom the f32 sum of
    `weight` over the side-chain atoms within `sap_radius` (itself included).  The per-atom SASA behind the weights is the caller's."""
    x, y, z = (np.ascontiguousarray(v, dtype="<f8") for v in (x, y, z))
    m = np.ascontiguousarray(sidechain, dtype=np.uint8)
    w = np.ascontiguousarray(weight, dtype="<f4")
    out = np.zeros(len(x), dtype="<f4")
    dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
    _check(lib.arp_sap_neighbor_sum(ctx._h, len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp), z.ctypes.data_as(dp), m.ctypes.data_as(C.POINTER(C.c_uint8)),
                                    w.ctypes.data_as(fp), C.c_float(sap_radius), out.ctypes.data_as(fp)))
    return out


def atomic_contacts_batch(contexts, atoms_list, params: _lib.arp_params | None = None) -> list:
    """Independent structures over one or more device contexts (arp_contacts_atomic_batch): longest-first deal over the
    contexts, small structures packed into shared launches.  Returns one pair array per structure, in input order."""
    contexts = list(contexts)
    keep = []
    views = [atoms_from_arrays(a, keep=keep) if isinstance(a, dict) else a for a in atoms_list]
    params = params or default_params()
    arr = (C.POINTER(_lib.arp_atoms) * max(len(views), 1))(*[C.pointer(v) for v in views])
    outs = (_lib.arp_pairs * max(len(views), 1))()
    handles = (C.c_void_p * len(contexts))(*[c._h for c in contexts])
    st = lib.arp_contacts_atomic_batch(handles, len(contexts), arr, len(views), C.byref(params), outs)
    if st != _lib.ARP_OK:
        for k in range(len(views)):
            lib.arp_pairs_free(C.byref(outs[k]))
        _check(st)
    result = []
    for k in range(len(views)):  # every list becomes a zero-copy numpy view that frees it with its last reference
        one = _lib.arp_pairs()
        one.n, one.data, one.location = outs[k].n, outs[k].data, outs[k].location
        result.append(_adopt_pairs(one))
    return result


_default_ctx: dict = {}


def _context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


def load_model(input_file, ignore_zero_occupancy: bool = False) -> Structure:
    """utils.rs:51-63."""
    return Structure.load(input_file, ignore_zero_occupancy)


def get_contacts(structure: Structure, groups: str = "/", vdw_comp: float = 0.1, dist_cutoff: float = 6.5, device: int = 0,
                 num_threads: int = -1):
    """`arpeggia::get_contacts(&pdb, groups, vdw_comp, dist_cutoff) -> DataFrame` (mod.rs:61).

    The table crosses into Python as one Arrow record batch (arp_table_export_arrow): no per-row Python work.
    `num_threads`: host workers of THIS call (0 = all cores, < 0 = the process-wide default of arp_set_num_threads)."""
    return _table(_context(device), structure, groups, vdw_comp, dist_cutoff, num_threads)


def _table(ctx: "Context", structure: Structure, groups: str, vdw_comp: float, dist_cutoff: float, num_threads: int = -1):
    import pyarrow as pa

    t = C.c_void_p()
    _check(lib.arp_get_contacts_mt(ctx._h, structure._h, groups.encode(), vdw_comp, dist_cutoff, int(num_threads), C.byref(t)))
    try:
        arr, sch = _lib.ArrowArray(), _lib.ArrowSchema()
        _check(lib.arp_table_export_arrow(t, C.byref(arr), C.byref(sch)))
    finally:
        lib.arp_table_free(t)
    table = pa.Table.from_batches([pa.RecordBatch._import_from_c(C.addressof(arr), C.addressof(sch))])
    try:  # the reference returns a polars.DataFrame (python.rs:55); same Arrow buffers when polars is installed
        import polars as pl

        return pl.from_arrow(table)
    except ImportError:
        return table


def contacts(input_file: str, groups: str = "/", vdw_comp: float = 0.1, dist_cutoff: float = 6.5,
             ignore_zero_occupancy: bool = False, num_threads: int = 1):
    """Drop-in for `arpeggia.contacts` (src/python.rs:31-56): same keywords and defaults, 20-column table.

    Returns a polars.DataFrame when polars is importable, else the identical pyarrow.Table.  `num_threads` sizes the host
    workers of this call only (the reference builds a scoped rayon pool per call, utils.rs:8-30; 0 = all cores); no
    process-wide state is touched, and the search and classification run on the GPU regardless.
    """
    s = Structure.load(input_file, ignore_zero_occupancy)
    return get_contacts(s, groups, vdw_comp, dist_cutoff, num_threads=int(num_threads))


def contacts_batch(input_files, groups: str = "/", vdw_comp: float = 0.1, dist_cutoff: float = 6.5, ignore_zero_occupancy: bool = False,
                   num_workers: int = 8, devices=None) -> list:
    """`contacts()` over many files: parsing, GPU pairs and table assembly of different files overlap on `num_workers` host threads
    (the C library releases the GIL), each with its own context; files are dealt round-robin over `devices` (default: all visible
    gfx950 devices).  Results come back in input order.  This is the file-level form of BASELINE config 5 (a batch of independent
    structures): on real files the parse and the host table, not the GPU, set the pace, and they scale with the workers."""
    import threading
    from concurrent.futures import ThreadPoolExecutor

    files = [str(f) for f in input_files]
    devs = list(devices) if devices is not None else list(range(max(device_count(), 1)))
    local = threading.local()

    def one(job):
        k, path = job
        dev = devs[k % len(devs)]
        ctxs = getattr(local, "ctxs", None)
        if ctxs is None:
            ctxs = local.ctxs = {}
        if dev not in ctxs:
            ctxs[dev] = Context(dev)
        s = Structure.load(path, ignore_zero_occupancy)
        return _table(ctxs[dev], s, groups, vdw_comp, dist_cutoff)

    if not files:
        return []
    with ThreadPoolExecutor(max_workers=max(1, min(int(num_workers), len(files)))) as pool:
        return list(pool.map(one, enumerate(files)))
