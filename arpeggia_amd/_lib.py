"""ctypes declarations for libarpeggia_amd.so (include/arpeggia_amd.h).

The HIP extension is the product: if the shared library is missing this module raises ImportError -- there is no
Python or CPU fallback for the compute path.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libarpeggia_amd.so"

ARP_OK = 0
(ARP_ERR_BAD_GROUPS, ARP_ERR_EMPTY_GROUPS, ARP_ERR_NO_RINGS, ARP_ERR_BAD_INPUT, ARP_ERR_HIP, ARP_ERR_OOM,
 ARP_ERR_NO_DEVICE, ARP_ERR_IO, ARP_ERR_CAPACITY) = range(1, 10)
ARP_MEM_HOST, ARP_MEM_DEVICE = 0, 1
ARP_NONE = 0xFFFFFFFF
ARP_FLAG_DETERMINISTIC = 0x1
ARP_FLAG_CONTACTS_ONLY = 0x2
ARP_FLAG_NO_SPECULATION = 0x4
ARP_FLAG_RESIDUE_RUNS = 0x8
ARP_FLAG_NO_RESIDUE_RUNS = 0x10

ATTR = dict(
    ELEM_MASK=0xF, DONOR=0x10, ACCEPTOR=0x20, WEAK_DONOR=0x40, POS=0x80, NEG=0x100, HYDROPHOBIC=0x200, CYS_SG=0x400,
    H=0x800, LIGAND=0x1000, RECEPTOR=0x2000, POS_RESN=0x4000,
)

INTERACTIONS = [
    "StericClash", "CovalentBond", "Disulfide", "VanDerWaalsContact", "IonicBond", "HydrogenBond", "WeakHydrogenBond",
    "PolarContact", "WeakPolarContact", "IonicRepulsion", "SaltBridge", "PiDisplacedStacking", "PiTStacking",
    "PiSandwichStacking", "PiParallelInPlaneStacking", "PiTiltedStacking", "PiLStacking", "CationPi", "HydrophobicContact",
]

_dp = C.POINTER(C.c_double)
_u32p = C.POINTER(C.c_uint32)
_u16p = C.POINTER(C.c_uint16)
_i32p = C.POINTER(C.c_int32)


class arp_atoms(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("attr", C.c_void_p),
        ("res_ord", C.c_void_p), ("chain_rank", C.c_void_p), ("model", C.c_void_p), ("res_id", C.c_void_p),
        ("n_res", C.c_uint64), ("res_h_ptr", C.c_void_p), ("res_h_idx", C.c_void_p), ("res_cb", C.c_void_p),
        ("res_sg", C.c_void_p), ("location", C.c_int32), ("reserved", C.c_int32),
    ]


class arp_params(C.Structure):
    _fields_ = [
        ("vdw_comp", C.c_double), ("dist_cutoff", C.c_double), ("cov_radius", C.c_double * 16), ("vdw_radius", C.c_double * 16),
        ("h_vdw_radius", C.c_double), ("flags", C.c_uint32), ("reserved", C.c_uint32),
    ]


class arp_pair(C.Structure):
    _fields_ = [("i", C.c_uint32), ("j", C.c_uint32), ("dist", C.c_float), ("kind", C.c_uint32)]


class arp_pairs(C.Structure):
    _fields_ = [("n", C.c_uint64), ("data", C.c_void_p), ("location", C.c_int32), ("reserved", C.c_int32)]


class arp_records(C.Structure):
    _fields_ = [
        ("n", C.c_uint64), ("x", C.c_void_p), ("y", C.c_void_p), ("z", C.c_void_p), ("occupancy", C.c_void_p),
        ("serial", C.c_void_p), ("resi", C.c_void_p), ("model_serial", C.c_void_p), ("name", C.c_void_p), ("resn", C.c_void_p),
        ("chain", C.c_void_p), ("altloc", C.c_void_p), ("icode", C.c_void_p), ("element", C.c_void_p), ("res_ord", C.c_void_p),
        ("res_id", C.c_void_p),
    ]


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (same SONAME as the system one,
    different file); if this library pulled in /opt/rocm's copy first and torch came later, the process would hold two
    runtimes and torch would see no GPU (and its device pointers would be foreign to ours).  So when torch is installed,
    its runtime is loaded first and libarpeggia_amd.so's DT_NEEDED libamdhip64.so.7 binds to it by SONAME.
    ARPEGGIA_AMD_HIP_RUNTIME=/path/to/libamdhip64.so overrides the choice."""
    import importlib.util
    import os

    path = os.environ.get("ARPEGGIA_AMD_HIP_RUNTIME")
    if not path:
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
            if cand.exists():
                path = str(cand)
    if path:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
    return path


class ArrowSchema(C.Structure):  # Arrow C Data Interface (ABI-stable), include/arpeggia_amd.h
    _fields_ = [("format", C.c_char_p), ("name", C.c_char_p), ("metadata", C.c_char_p), ("flags", C.c_int64), ("n_children", C.c_int64),
                ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


class ArrowArray(C.Structure):
    _fields_ = [("length", C.c_int64), ("null_count", C.c_int64), ("offset", C.c_int64), ("n_buffers", C.c_int64), ("n_children", C.c_int64),
                ("buffers", C.c_void_p), ("children", C.c_void_p), ("dictionary", C.c_void_p), ("release", C.c_void_p), ("private_data", C.c_void_p)]


def _load():
    import os

    global LIB_PATH
    if os.environ.get("ARPEGGIA_AMD_LIB"):  # diagnostic builds (timing ablations); same ABI
        LIB_PATH = Path(os.environ["ARPEGGIA_AMD_LIB"])
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()' "
            "or python arpeggia_amd/build.py). arpeggia_amd has no CPU fallback."
        )
    _preload_hip_runtime()
    L = C.CDLL(str(LIB_PATH))
    vp = C.c_void_p
    sig = {
        "arp_debug_set": (C.c_int32, [C.c_char_p, C.c_int64]),
        "arp_api_version": (C.c_int32, []),
        "arp_check_api_version": (C.c_int32, [C.c_int32]),
        "arp_strerror": (C.c_char_p, [C.c_int32]),
        "arp_last_error": (C.c_char_p, []),
        "arp_device_count": (C.c_int32, []),
        "arp_interaction_name": (C.c_char_p, [C.c_int32]),
        "arp_default_params": (None, [C.POINTER(arp_params)]),
        "arp_element_class": (C.c_int32, [C.c_char_p]),
        "arp_context_create": (C.c_int32, [C.c_int32, C.POINTER(vp)]),
        "arp_context_destroy": (None, [vp]),
        "arp_context_set_stream": (C.c_int32, [vp, vp]),
        "arp_context_synchronize": (C.c_int32, [vp]),
        "arp_contacts_atomic": (C.c_int32, [vp, C.POINTER(arp_atoms), C.POINTER(arp_params), C.c_int32, C.POINTER(arp_pairs)]),
        "arp_pairs_free": (None, [C.POINTER(arp_pairs)]),
        "arp_contacts_atomic_enqueue": (C.c_int32, [vp, C.POINTER(arp_atoms), C.POINTER(arp_params), vp, C.c_uint64]),
        "arp_contacts_atomic_result": (C.c_int32, [vp, C.POINTER(C.c_uint64)]),
        "arp_contacts_atomic_batch": (C.c_int32, [C.POINTER(vp), C.c_int32, C.POINTER(C.POINTER(arp_atoms)), C.c_int32, C.POINTER(arp_params), C.POINTER(arp_pairs)]),
        "arp_release_host_pool": (C.c_uint64, []),
        "arp_sap_weight": (C.c_float, [C.c_char_p, C.c_float]),
        "arp_sap_neighbor_sum": (C.c_int32, [vp, C.c_uint64, _dp, _dp, _dp, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.c_float, C.POINTER(C.c_float)]),
        "arp_profile_enable": (C.c_int32, [vp, C.c_int32]),
        "arp_profile_read": (C.c_int32, [vp, C.POINTER(C.c_char_p), C.POINTER(C.c_float), C.c_int32]),
        "arp_structure_load": (C.c_int32, [C.c_char_p, C.c_int32, C.POINTER(vp)]),
        "arp_structure_from_records": (C.c_int32, [C.POINTER(arp_records), C.c_int32, C.POINTER(vp)]),
        "arp_structure_free": (None, [vp]),
        "arp_structure_n_atoms": (C.c_uint64, [vp]),
        "arp_structure_atoms": (C.c_int32, [vp, C.c_char_p, C.POINTER(arp_atoms)]),
        "arp_structure_strings": (vp, [vp, C.c_char_p, C.POINTER(C.c_int32)]),
        "arp_structure_ints": (vp, [vp, C.c_char_p]),
        "arp_get_contacts": (C.c_int32, [vp, vp, C.c_char_p, C.c_double, C.c_double, C.POINTER(vp)]),
        "arp_get_contacts_mt": (C.c_int32, [vp, vp, C.c_char_p, C.c_double, C.c_double, C.c_int32, C.POINTER(vp)]),
        "arp_structure_n_residues": (C.c_uint64, [vp]),
        "arp_structure_fit_planes": (C.c_int32, [vp, vp, _dp, C.POINTER(C.c_uint8)]),
        "arp_set_num_threads": (None, [C.c_int32]),
        "arp_get_num_threads": (C.c_int32, []),
        "arp_table_free": (None, [vp]),
        "arp_table_rows": (C.c_uint64, [vp]),
        "arp_table_column": (vp, [vp, C.c_char_p, C.POINTER(C.c_int32)]),
        "arp_table_export_arrow": (C.c_int32, [vp, C.POINTER(ArrowArray), C.POINTER(ArrowSchema)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)  # AttributeError here == the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    return L, sorted(sig)


lib, EXPORTS = _load()
