"""Where the table path spends its time on a large structure (S1: ubiquitin copies on a lattice).  ARP_TIMING=1 prints the stages.
Usage (GPU box): ARP_TIMING=1 python tests/table_scaling.py [n_atoms]
The host-assembly rows (arp_debug_set("table_host", 1)) only mean something with the TEST-ONLY library, which is the one that contains that code:
ARPEGGIA_AMD_LIB=tests/hosttable/build/libarpeggia_amd_hosttable.so (arpeggia_amd/build.py build_host_table_library)."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import arpeggia_amd as aa  # noqa: E402

if os.environ.get("ARP_TIMING"):  # (a switch of THIS script: the library reads no environment)
    aa.debug_set("timing", 1)
import synth  # noqa: E402
from arpeggia_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rec = synth.gen_s1(n)
s = aa.Structure.from_records(rec, hierarchy=True)
ctx = aa.Context(0)
lib = _lib.lib


def c_call(threads, host):
    """arp_get_contacts_mt alone, then the Arrow export: (seconds, seconds, rows)."""
    aa.debug_set("table_host", 1 if host else 0)
    t = C.c_void_p()
    t0 = time.perf_counter()
    st = lib.arp_get_contacts_mt(ctx._h, s._h, b"/", 0.1, 6.5, threads, C.byref(t))
    t1 = time.perf_counter()
    assert st == 0, lib.arp_last_error()
    arr, sch = _lib.ArrowArray(), _lib.ArrowSchema()
    lib.arp_set_num_threads(threads)
    assert lib.arp_table_export_arrow(t, C.byref(arr), C.byref(sch)) == 0
    t2 = time.perf_counter()
    rows = int(lib.arp_table_rows(t))
    C.CFUNCTYPE(None, C.c_void_p)(arr.release)(C.addressof(arr)); C.CFUNCTYPE(None, C.c_void_p)(sch.release)(C.addressof(sch))
    lib.arp_table_free(t)
    return t1 - t0, t2 - t1, rows


ref = None
has_host_assembly = "hosttable" in os.environ.get("ARPEGGIA_AMD_LIB", "")  # only the test-only library contains the round-1 host assembly
first = c_call(16, False)  # first call on this structure: uploads the resident copy, builds the entity book, fits the planes
print(f"S1 {s.n_atoms} atoms, device table, 16 host thread(s), FIRST call of the process: get_contacts {first[0] * 1e3:7.1f} ms", file=sys.stderr)
c_call(16, False)
s_first, s = s, aa.Structure.from_records(rec, hierarchy=True)  # a second structure of the same size: its first call, in a process that has run one
print("-- second structure --", file=sys.stderr)
first = c_call(16, False)
print(f"S1 {s.n_atoms} atoms, device table, 16 host thread(s), FIRST call on a second structure: get_contacts {first[0] * 1e3:7.1f} ms", file=sys.stderr)
for host in ((False, True) if has_host_assembly and not os.environ.get("ARP_DEVICE_ONLY") else (False,)):
    for threads in (1, 16):
        c_call(threads, host)
        best = min((c_call(threads, host) for _ in range(3)), key=lambda r: r[0] + r[1])
        print(f"S1 {s.n_atoms} atoms, {'host assembly (round 1)' if host else 'device table'}, {threads:2d} host thread(s): "
              f"get_contacts {best[0] * 1e3:7.1f} ms + Arrow export {best[1] * 1e3:6.1f} ms, {best[2]} rows", file=sys.stderr)
aa.debug_set("table_host", 0)
for threads in (1, 16):
    cols = ctx.get_contacts(s)
    key = tuple(cols[c].tobytes() for c in ("model", "interaction", "distance", "from_atom", "to_atom", "sc_dihedral"))
    assert ref is None or key == ref, "the table must not depend on the thread count"
    ref = key
