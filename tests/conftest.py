import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# a fresh clone has no built libraries yet: build them once (hipcc cross-compiles gfx950 without a GPU; ~20 s)
if not (ROOT / "arpeggia_amd" / "libarpeggia_amd.so").exists() or not (ROOT / "oracle" / "liboracle.so").exists():
    import __graft_entry__

    __graft_entry__.build()

DATA = ROOT / "tests" / "data"
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # ARP_TEST_STRIP_ROWS=N: the whole suite on cell rows ordered in y strips of N rows (arp_debug_set "strip_rows"; by default only inputs above
    # ~2.5 x 10^6 atoms are) -- a soak of the strip order on every input the suite has.  Tests that set the switch themselves restore it.
    import os

    if os.environ.get("ARP_TEST_STRIP_ROWS"):
        import arpeggia_amd as aa

        aa.debug_set("strip_rows", int(os.environ["ARP_TEST_STRIP_ROWS"]))


@pytest.fixture(scope="session")
def ubq_path():
    return str(DATA / "1ubq.pdb")


@pytest.fixture(scope="session")
def bft_path():
    return str(DATA / "6bft.pdb")


@pytest.fixture(scope="session")
def c_consumer():
    """tests/c_abi/consumer.c compiled as plain C11 against include/arpeggia_amd.h and linked to libarpeggia_amd.so: the compiled consumer
    of the boundary (its _Static_asserts pin every field offset a #[repr(C)] binder hard-codes)."""
    import shutil
    import subprocess

    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc on this machine")
    out = ROOT / "tests" / "c_abi" / "build" / "consumer"
    out.parent.mkdir(exist_ok=True)
    lib_dir = ROOT / "arpeggia_amd"
    cmd = [gcc, "-std=c11", "-Wall", "-Werror", "-O1", f"-I{ROOT / 'include'}", str(ROOT / "tests" / "c_abi" / "consumer.c"), "-o", str(out), f"-L{lib_dir}",
           "-larpeggia_amd", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)
