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


@pytest.fixture(scope="session")
def ubq_path():
    return str(DATA / "1ubq.pdb")


@pytest.fixture(scope="session")
def bft_path():
    return str(DATA / "6bft.pdb")
