"""Build libarpeggia_amd.so (HIP kernels + C++ host engine) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only authoring container.  The built .so is
git-ignored but travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIB = PKG / "libarpeggia_amd.so"
SOURCES = ["kernels.hip", "engine.cpp", "structure.cpp", "table.cpp", "table_dev.hip"]
HEADERS = ["arp_internal.h", "host_common.h", "debug_knobs.h", "grid.inl", "pairs.inl", "pairs_emit.inl", "batch.inl", "sap.inl", "table_dev.h", "../../include/arpeggia_amd.h"]
TEST_HEADERS = ["../../tests/hosttable/table_host.inl"]  # only the test library (build_host_table_library) contains it: not part of the product's hash
STAMP = PKG / "build" / "libarpeggia_amd.sha256"  # hash of every source + the flags the library was last built from
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-fvisibility=default", "-Wall", "-Wno-unused-result",
         # the kernels aggregate their atomics by hand (one lane per wave / per run); the compiler's own wave aggregation only wraps
         # those single-lane atomics in dead mbcnt / readfirstlane / multiply sequences
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def source_hash(extra: tuple = ()) -> str:
    import hashlib

    h = hashlib.sha256()
    for f in SOURCES + HEADERS + (TEST_HEADERS if extra else []):
        h.update(f.encode()); h.update((CSRC / f).read_bytes())
    h.update(" ".join(FLAGS + list(extra)).encode())
    return h.hexdigest()


def needs_build() -> bool:
    """True unless the library was built from exactly the sources (and flags) in the tree: a content hash, not a timestamp -- a
    checkout, a copy to another machine or an edit that is later reverted neither forces nor hides a rebuild."""
    if not LIB.exists() or not STAMP.exists():
        return True
    return STAMP.read_text().strip() != source_hash()


def build_library(force: bool = False, verbose: bool = False, defines: tuple = (), out: Path | None = None) -> Path:
    """defines/out: test-only builds (build_host_table_library: -DARP_WITH_HOST_TABLE), loaded through ARPEGGIA_AMD_LIB."""
    if not force and not defines and not needs_build():
        return LIB
    objs = []
    build_dir = PKG / "build"
    build_dir.mkdir(exist_ok=True)
    cc = hipcc()
    for src in SOURCES:
        obj = build_dir / (src.replace(".", "_") + ".o")
        cmd = [cc, *FLAGS, *[f"-D{d}" for d in defines], "-c", str(CSRC / src), "-o", str(obj)]
        if src.endswith(".cpp"):
            cmd[1:1] = ["-x", "hip"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        objs.append(str(obj))
    target = Path(out) if out else LIB
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(target), "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    if target == LIB and not defines:
        STAMP.write_text(source_hash() + "\n")
    return target


HOST_TABLE_LIB = PKG.parent / "tests" / "hosttable" / "build" / "libarpeggia_amd_hosttable.so"


def build_host_table_library(force: bool = False) -> Path:
    """TEST-ONLY library: the product's objects with table.cpp recompiled under -DARP_WITH_HOST_TABLE, i.e. plus the round-1 host assembly
    of the contact table (tests/hosttable/table_host.inl) that tests/test_gpu_parity.py cross-checks the device table with.  The product library
    does not contain that code."""
    build_library()  # the product's objects must be current
    stamp = HOST_TABLE_LIB.with_suffix(".sha256")
    want = source_hash(("-DARP_WITH_HOST_TABLE",))
    if not force and HOST_TABLE_LIB.exists() and stamp.exists() and stamp.read_text().strip() == want:
        return HOST_TABLE_LIB
    HOST_TABLE_LIB.parent.mkdir(parents=True, exist_ok=True)
    cc = hipcc()
    obj = HOST_TABLE_LIB.parent / "table_cpp_host.o"
    subprocess.run([cc, "-x", "hip", *FLAGS, "-DARP_WITH_HOST_TABLE", "-c", str(CSRC / "table.cpp"), "-o", str(obj)], check=True)
    objs = [str(obj) if src == "table.cpp" else str(PKG / "build" / (src.replace(".", "_") + ".o")) for src in SOURCES]
    subprocess.run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(HOST_TABLE_LIB), "-lpthread"], check=True)
    stamp.write_text(want + "\n")
    return HOST_TABLE_LIB


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    print(build_library(force=True, verbose=not a.define, defines=tuple(a.define), out=a.out))
