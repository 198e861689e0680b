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
HEADERS = ["arp_internal.h", "host_common.h", "grid.inl", "pairs.inl", "pairs_emit.inl", "batch.inl", "sap.inl", "table_dev.h", "../../include/arpeggia_amd.h"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-fvisibility=default", "-Wall", "-Wno-unused-result",
         # the kernels aggregate their atomics by hand (one lane per wave / per run); the compiler's own wave aggregation only wraps
         # those single-lane atomics in dead mbcnt / readfirstlane / multiply sequences
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built (there is no CPU fallback)")


def needs_build() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    return any((CSRC / f).stat().st_mtime > t for f in SOURCES + HEADERS)


def build_library(force: bool = False, verbose: bool = False, defines: tuple = (), out: Path | None = None) -> Path:
    """defines/out: diagnostic builds only (e.g. -DARP_ABLATE=1 timing ablations, loaded through ARPEGGIA_AMD_LIB)."""
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
    return target


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    print(build_library(force=True, verbose=not a.define, defines=tuple(a.define), out=a.out))
