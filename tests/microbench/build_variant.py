"""Diagnostic builds of libarpeggia_amd.so from a patched COPY of the sources (the product tree carries no ablation switches).
Usage: python tests/microbench/build_variant.py NAME 'file::old::new' ['file::old::new' ...]   (old -> new, exact text, first occurrence)
       python tests/microbench/build_variant.py NAME -D MACRO=1 ...                              (extra hipcc defines)
The library lands in tests/microbench/build/libvar_NAME.so; select it with ARPEGGIA_AMD_LIB=... (arpeggia_amd/_lib.py)."""
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "arpeggia_amd"))
import build as B  # noqa: E402

name, args = sys.argv[1], sys.argv[2:]
work = Path("/tmp") / f"arp_var_{name}"
shutil.rmtree(work, ignore_errors=True)
(work / "arpeggia_amd").mkdir(parents=True)
shutil.copytree(ROOT / "arpeggia_amd" / "csrc", work / "arpeggia_amd" / "csrc")
shutil.copytree(ROOT / "include", work / "include")
defines = []
it = iter(args)
for a in it:
    if a == "-D":
        defines.append(next(it)); continue
    f, old, new = a.split("::")
    p = work / "arpeggia_amd" / "csrc" / f
    s = p.read_text()
    assert old in s, f"{f}: pattern not found: {old!r}"
    p.write_text(s.replace(old, new, 1))
out = ROOT / "tests" / "microbench" / "build" / f"libvar_{name}.so"
out.parent.mkdir(exist_ok=True)
objs = []
touched = {a.split("::")[0] for a in args if "::" in a}
only_kernels = not defines and all(f.endswith(".inl") or f == "kernels.hip" for f in touched)
for src in B.SOURCES:
    if only_kernels and src != "kernels.hip":  # the other objects of the last product build are unchanged
        objs.append(str(ROOT / "arpeggia_amd" / "build" / (src.replace(".", "_") + ".o")))
        continue
    obj = work / (src.replace(".", "_") + ".o")
    cmd = [B.hipcc(), *B.FLAGS, *[f"-D{d}" for d in defines], "-c", str(work / "arpeggia_amd" / "csrc" / src), "-o", str(obj)]
    if src.endswith(".cpp"):
        cmd[1:1] = ["-x", "hip"]
    subprocess.run(cmd, check=True, capture_output=True)
    objs.append(str(obj))
subprocess.run([B.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", str(out), "-lpthread"], check=True)
print(out)
