"""Writes the synthetic inputs of BASELINE.json configs 3-5 as mmCIF files a build of the reference (y1zhou/arpeggia, Rust) can read, so that
whoever has `cargo` can time the reference's own `arpeggia contacts` on the very structures this repo's bench steps on and diff the tables
("If a Rust toolchain ever becomes available, first action = dump the real tables and diff").  Nothing here runs or reads the
reference; the files are this repo's generator output (tests/synth.py gen_s1: rigid 1ubq copies on an fcc lattice, one chain id per copy).

Usage:  python tools/emit_reference_inputs.py OUT_DIR [--atoms 100000 1000000] [--batch 16]
Then, with the reference built (not part of this repo):
    arpeggia contacts -i OUT_DIR/s1_1000000.cif -o ref_out -t csv -j 1      # and -j 0 for all cores
and compare with  python -m arpeggia_amd contacts -i OUT_DIR/s1_1000000.cif -o amd_out -t csv  (same flags, same 20 columns, same row order
up to full ties of the ten sort keys)."""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "tests")]
import numpy as np  # noqa: E402
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("out_dir")
    ap.add_argument("--atoms", type=int, nargs="*", default=[100_000, 1_000_000], help="S1 clouds (configs 3 and 4)")
    ap.add_argument("--batch", type=int, default=16, help="how many config-5 structures (~5k atoms, N(5000, 500^2) clipped to [3000, 7000]) to write")
    args = ap.parse_args()
    out = Path(args.out_dir)
    out.mkdir(parents=True, exist_ok=True)
    for n in args.atoms:
        rec = synth.gen_s1(n)  # the bench's cloud for this size uses the same generator (its seed is printed in the bench line's workload string)
        p = out / f"s1_{n}.cif"
        synth.write_mmcif(rec, p, fancy=False)
        print(f"{p}: {len(rec['x'])} atoms, {len(np.unique(rec['chain']))} chains")
    rng = np.random.default_rng(5)
    sizes = np.clip(np.rint(rng.normal(5000.0, 500.0, args.batch)), 3000, 7000).astype(int)
    for k, n in enumerate(sizes):
        rec = synth.gen_s1(int(n), seed=900 + k)
        p = out / f"batch5k_{k:04d}.cif"
        synth.write_mmcif(rec, p, fancy=False)
    print(f"{out}/batch5k_*.cif: {len(sizes)} structures of {sizes.min()}..{sizes.max()} atoms")


if __name__ == "__main__":
    main()
