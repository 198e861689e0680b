"""Command-line driver: `python -m arpeggia_amd contacts -i model.pdb -o out/` -- the flags and defaults of the reference's
`arpeggia contacts` (src/cli/contacts.rs:9-52) over the MI355X engine.  Only the `contacts` subcommand exists here (the
one hot path this repository replaces); it writes <output>/<filename>.<format> like cli/contacts.rs:108-137.
"""
from __future__ import annotations

import argparse
import json
import logging
import sys
from pathlib import Path

FORMATS = ("csv", "parquet", "json", "ndjson")  # utils.rs:148-167 DataFrameFileType

log = logging.getLogger("arpeggia_amd")


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(prog="arpeggia_amd", description="Interatomic contacts on MI355X (arpeggia-compatible)")
    sub = ap.add_subparsers(dest="command", required=True)
    c = sub.add_parser("contacts", help="atomic and ring contacts of a PDB / mmCIF model (cli/contacts.rs)")
    c.add_argument("-i", "--input", required=True, type=Path, help="Path to the PDB or mmCIF file to be analyzed")
    c.add_argument("-o", "--output", required=True, type=Path, help="Output directory")
    c.add_argument("-g", "--groups", default="/", help="Chain groups, e.g. A,B/C,D ('/' = all against all)")
    c.add_argument("-f", "--filename", default="contacts", help="Name of the output file")
    c.add_argument("-t", "--output-format", default="csv", type=str.lower, choices=FORMATS, help="Output file type")
    c.add_argument("-c", "--vdw-comp", default=0.1, type=float, help="Compensation factor for VdW radii dependent interaction types")
    c.add_argument("-d", "--dist-cutoff", default=6.5, type=float, help="Distance cutoff when searching for neighboring atoms")
    c.add_argument("-j", "--num-threads", default=1, type=int, help="Host threads of the table path (0 = all cores); the search runs on the GPU")
    c.add_argument("--ignore-zero-occupancy", action="store_true", help="Ignore atoms with zero occupancy")
    return ap


def write_table(table, path: Path, fmt: str) -> None:
    """write_df_to_file (utils.rs:117-146): csv / parquet / json (one array) / ndjson (one object per line)."""
    import pyarrow as pa

    if not isinstance(table, pa.Table):
        table = table.to_arrow()  # polars.DataFrame
    if fmt == "csv":
        import pyarrow.csv as pacsv

        pacsv.write_csv(table, str(path))
    elif fmt == "parquet":
        import pyarrow.parquet as pq

        pq.write_table(table, str(path))
    else:
        rows = table.to_pylist()
        with open(path, "w") as f:
            if fmt == "json":
                json.dump(rows, f)
            else:
                for r in rows:
                    f.write(json.dumps(r) + "\n")


def run_contacts(args) -> int:
    import arpeggia_amd as aa
    from arpeggia_amd import _lib

    if not args.input.exists():
        log.error("Failed to retrieve input file: %s", args.input)  # cli/contacts.rs:58-64
        return 1
    _lib.lib.arp_set_num_threads(int(args.num_threads))
    s = aa.Structure.load(str(args.input.resolve()), args.ignore_zero_occupancy)
    if not (s.soa("/")["attr"] & _lib.ATTR["H"]).any():
        log.warning("No hydrogen atoms found in the structure. This may affect the accuracy of the results.")  # :91-100
    table = aa.get_contacts(s, args.groups, args.vdw_comp, args.dist_cutoff)
    args.output.mkdir(parents=True, exist_ok=True)
    out = (args.output / args.filename).with_suffix("." + args.output_format)
    arrow = table if hasattr(table, "column") and not hasattr(table, "to_arrow") else table.to_arrow()
    n_clash = sum(1 for v in arrow.column("interaction").to_pylist() if v == "StericClash")
    if n_clash:
        log.warning("Found %d steric %s", n_clash, "clash" if n_clash == 1 else "clashes")  # :117-132
    write_table(table, out, args.output_format)
    log.info("Results saved to %s", out)
    return 0


def main(argv=None) -> int:
    logging.basicConfig(level=logging.INFO, format="%(levelname)s %(message)s", stream=sys.stderr)
    args = build_parser().parse_args(argv)
    return run_contacts(args)


if __name__ == "__main__":
    raise SystemExit(main())
