"""Regenerate tests/golden/*.csv from the CPU oracle.

These tables are RESTATEMENT-DERIVED: they are produced by this repo's own plain-C restatement (oracle/), not by
the reference binary (Rust; not buildable here).  They are anchored to the reference only through the facts the
reference's own tests pin (532 rows on 1ubq, PHE4 ring, two cation-pi cases; see tests/test_oracle_golden.py).
Usage: python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT / "tests"))
import oracle_binding as ob  # noqa: E402

HEADER = ("model,interaction,distance,from_chain,from_resn,from_resi,from_insertion,from_altloc,from_atomn,from_atomi,"
          "to_chain,to_resn,to_resi,to_insertion,to_altloc,to_atomn,to_atomi,sc_centroid_dist,sc_dihedral,sc_centroid_angle")

for name in ("1ubq", "6bft"):
    s = ob.Structure.load(ROOT / "tests" / "data" / f"{name}.pdb")
    rows = s.get_contacts("/", 0.1, 6.5)
    out = ROOT / "tests" / "golden" / f"{name}_contacts.csv"
    out.write_text("\n".join([HEADER] + ob.rows_to_csv_lines(rows)) + "\n")
    pairs = s.atomic_contacts("/", 0.1, 6.5)
    print(name, "rows", len(rows), "candidate pairs", len(pairs))
