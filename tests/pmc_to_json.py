"""profiles/rNN_traffic.json from a PMC summary (tests/pmc_summary.py output): the HBM-side traffic and the instruction-issue figures of the
dominant kernel, TAGGED WITH THE CONTENT HASH OF THE SOURCES the profiled library was built from (arpeggia_amd/build.py source_hash) --
bench.py attaches the measurement to its `roofline` object only when that hash equals the running library's.
Usage: python tests/pmc_to_json.py SUMMARY.txt OUT.json --workload s2 --atoms 1000000 --pairs 28702955 [--kernel 'arp::k_emit<12, 1, false, false' (a prefix)]"""
import argparse
import importlib.util
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def source_hash() -> str:
    spec = importlib.util.spec_from_file_location("_arp_build", ROOT / "arpeggia_amd" / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary"); ap.add_argument("out")
    ap.add_argument("--workload", default="s2"); ap.add_argument("--atoms", type=int, default=1_000_000); ap.add_argument("--pairs", type=int, required=True)
    ap.add_argument("--kernel", default="arp::k_emit<12, 1, false, false")
    ap.add_argument("--source", default=None, help="path of the summary as committed under profiles/")
    a = ap.parse_args()
    txt = Path(a.summary).read_text()
    # (--kernel is a PREFIX of the summary's kernel line: the symbol grows a template argument now and then -- round 5 added two)
    m0 = re.search(r"^" + re.escape(a.kernel) + r"[^\n]*\n", txt, re.M)
    if not m0:
        raise SystemExit(f"no kernel line starting with {a.kernel!r} in {a.summary}")
    a.kernel = m0.group(0).strip()
    block = txt[m0.end():]
    j = re.search(r"^\S", block, re.M)
    block = block[: j.start()] if j else block
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\S+)\s+mean\s+([0-9.eE+-]+)", block, re.M)}
    src = a.source or a.summary
    fetch_kb, write_kb = c["FETCH_SIZE"], c["WRITE_SIZE"]
    cycles_per_xcd = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the eight XCDs
    out = {
        "workload": a.workload, "atoms": a.atoms, "pairs": a.pairs, "emitter": "single-pass", "kernel": "pairs_emit", "kernel_symbol": a.kernel,
        "csrc_hash": source_hash(),
        "fetch_size_kb": fetch_kb, "fetch_correction": 2.0, "write_size_kb": write_kb,
        "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
        "issue": {
            "valu_insts_per_launch": c["SQ_INSTS_VALU"], "salu_insts_per_launch": c["SQ_INSTS_SALU"], "lds_insts_per_launch": c["SQ_INSTS_LDS"],
            "branch_insts_per_launch": c["SQ_INSTS_BRANCH"], "vmem_rd_insts_per_launch": c["SQ_INSTS_VMEM_RD"],
            "lane_insts_per_pair": c["SQ_INSTS_VALU"] * 64.0 / a.pairs,
            # SQ_ACTIVE_INST_VALU counts 4-cycle quanta summed over the chip's 1024 SIMDs; the kernel's cycles come from GRBM_GUI_ACTIVE
            "vector_pipe_busy": c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles_per_xcd * 1024.0),
            "wave_cycles_waiting": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], "wave_cycles_issue_stalled": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
            "lds_bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"],
            "kernel_cycles_per_xcd": cycles_per_xcd,
        },
        "source": f"{src} (rocprofv3 --pmc passes of {a.kernel}, one counter group per run; FETCH_SIZE doubled per MI355X_MICROARCH.md; "
                  f"L2<->fabric requests, Infinity-Cache hits included)",
    }
    Path(a.out).write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out["issue"]))


if __name__ == "__main__":
    sys.exit(main())
