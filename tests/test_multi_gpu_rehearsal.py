"""Multi-GPU readiness without an 8-GPU node: `bench.py --gpus 2 --workload batch5k` as two fresh processes that share
cuda:0 and rendezvous over gloo (ARP_BENCH_REHEARSE=1) -- the launch line the driver uses, minus RCCL -- against the same job on one rank.
The path shards over independent structures with no data-path collective (reference: one process, a rayon pool, src/utils.rs:8-30), so
the job's pair total must not depend on the number of ranks."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(n_ranks: int, structures: int) -> dict:
    args = ["bench.py", "--gpus", str(n_ranks), "--workload", "batch5k", "--structures", str(structures), "--no-cpu-baseline", "--steps", "3", "--warmup", "1",
            "--profile-steps", "1"]
    env = dict(os.environ, ARP_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    if n_ranks == 1:
        cmd = [sys.executable, *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1", "--master-port",
               str(_free_port()), *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line from rank 0, got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_rank_rehearsal_of_the_config5_batch_matches_one_rank():
    one = _bench(1, 256)
    two = _bench(2, 256)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and one["scaling"] == "strong"  # --structures fixes the total
    assert two["metric"] == one["metric"] and two["unit"] == "classified atom-pairs/s"
    # the deal is longest-first over the ranks: rank 0 holds half of the structures, and the shares add up to the one-rank job
    assert two["config"]["pairs_all_gpus"] == one["config"]["pairs_all_gpus"] == one["config"]["pairs_per_gpu"]
    assert 0 < two["config"]["pairs_per_gpu"] < one["config"]["pairs_per_gpu"]
    assert abs(two["config"]["atoms_per_gpu"] * 2 - one["config"]["atoms_per_gpu"]) <= 7000  # within one structure of an even split
    assert two["value"] > 0 and two["roofline"]["frac"] > 0


def _bench_default(n_ranks: int, atoms: int) -> dict:
    """The driver's DEFAULT line -- no --workload: S2 weak scaling, with the S1 / 10^5 / files / SAP / batch5k legs of the one JSON line -- as n_ranks
    processes on the one card."""
    args = ["bench.py", "--gpus", str(n_ranks), "--atoms", str(atoms), "--steps", "3", "--warmup", "1", "--profile-steps", "1"]
    env = dict(os.environ, ARP_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), *args]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line from rank 0, got {len(lines)}"
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_rank_rehearsal_of_the_default_bench_line():
    """the line the driver runs for SCALE (`bench.py --gpus N --steps K --warmup W`, no --workload) had never run with more than
    one rank.  Both ranks must pass the same collectives while rank 0 alone runs the 10^5-atom, files, SAP and host-path legs in between; one
    JSON line; `pairs_all_gpus` = the sum over two DIFFERENT clouds (rank r seeds its own), checked against the oracle's count of rank 1's."""
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import oracle_binding as ob
    import synth

    import bench

    atoms = 100_000
    line = _bench_default(2, atoms)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["unit"] == "classified atom-pairs/s"
    rank1 = synth.gen_s2(atoms, seed=bench.SEED + 4 + 1000)
    n1 = len(ob.Structure.from_atoms(synth.records_to_oracle(rank1, flat=True), flat=True).atomic_contacts("/", 0.1, 6.5))
    cfg = line["config"]
    assert n1 != cfg["pairs_per_gpu"] and cfg["pairs_all_gpus"] == cfg["pairs_per_gpu"] + n1
    assert line["value"] > 0 and 0 < line["roofline"]["frac"] < 1 and "cpu_baseline" not in line  # (the CPU leg is N = 1 only)
    for leg in ("s1", "s2_1e5", "s1_1e5", "files", "sap", "batch5k"):
        assert leg in line, leg
    assert line["s1"]["value"] > 0 and line["batch5k"]["structures"] == 2500 and line["batch5k"]["value"] > 0
    assert line["files"]["1ubq"]["table_rows"] == 532 and line["files"]["6bft"]["pairs"] == 124047
    assert line["first_call_ms"] > 0 and "probe pass" in cfg["speculation"]
