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
