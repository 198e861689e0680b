"""N > 1 path on CPU: two gloo ranks shard a batch of independent structures with no data-path collective; only counts and
timings are reduced.  The per-structure compute here is the oracle (this is a test of the sharding, not of the kernels)."""
import os
import socket

import numpy as np
import pytest

import synth
from arpeggia_amd.sharding import lpt_assign, reduce_job


def test_lpt_assign_is_a_balanced_partition():
    rng = np.random.default_rng(0)
    sizes = np.clip(rng.normal(5000, 500, size=200), 3000, 7000).astype(int).tolist()  # BASELINE.json configs[4] shape
    for n in (1, 2, 4, 8):
        shards = lpt_assign(sizes, n)
        flat = sorted(k for s in shards for k in s)
        assert flat == list(range(len(sizes)))
        loads = [sum(sizes[k] for k in s) for s in shards]
        assert max(loads) - min(loads) <= max(sizes)
    assert lpt_assign([], 4) == [[], [], [], []]
    assert lpt_assign([5, 5, 5], 2) == [[0, 2], [1]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, sizes, seeds, queue):
    import torch
    import torch.distributed as dist

    import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = lpt_assign(sizes, world)[rank]
    n_pairs, checks = 0, {}
    for k in mine:
        rec = synth.gen_s1(sizes[k], seed=seeds[k])
        s = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
        p = s.atomic_contacts("/", 0.1, 6.5)
        n_pairs += len(p)
        checks[k] = (len(p), int(p["kind"].astype(np.uint64).sum()))
    dist.barrier()
    wall, total = reduce_job(dist, torch.device("cpu"), 0.001 * (rank + 1), n_pairs)
    queue.put((rank, mine, checks, wall, total))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_two_gloo_ranks_shard_a_batch_without_exchange():
    import torch.multiprocessing as mp

    sizes = [900, 1500, 700, 1200, 1100, 800]
    seeds = [40 + k for k in range(len(sizes))]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, sizes, seeds, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out.sort()
    # every structure processed exactly once; reduced totals agree on both ranks and with a single-process run
    assert sorted(out[0][1] + out[1][1]) == list(range(len(sizes)))
    assert out[0][3] == out[1][3] == 0.002  # max over ranks
    assert out[0][4] == out[1][4]
    import oracle_binding as ob

    merged = {**out[0][2], **out[1][2]}
    total = 0
    for k in range(len(sizes)):
        rec = synth.gen_s1(sizes[k], seed=seeds[k])
        p = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True).atomic_contacts("/", 0.1, 6.5)
        assert merged[k] == (len(p), int(p["kind"].astype(np.uint64).sum()))
        total += len(p)
    assert out[0][4] == float(total)
