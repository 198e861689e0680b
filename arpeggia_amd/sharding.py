"""Host-side sharding of independent structures over ranks / devices.

The contacts path has no cross-structure state (`get_contacts` takes one &PDB, mod.rs:61), so a batch shards with no
data-path collective: structures are dealt longest-processing-time-first by atom count, every rank runs its share, and
only the bookkeeping (pair counts, timings) is reduced.  arp_contacts_atomic_batch() applies the same deal inside one
process (one host thread + one context per device); bench.py and the tests use it across processes.
"""
from __future__ import annotations

from typing import Sequence


def lpt_assign(sizes: Sequence[int], n_shards: int) -> list[list[int]]:
    """Indices of `sizes` per shard: largest first, each to the currently lightest shard (ties: lowest shard id).

    Mirrors arp_contacts_atomic_batch (engine.cpp): load += size + 1, stable descending order."""
    order = sorted(range(len(sizes)), key=lambda k: -sizes[k])  # Python's sort is stable, like std::stable_sort
    shards: list[list[int]] = [[] for _ in range(n_shards)]
    load = [0] * n_shards
    for k in order:
        best = min(range(n_shards), key=lambda s: (load[s], s))
        shards[best].append(k)
        load[best] += sizes[k] + 1
    return shards


def reduce_job(dist, device, wall_seconds: float, n_units: float):
    """(max over ranks of wall time, sum over ranks of processed units).  `dist` is torch.distributed or None."""
    import torch

    t = torch.tensor([wall_seconds], dtype=torch.float64, device=device)
    p = torch.tensor([float(n_units)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(p, op=dist.ReduceOp.SUM)
    return float(t.item()), float(p.item())
