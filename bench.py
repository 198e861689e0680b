#!/usr/bin/env python3
"""bench.py -- classified atom-pairs/s of the contacts hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path (uniform-grid build + neighbour search + per-pair classification + compacted
pair table) over one synthetic structure already resident in HBM.  Default workload = BASELINE.json configs[3]: the
1e6-atom synthetic cloud at 6.5 A cutoff (generator S2 of SURVEY.md 8d, tests/synth.py).  With --gpus N every rank runs the
same-sized structure (its own seed): the path shards over independent structures, there is no data-path collective, and
torch.distributed is used only for the barrier and the max/sum of the timings ("scaling": "weak").

Prints ONE JSON line (rank 0) including `roofline` (dominant kernel, HIP-event timed) and `cpu_baseline` (the oracle,
timed on a bounded sample on this box's host cores; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--atoms", type=int, default=1_000_000)
    ap.add_argument("--workload", choices=["s2", "s1"], default="s2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-atoms", type=int, default=1_000_000,
                    help="size of the CPU-baseline sample (default: the full 10^6-atom workload, ~10 s on one host thread)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-check", action="store_true", help="diagnostic (ablation) builds: do not assert the pair count")
    ap.add_argument("--deterministic", action="store_true", help="two-pass ordered emitter (ARP_FLAG_DETERMINISTIC)")
    ap.add_argument("--contacts-only", action="store_true",
                    help="informational: ARP_FLAG_CONTACTS_ONLY (what the table path runs); `value` then counts emitted contacts, not the headline metric")
    return ap.parse_args()


def to_device(soa, torch, dev):
    import numpy as np

    out = {}
    for k, v in soa.items():
        if v.dtype == np.uint16:
            v = v.view(np.int16)
        elif v.dtype == np.uint32:
            v = v.view(np.int32)
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    return out


def cpu_baseline(n_atoms: int, workload: str):
    """The oracle ("port": this repo's C restatement, not the reference binary) on a bounded sample, one thread."""
    import oracle_binding as ob
    import synth

    rec = getattr(synth, f"gen_{workload}")(n_atoms, seed=0xBA5E)
    s = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    t0 = time.perf_counter()
    pairs = s.atomic_contacts("/", 0.1, 6.5)
    dt = time.perf_counter() - t0
    return {
        "value": len(pairs) / dt, "unit": "classified atom-pairs/s", "cores": 1, "kind": "port",
        "sample": f"{workload.upper()} synthetic cloud, {n_atoms} atoms -> {len(pairs)} pairs, grid search + per-pair rules, "
                  f"{dt:.1f} s on 1 of {os.cpu_count()} host threads",
    }


def main():
    args = parse()
    import numpy as np
    import torch

    import arpeggia_amd as aa
    import synth
    from arpeggia_amd import _lib
    from arpeggia_amd.sharding import reduce_job

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # ARP_BENCH_REHEARSE=1: all ranks share cuda:0 and rendezvous over gloo -- exercises this multi-process path on a one-GPU box
    # (the driver's real N > 1 runs use one GPU per rank over RCCL)
    rehearse = os.environ.get("ARP_BENCH_REHEARSE") == "1"
    dev_index = 0 if rehearse else local_rank
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # ---- synthetic structure of this rank (independent structures shard with no exchange) ----
    seed = 0xA11CE5EED00 + 4 + 1000 * rank
    rec = getattr(synth, f"gen_{args.workload}")(args.atoms, seed=seed)
    st = aa.Structure.from_records(rec, hierarchy=True)
    soa = st.soa("/")
    n = len(soa["x"])
    dsoa = to_device(soa, torch, dev)
    keep = []
    atoms = aa.atoms_from_arrays(dsoa, location=_lib.ARP_MEM_DEVICE, keep=keep)
    prm = aa.default_params(0.1, 6.5, deterministic=args.deterministic, contacts_only=args.contacts_only)
    stream = torch.cuda.current_stream(dev)
    ctx = aa.Context(dev_index, stream=stream.cuda_stream)

    # size the output once (count pass), then everything is allocation-free
    n_pairs = ctx.count(atoms, prm)
    cap = max(n_pairs, 1)  # (diagnostic ablation builds may report no pairs: still run the emit path)
    out = torch.empty((cap, 4), dtype=torch.int32, device=dev)

    def step():
        ctx.enqueue(atoms, prm, out.data_ptr(), cap)

    for _ in range(args.warmup):
        step()
    got = ctx.result()
    assert args.no_check or got == n_pairs

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    got = ctx.result()
    assert args.no_check or got == n_pairs
    dev_ms = ev0.elapsed_time(ev1)

    # ---- per-kernel durations (HIP events on the same stream, separate pass so they do not perturb the timed region) ----
    ctx.profile(True)
    acc: dict = {}
    for _ in range(args.profile_steps):
        step()
        ctx.result()
        for k, v in ctx.profile_read().items():
            acc[k] = acc.get(k, 0.0) + v / args.profile_steps
    ctx.profile(False)

    wall_max, pairs_all = reduce_job(dist, "cpu" if rehearse else dev, wall, n_pairs)  # max over ranks / sum over ranks; no data-path collective

    if rank == 0:
        # HBM traffic of the dominant kernel cannot be counted from inside this process; when the run matches the configuration
        # the committed rocprofv3 --pmc passes were taken on, report that measurement (profiles/, with its source), else null.
        traffic = None
        try:
            tr = json.loads((ROOT / "profiles" / "r01_traffic.json").read_text())
            if tr["workload"] == args.workload and tr["atoms"] == n and not args.deterministic and not args.contacts_only:
                traffic = {"hbm_bytes_per_launch": tr["hbm_bytes_per_launch"], "kernel": tr["kernel"], "source": tr["source"]}
        except (OSError, KeyError, ValueError):
            pass
        if not acc:  # --profile-steps 0 (external profiler runs): fall back to the whole device-side step
            acc = {"pipeline": dev_ms / args.steps}
        dom = max(acc, key=acc.get)
        alg_bytes = 36.0 * n + 16.0 * n_pairs  # 36 B/atom read once + 16 B per classified pair written
        dom_ms = acc[dom]
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
        pipeline_ms = sum(acc.values())
        line = {
            "metric": "classified atom-pairs/s per GPU at 6.5 A cutoff; achieved HBM GB/s vs peak",
            "value": pairs_all * args.steps / wall_max,
            "unit": "classified atom-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{args.workload.upper()} synthetic {n}-atom cloud per GPU (tests/synth.py gen_{args.workload}), groups='/', vdw_comp=0.1, dist_cutoff=6.5",
                "atoms_per_gpu": n, "pairs_per_gpu": n_pairs, "sharding": "one independent structure per rank, no collective",
                "emitter": ("ordered two-pass" if args.deterministic else "single-pass") + (", contacts only (kind != 0)" if args.contacts_only else ""),
            },
            "roofline": {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes": alg_bytes, "kernel_ms": dom_ms,
                "pipeline_ms": pipeline_ms, "pipeline_frac": alg_bytes / (pipeline_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "kernels_ms": acc,
            },
            "device_ms_per_step": dev_ms / args.steps,
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_atoms, args.workload)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
