#!/usr/bin/env python3
"""bench.py -- classified atom-pairs/s of the contacts hot path on MI355X (BASELINE.json metric).

One "step" = one full pass of the hot path (uniform-grid build + neighbour search + per-pair classification + compacted
pair table) over synthetic structures already resident in HBM.

  `value`     BASELINE.json configs[3]: the 1e6-atom S2 synthetic cloud at 6.5 A cutoff (tests/synth.py).  With
              --gpus N every rank runs its own same-sized cloud ("scaling": "weak"): the path shards over independent structures,
              there is no data-path collective; torch.distributed only carries the barrier and the max/sum of the timings.
  `s1`        the same pass on the chemistry-faithful S1 cloud of the same size ("the headline run reports both").
  `s2_1e5`, `s1_1e5`   BASELINE.json configs[2]: the 10^5-atom clouds (pairs/s, GB/s, roofline fraction).
  `files`     BASELINE.json configs[0..1]: test-data 1ubq and 6bft through the same pass, resident on the device: us per call on the
              stream, classified pairs, and the rows / warm wall time of the table path (arp_get_contacts) on the same files.
  `batch5k`   BASELINE.json configs[4]: a batch of ~5k-atom S1 structures (atoms ~ N(5000, 500^2) clipped to [3000, 7000]), dealt
              longest-first over the ranks (1250 per rank by default: 10^4 at 8 GPUs), each rank's share packed into one resident
              multi-model SoA; plus, on rank 0, the host-inclusive figure of arp_contacts_atomic_batch on 512 of them (PCIe both ways).
  `--workload batch5k` makes that batch the `value` instead (total size --structures, strong scaling over the ranks).

Prints ONE JSON line (rank 0) with `roofline` and `cpu_baseline`.  `roofline.frac` is SURVEY.md 8(d)'s quantity: the algorithmic bytes
of one step over the device time of the WHOLE launch sequence (grid build + search/classify/emit + probe pass + hole fix-up, HIP events on
the context's stream); `roofline.kernel_frac` is the same bytes over the dominant kernel alone.  `cpu_baseline`: see cpu_baseline()
(the oracle -- this repo's C restatement, "port" -- on a bounded sample, one thread and all host threads; N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)
SEED = 0xA11CE5EED00


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--atoms", type=int, default=1_000_000)
    ap.add_argument("--workload", choices=["s2", "s1", "batch5k"], default="s2")
    ap.add_argument("--structures", type=int, default=0, help="batch5k: structures in the whole job (default 1250 per rank; 10000 = BASELINE config 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="only the headline workload (no s1 / batch5k sub-objects)")
    ap.add_argument("--cpu-sample-atoms", type=int, default=1_000_000,
                    help="size of the one-thread CPU-baseline sample (default: the full 10^6-atom workload, ~10 s)")
    ap.add_argument("--profile-steps", type=int, default=5)
    ap.add_argument("--no-check", action="store_true", help="diagnostic (ablation) builds: do not assert the pair count")
    ap.add_argument("--deterministic", action="store_true", help="two-pass ordered emitter (ARP_FLAG_DETERMINISTIC)")
    ap.add_argument("--residue-runs", choices=["auto", "on", "off"], default="auto",
                    help="diagnostic: force (on) or rule out (off) the residue-rule kernels (ARP_FLAG_RESIDUE_RUNS / ARP_FLAG_NO_RESIDUE_RUNS); "
                         "auto = the engine's memo (a sample of the previous call's atoms), which is what a caller gets")
    ap.add_argument("--contacts-only", action="store_true",
                    help="informational: ARP_FLAG_CONTACTS_ONLY (what the table path runs); `value` then counts emitted contacts, not the headline metric")
    return ap.parse_args()


def to_device(soa, torch, dev):
    import numpy as np

    out = {}
    for k, v in soa.items():
        if v.dtype == np.uint32:
            v = v.view(np.int32)
        out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(dev)
    return out


def pack_soas(soas):
    """Independent structures as ONE multi-model SoA (what a packed batch looks like once resident): models renumbered to be
    distinct, residue tables offset.  The engine never pairs atoms of different models (complex.rs:96-98)."""
    import numpy as np

    none = np.uint32(0xFFFFFFFF)
    cat = {k: [] for k in ("x", "y", "z", "attr", "res_ord", "chain_rank", "model", "res_id", "res_h_ptr", "res_cb", "res_sg", "res_h_idx")}
    a0 = r0 = h0 = m0 = 0
    for s in soas:
        n, nr = len(s["x"]), len(s["res_cb"])
        for k in ("x", "y", "z", "attr", "res_ord", "chain_rank"):
            cat[k].append(s[k])
        cat["model"].append(s["model"].astype(np.uint32) + np.uint32(m0))
        cat["res_id"].append(s["res_id"] + np.uint32(r0))
        cat["res_h_ptr"].append(s["res_h_ptr"][:-1] + np.uint32(h0))
        for k in ("res_cb", "res_sg"):
            cat[k].append(np.where(s[k] == none, none, s[k] + np.uint32(a0)))
        cat["res_h_idx"].append(s["res_h_idx"] + np.uint32(a0))
        a0 += n; r0 += nr; h0 += int(s["res_h_ptr"][-1]); m0 += int(s["model"].max(initial=0)) + 1
    assert m0 < 65536, "a pack holds at most 65535 models (the per-model boxes of the workspace)"
    out = {k: np.concatenate(v) if v else np.zeros(0) for k, v in cat.items()}
    out["res_h_ptr"] = np.concatenate([out["res_h_ptr"], np.asarray([h0], dtype=np.uint32)]).astype(np.uint32)
    return out


_M64 = (1 << 64) - 1
_K1, _K2, _K3 = 0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0xD6E8FEB86659FD93


def record_hash_numpy(i, j, dist_f32, kind) -> int:
    """Order-independent 64-bit hash of a pair list: the wrapping sum over the records of a 64-bit mix of (i, j, distance bits, kind).
    The same arithmetic as record_hash_device, on the oracle's list (numpy uint64 wraps)."""
    import numpy as np

    x = i.astype(np.uint64) | (j.astype(np.uint64) << np.uint64(32))
    y = np.ascontiguousarray(dist_f32, dtype=np.float32).view(np.uint32).astype(np.uint64) | (kind.astype(np.uint64) << np.uint64(32))
    with np.errstate(over="ignore"):
        h = (x * np.uint64(_K1)) ^ (y * np.uint64(_K2))
        h ^= h >> np.uint64(29)
        h *= np.uint64(_K3)
        h ^= h >> np.uint64(32)
        return int(h.sum(dtype=np.uint64))


def record_hash_device(torch, out, n) -> int:
    """record_hash_numpy on the device-resident records out[0:n] (int32 [n, 4] = i, j, distance bits, kind): int64 arithmetic wraps like
    uint64, logical shifts are arithmetic shifts with the sign extension masked off."""
    def s64(v):  # the int64 with the bit pattern of the uint64 v
        return v - (1 << 64) if v >= (1 << 63) else v

    total = 0
    for lo in range(0, n, 1 << 24):  # (in slices: the temporaries of a 3 x 10^7-record list stay below 1 GB)
        r = out[lo:min(n, lo + (1 << 24))].to(torch.int64) & 0xFFFFFFFF
        x = r[:, 0] | (r[:, 1] << 32)
        y = r[:, 2] | (r[:, 3] << 32)
        h = (x * s64(_K1)) ^ (y * s64(_K2))
        h = h ^ ((h >> 29) & ((1 << 35) - 1))
        h = h * s64(_K3)
        h = h ^ ((h >> 32) & 0xFFFFFFFF)
        total = (total + int(h.sum().item())) & _M64
    return total


LAST: dict = {}  # what the most recent measure_resident() saw besides its return values: first_call_ms, pair_hash
WANT_HASH = False  # set by main() when the CPU-baseline leg will compare the records (the hash runs torch kernels: kept out of profiler runs)


def measure_resident(aa, _lib, torch, dev, dev_index, soa, prm, steps, warmup, profile_steps, barrier, check=True):
    """`steps` passes over one device-resident SoA: (wall seconds over the steps, device ms per step, pairs, per-kernel ms).
    LAST receives the device time of the FIRST pass on these arrays (no memo yet: the probe pass is launched, the kernels are the
    default ones) and, after the timed steps, the order-independent hash of the emitted records."""
    dsoa = to_device(soa, torch, dev)
    keep = []
    atoms = aa.atoms_from_arrays(dsoa, location=_lib.ARP_MEM_DEVICE, keep=keep)
    stream = torch.cuda.current_stream(dev)
    if os.environ.get("ARP_BENCH_STRIP_ROWS"):  # diagnostic: the cell rows in y strips of that many rows (arp_debug_set "strip_rows"; default: by input size)
        aa.debug_set("strip_rows", int(os.environ["ARP_BENCH_STRIP_ROWS"]))
    ctx = aa.Context(dev_index, stream=stream.cuda_stream)
    n_pairs = ctx.count(atoms, prm)  # size the output once (count pass), then everything is allocation-free
    cap = max(n_pairs, 1) if check else n_pairs + n_pairs // 16 + 4096  # (diagnostic ablation builds, --no-check: their passes need not agree on the count)
    out = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    LAST.clear()
    f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    f0.record(stream)
    ctx.enqueue(atoms, prm, out.data_ptr(), cap)  # the first pass on these arrays (untimed by the steps below)
    f1.record(stream)
    got = ctx.result()
    LAST["first_call_ms"] = f0.elapsed_time(f1)
    assert not check or got == n_pairs
    for _ in range(warmup):
        ctx.enqueue(atoms, prm, out.data_ptr(), cap)
    got = ctx.result() if warmup else n_pairs
    assert not check or got == n_pairs
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(steps):
        ctx.enqueue(atoms, prm, out.data_ptr(), cap)
    ev1.record(stream)
    barrier()
    wall = time.perf_counter() - t0
    got = ctx.result()
    assert not check or got == n_pairs
    dev_ms = ev0.elapsed_time(ev1) / steps
    if WANT_HASH:
        LAST["pair_hash"] = record_hash_device(torch, out, got)  # what the LAST timed step left in the buffer
    # per-kernel durations (HIP events on the same stream, separate pass so they do not perturb the timed region)
    acc: dict = {}
    if profile_steps:
        ctx.profile(True)
        for _ in range(profile_steps):
            ctx.enqueue(atoms, prm, out.data_ptr(), cap)
            ctx.result()
            for k, v in ctx.profile_read().items():
                acc[k] = acc.get(k, 0.0) + v / profile_steps
        ctx.profile(False)
    del out
    return wall, dev_ms, n_pairs, acc, len(soa["x"])


def measure_two_streams(aa, _lib, torch, dev, dev_index, soa, prm, steps, n_pairs):
    """Throughput of INDEPENDENT calls: the same resident input through two contexts (own stream, workspace and output buffer each), steps
    dealt alternately -- what a caller with a queue of structures does (arp_contacts_atomic_batch alternates two contexts per device for the
    same reason).  One call is a chain of dependent kernels with a draining tail each; a second stream fills those gaps with the next call's
    kernels.  Returns wall ms per step over `steps` steps (both streams drained inside the timed region).  Never the headline `value`."""
    dsoa = to_device(soa, torch, dev)
    keep = []
    atoms = aa.atoms_from_arrays(dsoa, location=_lib.ARP_MEM_DEVICE, keep=keep)
    lanes = []
    for _ in range(2):
        st = torch.cuda.Stream(dev)
        ctx = aa.Context(dev_index, stream=st.cuda_stream)
        out = torch.empty((max(n_pairs, 1), 4), dtype=torch.int32, device=dev)
        for _ in range(3):  # warm: workspace, memo
            ctx.enqueue(atoms, prm, out.data_ptr(), max(n_pairs, 1))
            assert ctx.result() == n_pairs
        lanes.append((st, ctx, out))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(steps):
        st, ctx, out = lanes[k & 1]
        if k >= 2:
            assert ctx.result() == n_pairs  # (the lane's previous call: its result is read before the next one is queued on it)
        ctx.enqueue(atoms, prm, out.data_ptr(), max(n_pairs, 1))
    for st, ctx, out in lanes:
        assert ctx.result() == n_pairs
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    del lanes
    return ms


def roofline_of(n_atoms, n_pairs, acc, dev_ms, traffic=None):
    """algorithmic bytes = 36 B per atom read once + 16 B per classified pair written; t_kernel = grid build + search /
    classify / emit + second-pass kernels.  `frac` / `achieved` are over that whole launch sequence, `kernel_frac` / `kernel_achieved` over
    the dominant kernel alone."""
    if not acc:  # --profile-steps 0 (external profiler runs): fall back to the whole device-side step
        acc = {"pipeline": dev_ms}
    dom = max(acc, key=acc.get)
    alg_bytes = 36.0 * n_atoms + 16.0 * n_pairs
    dom_ms, pipeline_ms = acc[dom], sum(acc.values())
    achieved = alg_bytes / (pipeline_ms * 1e-3) / 1e9
    kernel_achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    return {
        "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
        "scope": "whole launch sequence of one step (SURVEY.md 8d t_kernel): " + " + ".join(acc),
        "algorithmic_bytes": alg_bytes, "pipeline_ms": pipeline_ms,
        "kernel": dom, "kernel_ms": dom_ms, "kernel_achieved": kernel_achieved, "kernel_frac": kernel_achieved / HBM_PEAK_GBS,
        "kernels_ms": acc,
    }


def batch5k_sizes(n_structures: int):
    import numpy as np

    rng = np.random.default_rng(SEED + 5)
    return np.clip(np.rint(rng.normal(5000.0, 500.0, n_structures)), 3000, 7000).astype(int)


def batch5k_share(aa, synth, sizes, mine, pool=48):
    """SoAs of this rank's structures.  Generating 10^4 distinct structures would take minutes of numpy time per run, so `pool` distinct
    S1 structures (sizes spread over the WHOLE batch's size range, the same on every rank) stand in for the rest: structure k is the pool
    member closest in size, whatever rank it lands on, so the job's pair total does not depend on the number of ranks.  Every one is
    still an independent structure of its own model in the pack."""
    import numpy as np

    order = sorted(range(len(sizes)), key=lambda k: sizes[k])
    picks = sorted({order[int(round(q))] for q in np.linspace(0, len(order) - 1, min(pool, len(order)))}, key=lambda k: sizes[k])
    need = {min(picks, key=lambda q: (abs(int(sizes[q]) - int(sizes[k])), q)) for k in mine}
    made = {}
    for k in need:
        rec = synth.gen_s1(int(sizes[k]), seed=SEED + 5 + k)
        made[k] = aa.Structure.from_records(rec, hierarchy=True).soa("/")
    return [made[min(picks, key=lambda q: (abs(int(sizes[q]) - int(sizes[k])), q))] for k in mine]


def host_cpu_share() -> tuple:
    """(threads this process may run at once, how that is known): the cgroup's CPU quota when there is one (a GPU box leases a share of the
    host with its GPU), else the affinity mask."""
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period) + 0.5)), f"cgroup cpu.max {quota}/{period}"
    except (OSError, ValueError):
        pass
    try:
        return len(os.sched_getaffinity(0)), "sched_getaffinity"
    except (AttributeError, OSError):
        return os.cpu_count() or 1, "os.cpu_count"


def cpu_baseline(n_atoms: int, workload: str, seed: int = 0xBA5E, gpu_pairs=None, gpu_hash=None):
    """The oracle ("port": this repo's C restatement, NOT the reference binary) on bounded samples: one thread on the headline
    cloud, then all host threads at once, each on its own independent 10^5-atom cloud (the reference's -j 0, utils.rs:8-30)."""
    import threading

    import oracle_binding as ob
    import synth

    gen = getattr(synth, f"gen_{workload if workload in ('s1', 's2') else 's2'}")
    rec = gen(n_atoms, seed=seed)  # (the headline cloud itself when the sample has its size: the pair counts must then agree)
    s = ob.Structure.from_atoms(synth.records_to_oracle(rec, flat=True), flat=True)
    t0 = time.perf_counter()
    pairs = s.atomic_contacts("/", 0.1, 6.5)
    dt = time.perf_counter() - t0
    share, share_src = host_cpu_share()
    threads = max(1, min(share, os.cpu_count() or 1, 256))  # every host thread this process may use (round 4 capped this at 64 whatever the box)
    small = ob.Structure.from_atoms(synth.records_to_oracle(gen(100_000, seed=0xBA5E + 1), flat=True), flat=True)
    counts = [0] * threads

    def work(k):
        counts[k] = len(small.atomic_contacts("/", 0.1, 6.5))  # (the C call releases the GIL; the structure is only read)

    th = [threading.Thread(target=work, args=(k,)) for k in range(threads)]
    t1 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt_all = time.perf_counter() - t1
    cargo = shutil.which("cargo")
    if cargo:
        try:
            cargo = subprocess.run([cargo, "--version"], capture_output=True, text=True, timeout=20).stdout.strip() or cargo
        except (OSError, subprocess.SubprocessError):
            pass
    return {
        "value": len(pairs) / dt, "unit": "classified atom-pairs/s", "cores": 1, "kind": "port",
        "sample": f"{workload.upper()} synthetic cloud, {n_atoms} atoms -> {len(pairs)} pairs, grid search + per-pair rules, "
                  f"{dt:.1f} s on 1 of {os.cpu_count()} host threads",
        "all_cores": {"value": sum(counts) / dt_all, "cores": threads,
                      "sample": f"{threads} threads = this process's share of the host's {os.cpu_count()} ({share_src}), each one independent 100000-atom cloud "
                                f"({counts[0]} pairs), {dt_all:.1f} s"},
        "reference_toolchain": cargo or "cargo not found on this box: the reference (Rust) cannot be built or timed here",
        "note": "restatement CPU baseline (oracle/arp_oracle.c), not the reference binary",
        **({"same_cloud_as_gpu": True, "pair_count_equals_gpu": bool(len(pairs) == gpu_pairs)} if gpu_pairs is not None else {}),
        # the records themselves: order-independent 64-bit hash over (i, j, f32 distance bits, kind) of the oracle's list against the same hash of
        # what the last timed GPU step left in the output buffer (the timed call path itself, content-checked at the headline size)
        **({"pair_set_hash": f"{record_hash_numpy(pairs['i'], pairs['j'], pairs['dist'].astype('float32'), pairs['kind']):016x}", "pair_set_hash_gpu": f"{gpu_hash:016x}",
            "pair_set_hash_equals_gpu": bool(record_hash_numpy(pairs['i'], pairs['j'], pairs['dist'].astype('float32'), pairs['kind']) == gpu_hash)}
           if gpu_hash is not None and gpu_pairs is not None else {}),
    }


def main():
    args = parse()
    import numpy as np
    import torch

    import arpeggia_amd as aa
    import synth
    from arpeggia_amd import _lib
    from arpeggia_amd.sharding import lpt_assign, reduce_job

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
    red_dev = "cpu" if rehearse else dev

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    global WANT_HASH
    WANT_HASH = world == 1 and not args.no_cpu_baseline
    prm = aa.default_params(0.1, 6.5, deterministic=args.deterministic, contacts_only=args.contacts_only,
                            residue_runs={"auto": None, "on": True, "off": False}[args.residue_runs])
    check = not args.no_check

    recs = {}  # (workload, atoms) -> the generated records (the SAP leg reuses the S1 clouds)

    def cloud(workload, atoms=None):  # one structure per rank, its own seed: independent structures shard with no exchange
        atoms = atoms or args.atoms
        rec = getattr(synth, f"gen_{workload}")(atoms, seed=SEED + (4 if workload == "s2" else 3) + 1000 * rank + (0 if atoms == args.atoms else 7))
        recs[(workload, atoms)] = rec
        order = os.environ.get("ARP_BENCH_ORDER")  # diagnostic: the same cloud with its atoms in another input order
        if order:
            import numpy as np
            n = len(rec["x"])
            if order == "random":
                perm = np.random.default_rng(1).permutation(n)
            else:  # "cell": sorted by 6.5 A cell, x fastest (the order the grid build produces)
                perm = np.lexsort((np.floor(rec["x"] / 6.5), np.floor(rec["y"] / 6.5), np.floor(rec["z"] / 6.5)))
            rec = {k: (v[perm] if isinstance(v, np.ndarray) and len(v) == n else v) for k, v in rec.items()}
        return aa.Structure.from_records(rec, hierarchy=True).soa("/")

    def batch(total):  # this rank's longest-first share of the batch, as one resident pack (packs of <= 65535 models)
        sizes = batch5k_sizes(total)
        mine = lpt_assign([int(v) for v in sizes], world)[rank]
        soas = batch5k_share(aa, synth, sizes, mine)
        packs = [pack_soas(soas[k:k + 60000]) for k in range(0, len(soas), 60000)]
        return packs, len(mine)

    sub = {}
    if args.workload == "batch5k":
        total = args.structures or 1250 * world
        packs, n_mine = batch(total)
        wall = dev_ms = 0.0
        n_pairs = n_atoms = 0
        acc: dict = {}
        for p in packs:  # (one pack unless a rank holds more than 60000 structures)
            w, d, npair, a, na = measure_resident(aa, _lib, torch, dev, dev_index, p, prm, args.steps, args.warmup, args.profile_steps, barrier, check)
            wall += w; dev_ms += d; n_pairs += npair; n_atoms += na
            head = dict(LAST)
            for k, v in a.items():
                acc[k] = acc.get(k, 0.0) + v
        label = (f"batch of {total} S1 structures of ~5k atoms (N(5000, 500^2) clipped to [3000, 7000]), {n_mine} on this rank as one resident multi-model pack; "
                 f"up to 48 distinct generated structures (one pool, the same on every rank) stand in for the rest")
        scaling = "strong" if args.structures else "weak"
    else:
        wall, dev_ms, n_pairs, acc, n_atoms = measure_resident(aa, _lib, torch, dev, dev_index, cloud(args.workload), prm, args.steps, args.warmup,
                                                                args.profile_steps, barrier, check)
        label = f"{args.workload.upper()} synthetic {n_atoms}-atom cloud per GPU (tests/synth.py gen_{args.workload})"
        head = dict(LAST)
        if rank == 0 and not args.no_extras and not args.deterministic:  # informational: two independent calls in flight on two streams
            ms2 = measure_two_streams(aa, _lib, torch, dev, dev_index, cloud(args.workload), prm, max(args.steps, 20), n_pairs)
            sub["two_streams"] = {"workload": "the headline cloud through TWO contexts (own stream / workspace / output buffer each), calls dealt alternately: "
                                              "throughput of independent calls, never the `value`", "ms_per_step": ms2, "value": n_pairs / (ms2 * 1e-3),
                                  "unit": "classified atom-pairs/s"}
        scaling = "weak"
        if not args.no_extras and not args.deterministic and not args.contacts_only:
            other = "s1" if args.workload == "s2" else "s2"
            w2, d2, p2, a2, n2 = measure_resident(aa, _lib, torch, dev, dev_index, cloud(other), prm, args.steps, args.warmup, args.profile_steps, barrier, check)
            w2max, p2all = reduce_job(dist, red_dev, w2, p2)
            sub[other] = {"workload": f"{other.upper()} synthetic {n2}-atom cloud per GPU", "atoms_per_gpu": n2, "pairs_per_gpu": p2,
                          "value": p2all * args.steps / w2max, "ms_per_step": w2max / args.steps * 1e3, "first_call_ms": LAST.get("first_call_ms"), "roofline": roofline_of(n2, p2, a2, d2)}
            # BASELINE config 3: the 10^5-atom clouds (rank 0 only: they are small, and the multi-rank job is about configs 4 and 5)
            if rank == 0:
                for wl in ("s2", "s1"):
                    w3, d3, p3, a3, n3 = measure_resident(aa, _lib, torch, dev, dev_index, cloud(wl, 100_000), prm, max(args.steps, 50), args.warmup, args.profile_steps,
                                                          lambda: torch.cuda.synchronize(dev), check)
                    sub[f"{wl}_1e5"] = {"workload": f"{wl.upper()} synthetic {n3}-atom cloud, 1 GPU (BASELINE config 3)", "atoms_per_gpu": n3, "pairs_per_gpu": p3,
                                        "value": p3 * max(args.steps, 50) / w3, "unit": "classified atom-pairs/s", "ms_per_step": w3 / max(args.steps, 50) * 1e3,
                                        # the first pass on these arrays runs the chunked sequence with its fix-up and probe launch; the timed (repeated) steps the
                                        # scratch-staged hole-free one (config.speculation)
                                        "first_call_ms": LAST.get("first_call_ms"), "roofline": roofline_of(n3, p3, a3, d3)}
                # BASELINE configs 1-2: the reference's own test files, resident on the device (launch-bound: microseconds, not a roofline)
                files = {}
                for name in ("1ubq", "6bft"):
                    path = str(ROOT / "tests" / "data" / f"{name}.pdb")
                    st = aa.load_model(path)
                    wf, df, pf, af, nf = measure_resident(aa, _lib, torch, dev, dev_index, st.soa("/"), prm, 200, 10, args.profile_steps, lambda: torch.cuda.synchronize(dev), check)
                    cf = aa.Context(dev_index)
                    table = cf.get_contacts(st, "/", 0.1, 6.5)  # first call: uploads the resident copy, fits planes, ranks entities
                    best = None
                    for _ in range(20):
                        t0 = time.perf_counter()
                        table = cf.get_contacts(st, "/", 0.1, 6.5)
                        dt = time.perf_counter() - t0
                        best = dt if best is None else min(best, dt)
                    # the same call at the C boundary (arp_get_contacts + arp_table_free: what a Rust binder pays; the Python figure above adds the
                    # extraction of twenty numpy columns, ten of them strings)
                    import ctypes as C
                    best_c = None
                    for _ in range(20):
                        tp = C.c_void_p()
                        t0 = time.perf_counter()
                        stc = _lib.lib.arp_get_contacts(cf._h, st._h, b"/", 0.1, 6.5, C.byref(tp))
                        dt = time.perf_counter() - t0
                        assert stc == 0, _lib.lib.arp_last_error()
                        _lib.lib.arp_table_free(tp)
                        best_c = dt if best_c is None else min(best_c, dt)
                    files[name] = {"atoms": nf, "pairs": pf, "us_per_call_on_stream": df * 1e3, "us_per_call_wall": wf / 200 * 1e6, "kernels_us": {k: v * 1e3 for k, v in af.items()},
                                   "table_rows": int(len(table["model"])), "get_contacts_warm_us": best * 1e6, "get_contacts_c_abi_warm_us": best_c * 1e6}
                sub["files"] = {"workload": "tests/data/1ubq.pdb and 6bft.pdb (= the reference's test-data), groups='/', vdw_comp=0.1, dist_cutoff=6.5: the pair pass on "
                                            "device-resident arrays (200 calls on the stream) and the whole table (arp_get_contacts, best of 20 warm calls, host wall)", **files}
            if rank == 0:  # the SAP neighbour sum on the same S1 clouds (device time of the grid build + sum kernel per call)
                import sap_timing

                sap_ctx = aa.Context(dev_index)
                sub["sap"] = {"workload": "arp_sap_neighbor_sum on the S1 clouds, radius 5 A, side-chain atoms only in the grid; per-kernel HIP events, host staging "
                                          "and PCIe excluded; algorithmic bytes 36 B read + 4 B written per side-chain atom",
                              **{f"s1_{a}": sap_timing.measure(sap_ctx, a, rec=recs.get(("s1", a))) for a in (100_000, args.atoms)}}
                del sap_ctx
            packs, n_mine = batch(1250 * world)
            wb, db, pb, ab, nb = measure_resident(aa, _lib, torch, dev, dev_index, packs[0], prm, args.steps, args.warmup, args.profile_steps, barrier, check)
            wbmax, pball = reduce_job(dist, red_dev, wb, pb)
            sub["batch5k"] = {"workload": f"BASELINE config 5 shape: {1250 * world} S1 structures of ~5k atoms over {world} GPU(s), longest-first deal, "
                                          f"{n_mine} on rank 0 as one resident multi-model pack; up to 48 distinct generated structures (one pool, the same on every rank) stand in "
                                          f"for the rest (each still its own model of the pack)", "structures": 1250 * world, "atoms_per_gpu": nb, "pairs_per_gpu": pb,
                              "value": pball * args.steps / wbmax, "unit": "classified atom-pairs/s", "ms_per_step": wbmax / args.steps * 1e3,
                              "us_per_structure": wbmax / args.steps / max(n_mine, 1) * 1e6, "roofline": roofline_of(nb, pb, ab, db)}
            if rank == 0:  # the host-inclusive path of the same shape: host arrays in, host pair lists out, packed launches (PCIe both ways)
                import ctypes as C

                sizes = batch5k_sizes(512)
                soas = batch5k_share(aa, synth, sizes, list(range(512)), pool=16)
                keep_host = []
                views = [aa.atoms_from_arrays(s, keep=keep_host) for s in soas]
                arr = (C.POINTER(_lib.arp_atoms) * len(views))(*[C.pointer(v) for v in views])
                c2, c3 = aa.Context(dev_index), aa.Context(dev_index)
                host = {}
                for name, only, n_ctx in (("contacts_only", True, 1), ("all_candidates", False, 1), ("contacts_only_2ctx", True, 2), ("all_candidates_2ctx", False, 2)):
                    # n_ctx = 2: two contexts on this one device stand in for two GPUs -- the in-process longest-first deal over devices, timed
                    handles = (C.c_void_p * n_ctx)(*[c._h for c in (c2, c3)[:n_ctx]])
                    hp = aa.default_params(0.1, 6.5, contacts_only=only)
                    outs = (_lib.arp_pairs * len(views))()
                    its = []
                    for _ in range(6):  # (the first calls of a leg allocate -- peer contexts, pinned blocks of the leg's sizes --: 3 iterations did not always get past them)
                        t0 = time.perf_counter()
                        st = _lib.lib.arp_contacts_atomic_batch(handles, n_ctx, arr, len(views), C.byref(hp), outs)
                        dt = time.perf_counter() - t0
                        assert st == 0, _lib.lib.arp_last_error()
                        n_out = sum(int(outs[k].n) for k in range(len(views)))
                        for k in range(len(views)):
                            _lib.lib.arp_pairs_free(C.byref(outs[k]))
                        its.append(dt)
                    host[name] = {"us_per_structure": min(its) / len(views) * 1e6, "us_per_structure_median": sorted(its)[len(its) // 2] / len(views) * 1e6,
                                  "records_out": n_out, "contexts": n_ctx}
                sub["batch5k"]["host_path"] = {"structures": len(views), "note": "arp_contacts_atomic_batch: pageable host arrays in, host pair lists out "
                                               "(PCIe both ways, never the `value`); 16 distinct generated structures stand in for the 512; *_2ctx: two contexts on the one device",
                                               **host}

    wall_max, pairs_all = reduce_job(dist, red_dev, wall, n_pairs)  # max over ranks / sum over ranks; no data-path collective

    if rank == 0:
        # HBM traffic of the dominant kernel cannot be counted from inside this process; when the run matches the configuration
        # the committed rocprofv3 --pmc passes were taken on, report that measurement (profiles/, with its source), else null.
        # ... and ONLY when the profile was taken on the sources this library was built from (profiles/rNN_traffic.json carries the content
        # hash of arpeggia_amd/csrc, tests/pmc_to_json.py): a stale profile is reported as null with the reason, never attached silently.
        traffic, traffic_note, issue = None, None, None
        try:
            from arpeggia_amd import build as _build
            running_hash = _build.source_hash()
        except Exception as e:  # noqa: BLE001
            running_hash = None; traffic_note = f"source hash of the running library unavailable: {e}"
        newest = sorted((ROOT / "profiles").glob("r*_traffic.json"), reverse=True)
        for path in newest:
            try:
                tr = json.loads(path.read_text())
                if tr["workload"] != args.workload or tr["atoms"] != n_atoms or args.deterministic or args.contacts_only:
                    continue
                if running_hash is None:
                    break
                if tr.get("csrc_hash") != running_hash:
                    traffic_note = (f"{path.name} was taken on other sources (csrc hash {str(tr.get('csrc_hash'))[:12]}.. != running {running_hash[:12]}..): "
                                    f"not attached; re-run tests/run_gpu_pmc.sh + tests/pmc_to_json.py")
                    break
                traffic = {"hbm_bytes_per_launch": tr["hbm_bytes_per_launch"], "kernel": tr["kernel"], "source": tr["source"], "csrc_hash": tr["csrc_hash"]}
                issue = dict(tr.get("issue") or {}, source=tr["source"], csrc_hash=tr["csrc_hash"]) or None
                break
            except (OSError, KeyError, ValueError):
                pass
        if traffic is None and traffic_note is None:
            traffic_note = "no committed counter profile for this workload / size / emitter"
        line = {
            "metric": "classified atom-pairs/s per GPU at 6.5 A cutoff; achieved HBM GB/s vs peak",
            "value": pairs_all * args.steps / wall_max,
            "unit": "classified atom-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"{label}, groups='/', vdw_comp=0.1, dist_cutoff=6.5",
                "atoms_per_gpu": n_atoms, "pairs_per_gpu": n_pairs, "pairs_all_gpus": pairs_all, "sharding": "independent structures per rank, no collective",
                "emitter": ("ordered two-pass" if args.deterministic else "single-pass") + (", contacts only (kind != 0)" if args.contacts_only else ""),
                # what the timed steps (2..N on the same arrays) do not pay and a first call does: see first_call_ms
                "speculation": "none (ordered emitter)" if args.deterministic else
                               "steps 2..N on the same device arrays skip the launch of the EMPTY probe pass (memo of the previous call, validated on the device by k_fixup); "
                               "the emit kernels are picked from a sample of the previous call's atoms (residue-rule kernels for inputs whose residues are runs of atoms); "
                               "under the same memo inputs of 20 480 .. ~131 000 atoms (the 10^5-atom legs) take the scratch-staged hole-free sequence, which has no fix-up launch",
            },
            "first_call_ms": head.get("first_call_ms"),  # device time of the first pass on these arrays: probe pass launched, default kernels
            "roofline": roofline_of(n_atoms, n_pairs, acc, dev_ms, traffic),
            "device_ms_per_step": dev_ms,
        }
        # what actually binds: the graded bound stays HBM, `roofline.issue` says how far instruction issue is from idle --
        # the counters of the dominant kernel (same provenance rule as `traffic`), or the reason they are missing
        line["roofline"]["issue"] = issue
        if traffic_note:
            line["roofline"]["traffic_note"] = traffic_note
        line.update(sub)
        if world == 1 and not args.no_cpu_baseline:
            wl = args.workload if args.workload in ("s1", "s2") else "s2"
            same = args.cpu_sample_atoms == args.atoms and args.workload in ("s1", "s2") and not os.environ.get("ARP_BENCH_ORDER")
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample_atoms, wl, SEED + (4 if wl == "s2" else 3) if same else 0xBA5E,
                                                n_pairs if same and not args.contacts_only else None, head.get("pair_hash") if same and not args.contacts_only else None)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
