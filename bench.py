#!/usr/bin/env python3
"""bench.py — full-swarm match throughput on MI355X (one rank per GPU).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 100k tasks x 10k workers per GPU, mixed gpu/mem/storage/cpu
constraints, seeded synthetic swarm (protocol_amd/swarm.py), tables resident in HBM before the timed
region.  One step = one full-swarm match from a cold group state through the C ABI (pm_tick):
compat sweep (W x C) -> greedy proximity carve -> solo merge -> T x W pair sweep + claim -> published
assignment table.  For N > 1 the swarm is N x 10k workers hash-sharded by address (splitmix64(address)
% N), tasks replicated, each shard an independent carve domain, and every step ends with the one
exchange the path has: an RCCL all-gather of the per-worker task table shards.

value = whole-job task x worker pair-evals per second = N * T * W_local * K / t  (max over ranks).
The JSON line also carries:
  roofline     — the dominant kernel (carve_kernel) against the HBM roofline with the algorithmic-bytes
                 convention of SURVEY.md §8(d) (W_remaining*20 + 8 bytes per carve step), duration from
                 hipEvents recorded around the launches on the engine's own stream;
  kernels      — the same for the compat sweep and the pair sweep (+ VALU ceiling for the sweep);
  cpu_baseline — the C oracle (a port of the reference path; oracle/) timed on this box's host cores,
                 rank 0, N=1 only, on a stated sample.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_LANE_OPS = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz lane-ops/s


def shard_swarm(sw, rank: int, world: int):
    """hash-shard the workers: shard = splitmix64(address) % world (protocol_amd/dist.py)."""
    from protocol_amd.dist import shard_of
    if world == 1:
        return sw, np.arange(sw.W)
    idx = np.nonzero(shard_of(sw.address, world) == rank)[0]
    import copy
    s = copy.copy(sw)
    for k in ("address", "status", "has_p2p", "has_specs", "has_gpu", "gpu_count_some", "gpu_mem_some",
              "gpu_model_some", "has_cpu", "cpu_cores_some", "ram_some", "storage_some", "gpu_count", "gpu_mem_mb",
              "gpu_model_id", "cpu_cores", "ram_mb", "storage_gb", "price", "has_loc", "lat", "lon"):
        setattr(s, k, getattr(sw, k)[idx])
    return s, idx


def cpu_baseline(sw, budget_s: float = 30.0) -> dict:
    """Time the oracle (kind "port": C restatement of the reference path, single thread) on this host."""
    from oracle import oracle_ffi as orc
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=True)
    t0 = time.perf_counter()
    masks = orc.compat_masks(nodes, cfgs)
    t_compat = time.perf_counter() - t0
    t0 = time.perf_counter()
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    t_carve = time.perf_counter() - t0
    cfg_of_node = np.full(sw.W, -1, dtype=np.int32)
    for _, _, c, mem, _ in st.groups():
        cfg_of_node[mem] = c
    # sweep sample: as many tasks as fit the remaining budget (the sweep is linear in T)
    t_s = max(1000, min(sw.T, 10000))
    t0 = time.perf_counter()
    orc.pair_sweep_per_worker(tasks[:t_s], cfgs, cfg_of_node)
    dt = time.perf_counter() - t0
    remaining = max(budget_s - t_carve - t_compat - dt, 0.0)
    t_more = int(min(sw.T - t_s, remaining / max(dt / t_s, 1e-9)))
    if t_more > 0:
        t0 = time.perf_counter()
        orc.pair_sweep_per_worker(tasks[t_s:t_s + t_more], cfgs, cfg_of_node)
        dt += time.perf_counter() - t0
        t_s += t_more
    t_sweep_full = dt * sw.T / t_s
    t_full = t_compat + t_carve + t_sweep_full
    # the same match with every host core on the one phase that parallelises in the reference (heartbeats of
    # different nodes are independent requests); group formation is sequential there and stays so here
    n_thr = os.cpu_count() or 1
    t_mt = min(sw.T, 20000)
    t0 = time.perf_counter()
    orc.pair_sweep_per_worker(tasks[:t_mt], cfgs, cfg_of_node, threads=n_thr)
    t_sweep_mt = (time.perf_counter() - t0) * sw.T / t_mt
    t_full_mt = t_compat + t_carve + t_sweep_mt
    all_cores = {"value": sw.T * sw.W / t_full_mt, "unit": "pair-evals/s", "cores": n_thr,
                 "sample": (f"pair sweep on {n_thr} threads ({t_mt} of {sw.T} tasks, scaled: {t_sweep_mt:.2f} s) + the "
                            f"sequential compat/carve/merge of the 1-thread run ({t_compat + t_carve:.2f} s)"),
                 "seconds_full_match_est": t_full_mt}
    return {
        "value": sw.T * sw.W / t_full, "unit": "pair-evals/s", "cores": 1, "kind": "port",
        "sample": (f"oracle/pm_oracle.c -O2, 1 thread of {os.cpu_count()} host cores: compat sweep + reference-shaped "
                   f"carve+merge on all {sw.W} workers ({t_compat + t_carve:.2f} s) + pair sweep on {t_s} of {sw.T} "
                   f"tasks ({dt:.2f} s, scaled linearly to {t_sweep_full:.2f} s); Redis/JSON costs of the real "
                   f"reference excluded"),
        "seconds_full_match_est": t_full,
        "all_cores": all_cores,
    }


def pmc_traffic(kernel: str):
    """HBM bytes per full-swarm match from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json,
    FETCH_SIZE and WRITE_SIZE collected in separate runs of this same command); None if absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(kernel, {}).get("hbm_bytes_per_match")
    except Exception:
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=21)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs index (default 1: 100k x 10k)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sweep-variant", type=int, default=0)
    ap.add_argument("--carve-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check", action="store_true", help="also verify the groups against the oracle (slow)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (the engine has no CPU fallback)", file=sys.stderr)
        return 2
    # PM_BENCH_BACKEND=gloo + PM_BENCH_SHARE_DEVICE=1: plumbing check of the N>1 path on a 1-GPU box (all ranks
    # on device 0, collectives on host tensors).  The real path is nccl (= RCCL over xGMI), one GPU per rank.
    backend = os.environ.get("PM_BENCH_BACKEND", "nccl")
    if os.environ.get("PM_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")
    if world > 1:
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from protocol_amd import engine as E, host
    from protocol_amd.swarm import baseline_config, make_swarm

    # global swarm: weak scaling — per-GPU work is fixed (tasks replicated, W_local ~ 10k workers per rank)
    if args.config == 1:
        sw_global = make_swarm(args.seed, 100_000, 10_000 * world)
        workload = f"100k tasks x {10_000 * world} workers, mixed gpu/mem/storage/cpu constraints (BASELINE configs[1])"
    else:
        sw_global = baseline_config(args.config, seed=args.seed)
        workload = f"BASELINE configs[{args.config}]"
    sw, shard_idx = shard_swarm(sw_global, rank, world)

    eng = E.Engine(device=local_rank, sweep_variant=args.sweep_variant, carve_variant=args.carve_variant,
                   group_id_seed=args.seed + rank)
    host.load_swarm(eng, sw)
    if world > 1:
        w_counts = [torch.zeros(1, dtype=torch.int64, device=coll_dev) for _ in range(world)]
        dist.all_gather(w_counts, torch.tensor([sw.W], dtype=torch.int64, device=coll_dev))
        w_counts = [int(x.item()) for x in w_counts]
        w_max = max(w_counts)
        gather_out = torch.empty(world * w_max, dtype=torch.int32, device=coll_dev)
        local_tbl = torch.full((w_max,), -1, dtype=torch.int32, device=coll_dev)
    else:
        w_counts = [sw.W]

    class _DevCol:  # torch view of the engine's device-resident task column (no copy)
        def __init__(self, ptr, n):
            self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3}

    def step_fast():
        eng.reset_groups()
        s = eng.tick()
        if world > 1:  # the one exchange step: all-gather the published table shards over RCCL/xGMI
            ptr, n = eng.device_task_column()
            local_tbl[:n].copy_(torch.as_tensor(_DevCol(ptr, n), device=dev))   # stays on the GPU for nccl
            dist.all_gather_into_tensor(gather_out, local_tbl)
        return s

    for _ in range(args.warmup):
        step_fast()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step_fast())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the gathered table must hold every shard's published column
        got = gather_out.view(world, w_max).cpu().numpy()
        ptr, n = eng.device_task_column()
        mine = torch.as_tensor(_DevCol(ptr, n), device=dev).cpu().numpy()
        assert (got[rank, :n] == mine).all(), "all-gathered table does not contain this rank's column"

    T = sw.T
    total_pairs = float(T) * float(sum(w_counts)) * args.steps
    value = total_pairs / elapsed
    ms = [s["ms_total"] for s in stats]
    med = lambda k: statistics.median(s[k] for s in stats)

    # ---- roofline of the dominant kernel (carve) with the algorithmic-bytes convention
    carve_ms = med("ms_carve_kernel")
    carve_bytes = 20.0 * med("carve_cand_sum") + 8.0 * med("carve_steps")
    carve_gbs = carve_bytes / (carve_ms * 1e-3) / 1e9 if carve_ms > 0 else 0.0
    n_cfgs, n_alts = len(sw.configs), sum(1 for c in sw.configs if c[3]) * 2
    compat_bytes = sw.W * 32 + n_cfgs * 32 + n_alts * 32 + sw.W * 8
    sweep_bytes = T * 16 + sw.W * 16 + sw.W * 8          # per-worker orientation: (16T + 24W)
    compat_ms, sweep_ms = med("ms_compat_kernel"), med("ms_sweep_kernel")
    kernels = {
        "compat_kernel": {"ms": compat_ms, "alg_bytes": compat_bytes,
                          "GB/s": compat_bytes / (compat_ms * 1e-3) / 1e9 if compat_ms > 0 else None},
        "pair_sweep": {"ms": sweep_ms, "alg_bytes": sweep_bytes,
                       "GB/s": sweep_bytes / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else None,
                       "pair_evals_per_s": T * sw.W / (sweep_ms * 1e-3) if sweep_ms > 0 else None},
        "carve_kernel": {"ms": carve_ms, "alg_bytes": carve_bytes, "GB/s": carve_gbs,
                         "steps": med("carve_steps"), "fast_steps": med("carve_fast_steps"),
                         "us_per_step": 1e3 * carve_ms / max(med("carve_steps"), 1)},
    }
    out = {
        "metric": "task x worker pair-evals/sec (full-swarm match)", "value": value, "unit": "pair-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "tasks": T, "workers_per_gpu": sw.W, "workers_total": sum(w_counts),
                   "configs": n_cfgs, "seed": args.seed, "sharding": "hash(address) % n_gpus" if world > 1 else "none",
                   "sweep_variant": args.sweep_variant, "carve_variant": args.carve_variant},
        "p50_match_latency_ms": statistics.median(ms),
        "match_latency_ms": {"min": min(ms), "p50": statistics.median(ms), "max": max(ms)},
        "phase_ms_p50": {k: med(k) for k in ("ms_compat", "ms_carve", "ms_merge", "ms_sweep", "ms_publish")},
        "groups": int(stats[-1]["n_groups"]), "host_resolved_steps": int(stats[-1]["host_resolved_steps"]),
        "roofline": {"bound": "hbm", "kernel": "carve (carve_propose_kernel + carve_kernel launch sequence)",
                     "achieved": carve_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": carve_gbs / HBM_PEAK_GBS,
                     "traffic": pmc_traffic("carve"),
                     "note": "dependent chain of ~2k carve steps: latency-bound by construction, see DESIGN.md §6"},
        "kernels": kernels,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # PCIe-inclusive rate, reported beside (never as) `value`: the same match when the worker and task
        # columns arrive as host buffers through the C ABI (pm_upload_workers + pm_upload_tasks) every time
        packed = host.pack_workers(sw)
        tmasks = sw.task_masks()
        t_up = []
        for _ in range(5):
            torch.cuda.synchronize()
            u0 = time.perf_counter()
            eng.upload_workers(packed)
            eng.upload_tasks(tmasks, sw.created_at, sw.task_uid)
            eng.reset_groups()
            eng.tick()
            t_up.append(time.perf_counter() - u0)
        up = sorted(t_up)[len(t_up) // 2]
        out["pcie_inclusive"] = {"ms_per_match": up * 1e3, "value": float(sw.T) * float(sw.W) / up,
                                 "unit": "pair-evals/s",
                                 "note": "host SoA columns -> HBM (workers + tasks) + match, p50 of 5"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sw)
    else:
        out["cpu_baseline"] = None
    if args.check and rank == 0:
        from oracle import oracle_ffi as orc
        nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
        st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False,
                       group_id_seed=args.seed + rank)
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        _, groups, members = eng.get_groups()
        got = [(int(g["id"]), int(g["config"]),
                members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])].tolist()) for g in groups]
        out["parity_vs_oracle"] = got == [(gid, c, m) for (_s, gid, c, m, _t) in st.groups()]
    if rank == 0:
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
