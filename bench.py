#!/usr/bin/env python3
"""bench.py — full-swarm match throughput on MI355X (one rank per GPU).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One step = one full-swarm match from a cold group state through the C ABI: compat sweep (W x C) -> greedy
proximity carve -> solo merge -> T x W pair sweep + claim -> published assignment table, tables resident in HBM
before the timed region.   value = task x worker pair-evals per second = T * W * K / t  (max over ranks).

N = 1   BASELINE.json configs[1]: 100k tasks x 10k workers, mixed gpu/mem/storage/cpu constraints (pm_tick).
        The line also carries, each from a few extra steps:
          configs2   configs[2] (1M tasks x 100k workers, Zipf): ms per match, carve / proposer / sweep split, the
                     proposer against the FP64 vector peak, the sweep against the VALU ceiling
          churn      configs[4] on one GPU: 100k workers, per tick +10k tasks (pm_tasks_insert_front), 1 % of the
                     workers leave (pm_on_worker_status) and 1 % brand-new ones join (pm_append_workers)
          per_task   the north_star orientation (pm_match_per_task) timed
          pools_on_one_gpu  2 / 4 independent configs[1] pools matched concurrently on the one GPU (aggregate rate
                     and per-match latency): one match is a latency chain on one CU, so pools overlap
          roofline   the dominant kernel sequence (carve) with SURVEY section 8(d)'s algorithmic bytes against HBM —
                     and, because the carve is a dependent chain, `chain`: achieved time per step against a floor
          cpu_baseline  the C oracle (a port of the reference path) on this box's host cores: reference-shaped on
                     one thread, the pair sweep on all cores, and a best-effort CPU variant
N > 1   REPLICAS (weak scaling): one configs[1] pool per rank (seed + rank), the loop of the N = 1 line on every GPU, no
        data-path collective — value = N x T x W x K / (max-over-ranks time between barriers).  The reference runs one
        pool per orchestrator process and its carve is a chain of dependent steps that N GPUs do not divide (DESIGN.md
        section 7: "replicas only" for the dependent stream), so N GPUs serve N pools.
        Beside it, `dist.one_pool_sharded`: BASELINE configs[3] — ONE 1M x 100k pool matched by all N ranks (strong
        scaling): workers owned by hash(address) % N, the carve REPLICATED (one streaming launch per rank, nothing
        exchanged), the pair sweep + claim over the owned workers, ONE RCCL all-gather of the published rows per tick
        (protocol_amd/dist.py; the compiled hosts: GpuMatchPlugin::tick_dist).  Every rank ends with identical groups
        (checked by digest).  With it: one_gpu_same_workload (the same swarm on one unsharded engine) and the
        replicated / sharded / exchange split (the Amdahl bound of that design, ~1.1x by construction).
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

HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9    # 256 CUs x 4 SIMDs x 16 lanes/clk x 2.4 GHz = lane-ops per second
OPS_PER_PAIR_EVAL = 20.0                # ~20 integer ops per (task, worker) predicate + fold
FP64_VECTOR_TFLOPS = 78.6               # half the FP32 vector rate (157.3 TFLOPS spec)
FLOP_PER_KEY = 9.0                      # chord form of the Haversine term: 3 subtractions, 1 multiply, 2 fma, 1 scale
VALU_OPS_PER_KEY = 45.0                 # proposer hot loop, counted in the gfx950 ISA: ~40 VALU instructions per
                                        # 64-key stride (LDS reads aside) + the amortised insertions, per lane = per key
HIP_HW_QUEUES = "16"                    # hardware queues asked of the HIP runtime (GPU_MAX_HW_QUEUES; its default is 4)
LDS_ROUND_TRIP_CYC, CLOCK_GHZ = 50.0, 2.4   # MI355X_MICROARCH.md: ds_read issue->use ~50 cycles


def med(stats, k):
    return statistics.median(s[k] for s in stats)


def cpu_baseline(sw, budget_s: float = 25.0) -> dict:
    """Time the oracle (kind "port": C restatement of the reference path) on this host."""
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
    # best-effort CPU (so the GPU speed-up is not overstated): predicate evaluated once per
    # (configuration, node), distances cached per seed, and the pair sweep on interned masks — first applicable task
    # and count per configuration bit (a group's selector has one bit), then one look-up per worker
    st2 = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    t0 = time.perf_counter()
    st2.try_form_new_groups()
    st2.try_merge_solo_groups()
    t_carve_be = time.perf_counter() - t0
    tm = sw.task_masks()
    t0 = time.perf_counter()
    first = np.full(len(sw.configs), 0xFFFFFFFF, dtype=np.uint32)
    for c in range(len(sw.configs)):
        hit = np.nonzero(tm & (np.uint64(1) << np.uint64(c)))[0]
        if len(hit):
            first[c] = hit[0]
    _task = np.where(cfg_of_node >= 0, first[np.maximum(cfg_of_node, 0)], 0xFFFFFFFF)
    t_sweep_be = time.perf_counter() - t0
    t_be = t_compat + t_carve_be + t_sweep_be
    best_effort = {"value": sw.T * sw.W / t_be, "unit": "pair-evals/s", "cores": 1,
                   "sample": (f"full match, 1 thread: compat {t_compat:.3f} s + carve/merge with the predicate evaluated "
                              f"once per (config, node) and cached distances {t_carve_be:.2f} s + mask-interned sweep "
                              f"{t_sweep_be:.3f} s (numpy)"),
                   "seconds_full_match": t_be}
    return {
        "value": sw.T * sw.W / t_full, "unit": "pair-evals/s", "cores": 1, "kind": "port",
        "sample": (f"oracle/pm_oracle.c -O2, 1 thread of {os.cpu_count()} host cores: compat sweep + reference-shaped "
                   f"carve+merge on all {sw.W} workers ({t_compat + t_carve:.2f} s) + pair sweep on {t_s} of {sw.T} "
                   f"tasks ({dt:.2f} s, scaled linearly to {t_sweep_full:.2f} s); Redis/JSON costs of the real "
                   f"reference excluded"),
        "seconds_full_match_est": t_full,
        "all_cores": all_cores,
        "best_effort": best_effort,
    }


def pmc_traffic(kernel: str, key: str = "hbm_bytes_per_match"):
    """HBM bytes per full-swarm match from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json: FETCH_SIZE
    and WRITE_SIZE collected in separate runs of this same command, newest round first).  A constant read from a
    committed profile, not a measurement of this run; None if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                v = json.load(f).get(kernel, {}).get(key)
            if v is not None:
                return v, os.path.basename(path)
        except Exception:
            pass
    return None, None


def per_task_times(eng, reps=5):
    """pm_match_per_task (best bid + bidder count per task): median of `reps` warm calls, with the D2H copy of both
    columns (what the C ABI call returns) and with the columns left in HBM (pm_match_per_task_device)"""
    eng.match_per_task()
    full, dev = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        eng.match_per_task()
        full.append(1e3 * (time.perf_counter() - t0))
        t0 = time.perf_counter()
        eng.match_per_task_device()
        dev.append(1e3 * (time.perf_counter() - t0))
    return {"per_task_ms": sorted(full)[reps // 2], "per_task_device_ms": sorted(dev)[reps // 2]}


def proposer_split(E, host, sw, seed, steps=3):
    """The proposer's share of the carve: a separate engine whose proposer launches are bracketed by their own
    hipEvents (pm_engine_config.time_proposer).  The events break the back-to-back dispatch of the launch sequence
    (~0.5 ms per match at configs[1]), so the headline loop runs without them."""
    eng = E.Engine(group_id_seed=seed, time_proposer=True)
    host.load_swarm(eng, sw)
    st = []
    for k in range(steps + 1):
        eng.reset_groups()
        s = eng.tick()
        if k:
            st.append(s)
    cc = eng.debug_carve_counters()
    eng.close()
    return {"ms": med(st, "ms_propose_kernel"), "proposals": med(st, "proposals"), "keys": med(st, "propose_keys"),
            "ms_sweep_kernel": med(st, "ms_sweep_kernel"),   # (this engine brackets the sweep's kernel too: the headline's does not)
            "ms_carve_kernel_with_events": med(st, "ms_carve_kernel"),
            "index": {"grid": cc["cell_g"], "indexed_positions": cc["n_indexed"], "batches": cc["batches"],
                      "batches_walked": cc["pruned_batches"], "walks_given_up": cc["prune_fallbacks"]}}


def kernel_table(sw, stats, T, W, prop):
    n_cfgs, n_alts = len(sw.configs), sum(1 for c in sw.configs if c[3]) * 2
    carve_ms = med(stats, "ms_carve_kernel")
    carve_bytes = 20.0 * med(stats, "carve_cand_sum") + 8.0 * med(stats, "carve_steps")
    compat_bytes = W * 32 + n_cfgs * 32 + n_alts * 32 + W * 8
    sweep_bytes = T * 16 + W * 16 + W * 8          # per-worker orientation: (16T + 24W)
    # (the sweep kernel's own events are two barrier packets in a match: only the time_proposer engine of proposer_split records them)
    compat_ms, sweep_ms = med(stats, "ms_compat_kernel"), (prop.get("ms_sweep_kernel") or med(stats, "ms_sweep_kernel"))
    prop_ms, keys = prop["ms"], prop["keys"]
    gbs = lambda b, ms: b / (ms * 1e-3) / 1e9 if ms > 0 else None
    pair_rate = T * W / (sweep_ms * 1e-3) if sweep_ms > 0 else None
    steps = max(med(stats, "carve_steps"), 1)
    return {
        "compat_kernel": {"ms": compat_ms, "alg_bytes": compat_bytes, "GB/s": gbs(compat_bytes, compat_ms)},
        "pair_sweep": {"ms": sweep_ms, "alg_bytes": sweep_bytes, "GB/s": gbs(sweep_bytes, sweep_ms),
                       "pair_evals_per_s": pair_rate,
                       "valu_ceiling_pair_evals_per_s": VALU_LANE_OPS / OPS_PER_PAIR_EVAL,
                       "x_one_pair_per_op_ceiling": pair_rate / (VALU_LANE_OPS / OPS_PER_PAIR_EVAL) if pair_rate else None,
                       "note": ("x_one_pair_per_op_ceiling is a RATIO, not a fraction of a peak: the kernel is bit-sliced (one "
                                "64-bit op evaluates 64 pairs), so it runs tens of times above SURVEY 8d's one-pair-per-lane-op "
                                "ceiling (CUs x lanes x clk / 20 ops); as a share of the bit-sliced ceiling (x 64) it is that / 64")},
        "carve": {"ms": carve_ms, "alg_bytes": carve_bytes, "GB/s": gbs(carve_bytes, carve_ms),
                  "steps": steps, "fast_steps": med(stats, "carve_fast_steps"), "us_per_step": 1e3 * carve_ms / steps,
                  "launches": med(stats, "carve_launches")},
        "carve_propose_kernel": {"ms": prop_ms, "proposals": prop["proposals"], "keys": keys,
                                 "keys_per_proposal": keys / prop["proposals"] if prop["proposals"] else None,
                                 "proposals_per_group": prop["proposals"] / steps,
                                 "spatial_index": prop.get("index"),
                                 "timing": ("separate pass with hipEvents around every proposer launch" if prop_ms > 0 else
                                            "streaming carve: the proposer workgroups run INSIDE carve_stream_kernel, "
                                            "concurrently with the validator's chain — no launch of their own to time; "
                                            "proposals = tickets issued, keys from a separate counting pass"),
                                 "keys_per_s": keys / (prop_ms * 1e-3) if prop_ms > 0 else None,
                                 "valu_ops_per_key": VALU_OPS_PER_KEY,
                                 "valu_frac": (keys * VALU_OPS_PER_KEY / (prop_ms * 1e-3) / VALU_LANE_OPS) if prop_ms > 0 else None,
                                 "fp64_TFLOP/s": keys * FLOP_PER_KEY / (prop_ms * 1e-3) / 1e12 if prop_ms > 0 else None,
                                 "fp64_vector_peak_TFLOP/s": FP64_VECTOR_TFLOPS,
                                 "note": ("keys = live candidates evaluated (dead slots of a thinning list are swept too and "
                                          "not counted); the chord form needs 9 fp64 flop per key, so the kernel is bound "
                                          "by integer / compare / select issue, not by the FP64 rate")},
    }


def chain_model(steps, carve_ms, prop_ms):
    floor_us = 3.0 * LDS_ROUND_TRIP_CYC / (CLOCK_GHZ * 1e3)
    val_ms = max(carve_ms - prop_ms, 0.0)
    return {"model": ("group g+1's seed depends on what group g removed: a chain of dependent steps; floor per step = 3 "
                      "dependent LDS round trips (row -> live bits -> kill) of ~50 cycles at 2.4 GHz.  Streaming carve "
                      "(carve_variant 0): ONE launch, the rows are made by the other workgroups while the chain runs, so "
                      "what is left on the chain besides the steps is waiting for rows at the start of a configuration; "
                      "batch pipeline (variant 3): the preparation and proposer launches between the validation "
                      "launches are on the chain too.  What a step costs beyond the floor (round 6, priced in situ with padded "
                      "builds, s_memtime marks, and alone on a CU: profiles/r06_chain_step_microbench.txt): the chain wave's own plain step is "
                      "168 cycles alone (75 of them the look at the bitmap), 0.117 us per commit inside the kernel, 0.2 - 0.25 under the "
                      "load of the seven waves beside it; two thirds of a configs[1] launch are not chain runs but the first rows and the "
                      "ends of its twelve configurations (profiles/r06_pipeline_experiments.txt: which wave is on the critical path) — "
                      "the chain takes entries as they arrive, the collector frees their slots, the parkers refill them; "
                      "round 6 rewrote the chain's run and the collector"),
            "steps": steps, "floor_us_per_step": floor_us, "achieved_us_per_step": 1e3 * carve_ms / max(steps, 1),
            "validate_only_us_per_step": 1e3 * val_ms / max(steps, 1),
            "frac": floor_us / (1e3 * carve_ms / max(steps, 1)) if carve_ms > 0 else None}


def run_extra_configs2(E, host, seed):
    from protocol_amd.swarm import baseline_config
    sw = baseline_config(2, seed=seed)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    stats = []
    for k in range(4):
        eng.reset_groups()
        s = eng.tick()
        if k:
            stats.append(s)
    prop = proposer_split(E, host, sw, seed, steps=2)
    kt = kernel_table(sw, stats, sw.T, sw.W, prop)
    out = {"workload": "BASELINE configs[2]: 1M tasks x 100k workers, Zipf-skewed topologies, cold match",
           "steps": len(stats), "ms_per_match": med(stats, "ms_total"), "groups": int(stats[-1]["n_groups"]),
           "pair_evals_per_s": sw.T * sw.W / (med(stats, "ms_total") * 1e-3),
           "carve_ms": med(stats, "ms_carve"), "carve_kernel_ms": med(stats, "ms_carve_kernel"),
           "carve_launches": med(stats, "carve_launches"),
           "sweep_ms": med(stats, "ms_sweep"), "compat_ms": med(stats, "ms_compat"), "publish_ms": med(stats, "ms_publish"),
           "host_resolved_steps": int(stats[-1]["host_resolved_steps"]),
           "roofline": {"bound": "hbm", "binds": "latency-chain", "kernel": "carve_stream_kernel (the dominant kernel of this match)",
                        "achieved": kt["carve"]["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": (kt["carve"]["GB/s"] or 0.0) / HBM_PEAK_GBS,
                        "traffic": pmc_traffic("configs2_carve")[0],
                        "note": ("SURVEY 8(d) accounting (W_remaining*20+8 bytes per step) over the carve's launch sequence; "
                                 "what binds it is the dependent chain of steps — see `chain` (floor per step vs achieved)")},
           "kernels": kt,
           "chain": chain_model(kt["carve"]["steps"], kt["carve"]["ms"], kt["carve_propose_kernel"]["ms"])}
    # the north_star orientation at this size (the first call allocates its buffers: not timed)
    out.update(per_task_times(eng))
    eng.close()
    return out


def run_extra_pools(E, host, seed, ks=(2, 4, 8), steps=8):
    """K independent pools on ONE GPU in one process: K engines (each with its own stream and its own configs[1] swarm),
    driven from K host threads (the C ABI releases the GIL), each running the N = 1 line's loop.  The carve's launch keeps
    a workgroup resident on every CU it uses, so every engine is told its share of the CUs (pm_set_carve_workgroups):
    K launches then run side by side instead of queueing behind one another.  The aggregate rate is what a GPU shared by
    several orchestrator pools delivers, the per-match latency what each of them sees."""
    import threading
    from protocol_amd.swarm import baseline_config
    engines = []
    for k in range(max(ks)):
        sw = baseline_config(1, seed=seed + 100 + k)
        eng = E.Engine(group_id_seed=seed + 100 + k)
        host.load_swarm(eng, sw)
        eng.tick()
        engines.append((eng, sw))
    # the one-pool rate of the same loop (engine 0 alone, the whole GPU)
    t1 = []
    for _ in range(steps):
        engines[0][0].reset_groups()
        t0 = time.perf_counter()
        engines[0][0].tick()
        t1.append(time.perf_counter() - t0)
    one = statistics.median(t1)
    one_rate = float(engines[0][1].T) * float(engines[0][1].W) / one
    out = {"workload": "K x BASELINE configs[1] (one swarm per pool, seeds differ), concurrent cold matches on one GPU",
           "steps_per_pool": steps, "one_pool": {"match_ms_p50": 1e3 * one, "pair_evals_per_s": one_rate},
           "supported_way": ("tick_many (pm_tick_many: ONE call for the K pools — include/pm_engine.h); by_k_python_threads is K "
                             "Python threads each calling pm_tick: what that leg reaches is the harness's thread scheduling "
                             "(single matches of 7 - 9 ms among medians of 1.5), kept for the record"),
           "by_k_python_threads": {},
           "hip_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)"),
           "note": ("K engines in ONE process, one host thread and one HIP stream each; every engine takes (CUs - K) / K "
                    "row-making workgroups for its carve (pm_set_carve_workgroups).  x_one_pool = aggregate rate / the rate "
                    "of one pool that has the GPU to itself.  hip_hw_queues = GPU_MAX_HW_QUEUES of this process: with the "
                    "runtime's default of 4 the engines' streams share hardware queues and K = 4 reaches 1.4x "
                    "(profiles/r04_pools_hw_queues.json: the same legs under 4, 8, 16 and 32 queues)")}
    for K in ks:
        share = max(16, (248 - K) // K)
        for i in range(K):
            engines[i][0].set_carve_workgroups(share)
        lat = [[] for _ in range(K)]
        go = threading.Barrier(K + 1)

        def run(i):
            eng = engines[i][0]
            go.wait()
            for _ in range(steps):
                eng.reset_groups()
                t0 = time.perf_counter()
                eng.tick()
                lat[i].append(1e3 * (time.perf_counter() - t0))

        th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
        for t in th:
            t.start()
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        pairs = sum(float(engines[i][1].T) * float(engines[i][1].W) for i in range(K)) * steps
        allm = sorted(x for l in lat for x in l)
        out["by_k_python_threads"][str(K)] = {"carve_workgroups_per_pool": share, "pair_evals_per_s": pairs / el, "x_one_pool": pairs / el / one_rate,
                               "match_ms_p50": allm[len(allm) // 2], "match_ms_max": allm[-1], "wall_ms": 1e3 * el}
    # ---- the same pools through pm_tick_many: ONE call per round of K matches — from one host thread (the carves are
    # started before the first is waited for), and with the library's thread-per-engine variant (K threads inside the
    # library: no Python between the calls)
    try:
        tm = {"note": ("pm_tick_many(engines, K): per round K x pm_reset_groups + ONE call; x_one_pool = aggregate rate / the "
                       "one-pool rate above; match_ms = per pool, the host's clock from the start of its tick to its published "
                       "table (pm_stats.ms_total); `staged` = one host thread walks the engines (start every carve, then "
                       "finish each, then publish each), `threads` = a host thread per engine inside the library"),
              "by_k": {}}
        for K in tuple(ks) + (16,):
            while len(engines) < K:
                k = len(engines)
                sw = baseline_config(1, seed=seed + 100 + k)
                eng = E.Engine(group_id_seed=seed + 100 + k)
                host.load_swarm(eng, sw)
                eng.tick()
                engines.append((eng, sw))
            share = max(8, (248 - K) // K)
            batch = [engines[i][0] for i in range(K)]
            for eng in batch:
                eng.set_carve_workgroups(share)
            pairs = sum(float(engines[i][1].T) * float(engines[i][1].W) for i in range(K)) * steps
            row = {"carve_workgroups_per_pool": share}
            for mode, threads in (("staged", False), ("threads", True)):
                for eng in batch:       # (one untimed round: the engines' first launch with this share)
                    eng.reset_groups()
                E.tick_many(batch, threads=threads)
                lat, t0 = [], time.perf_counter()
                for _ in range(steps):
                    for eng in batch:
                        eng.reset_groups()
                    lat += [s["ms_total"] for s in E.tick_many(batch, threads=threads)]
                el = time.perf_counter() - t0
                lat.sort()
                row[mode] = {"pair_evals_per_s": pairs / el, "x_one_pool": pairs / el / one_rate,
                             "round_ms": 1e3 * el / steps, "match_ms_p50": lat[len(lat) // 2], "match_ms_max": lat[-1]}
            tm["by_k"][str(K)] = row
        out["tick_many"] = tm
    except Exception as ex:
        out["tick_many"] = {"error": repr(ex)}
    for eng, _ in engines:
        eng.close()
    return out


def run_extra_merge(E, host, seed, reps=3):
    """try_merge_solo_groups (mod.rs:631-971) on its own workload: 100k workers, ~5,000 standing solo groups, a (2, 8)
    configuration gets enabled and pm_merge_solo_groups merges them.  The digest covers the life-cycle feed of the merge
    (per merged group, in creation order: the dissolved solos in batch order, then the new group) —
    tests/test_gpu_scale.py::test_solo_merge_of_the_bench_workload checks the same digest against the oracle."""
    from protocol_amd.swarm import events_digest, solo_merge_swarm
    sw = solo_merge_swarm(seed)
    eng = E.Engine(group_id_seed=seed)
    host.load_swarm(eng, sw, enabled=0b01)
    eng.enable_group_events()
    ms, digest, n_solo, n_merged, launches = [], None, 0, 0, 0
    for k in range(reps + 1):
        eng.reset_groups()
        eng.set_enabled_mask(0b01)
        n_solo = eng.form_groups()
        eng.drain_group_events()
        eng.set_enabled_mask(0b11)
        t0 = time.perf_counter()
        n_merged = eng.merge_solo_groups()
        dt = 1e3 * (time.perf_counter() - t0)
        ev = eng.drain_group_events()
        if k:
            ms.append(dt)
        if digest is None:   # (group ids continue the engine's id stream: the first pass is the seeded one)
            digest = events_digest(ev)
    eng.close()
    return {"workload": "100k workers, ~5,000 solo groups of a (1, 1) configuration; a (2, 8) configuration is enabled and "
                        "pm_merge_solo_groups merges them",
            "workers": int(sw.W), "solos": int(n_solo), "merged_groups": int(n_merged), "ms_p50": statistics.median(ms),
            "ms": ms, "events_digest": digest,
            "note": "host bookkeeping is linear in the solos (dead-marking + one compaction, one host wait per kernel launch)"}


def run_extra_churn(E, host, seed, ticks=6):
    """BASELINE configs[4] on one GPU (the 8-GPU form is `--gpus 8`): 100k workers; per tick 10k tasks arrive, 1 % of
    the workers die and 1 % brand-new workers join; one incremental match on the standing groups.  The stream is
    protocol_amd/churn.py — the one tests/golden/churn_digests.json pins against the oracle."""
    from protocol_amd.churn import ChurnStream
    cs = ChurnStream(seed, ticks + 2)
    sw_all = cs.sw_all
    packed = host.pack_workers(sw_all)
    rows = lambda idx: {k: np.ascontiguousarray(v[idx]) for k, v in packed.items()}
    eng = E.Engine()
    cfg_rows, alt_rows, req_models = host.pack_configs(sw_all.configs)
    eng.set_configs(cfg_rows, alt_rows)
    eng.set_model_table(host.build_model_table(req_models, sw_all.model_names), len(req_models), len(sw_all.model_names))
    eng.upload_workers(rows(np.arange(cs.W0)))
    eng.upload_tasks(cs.masks, cs.created, cs.uid)
    eng.set_enabled_mask(sw_all.enabled_mask())
    s0 = eng.tick()
    flags = packed["flags"].astype(np.int64)
    out_ticks = []
    for t in range(ticks + 2):
        leave, idx_new, new_tasks = cs.step()  # (harness, not timed)
        t0 = time.perf_counter()
        eng.on_worker_status_many(leave, flags[leave] & ~E.W_HEALTHY, np.ones(len(leave), dtype=np.uint32))
        t1 = time.perf_counter()
        new_rows = rows(idx_new)
        t1b = time.perf_counter()
        eng.append_workers(new_rows)
        t2 = time.perf_counter()
        eng.tasks_insert_front(*new_tasks[:3])
        t3 = time.perf_counter()
        s = eng.tick()
        t4 = time.perf_counter()
        if t >= 2:
            out_ticks.append({"status_ms": 1e3 * (t1 - t0), "append_ms": 1e3 * (t2 - t1b), "tasks_ms": 1e3 * (t3 - t2),
                              "match_ms": 1e3 * (t4 - t3), "carve_ms": s["ms_carve"], "sweep_ms": s["ms_sweep"],
                              "publish_ms": s["ms_publish"], "formed": s["n_formed"], "groups": s["n_groups"]})
    eng.close()
    m = lambda k: statistics.median(x[k] for x in out_ticks)
    return {"workload": ("BASELINE configs[4] on one GPU: 100k workers, per tick +10k tasks (pm_tasks_insert_front), 1% "
                         "workers die (one pm_on_worker_status_many call), 1% brand-new workers (pm_append_workers), incremental "
                         "pm_tick on the standing groups"),
            "pinned_by": "tests/golden/churn_digests.json (all eight ticks of this stream — the two warm-up ticks and the six timed ones — against the oracle)",
            "ticks": len(out_ticks), "cold_match_ms": s0["ms_total"],
            "ms_per_tick": m("status_ms") + m("append_ms") + m("tasks_ms") + m("match_ms"),
            "split_ms_p50": {k: m(k) for k in ("status_ms", "append_ms", "tasks_ms", "match_ms", "carve_ms", "sweep_ms",
                                               "publish_ms")},
            "formed_per_tick_p50": m("formed"), "groups": out_ticks[-1]["groups"],
            "carve_hbm_bytes_per_tick": pmc_traffic("churn_carve", "hbm_bytes_per_tick")[0],
            "note": ("ms_per_tick is the SUM of the four phases' medians (status + append + tasks + match), each timed on its "
                     "own with the host clock; tools/churn_probe.py (profiles/*_churn_ticks.txt) times the tick call alone "
                     "and prints single ticks, so the two differ by the delta calls (0.4 - 0.5 ms) and by what a median hides")}


def run_one_pool_sharded(E, host, torch, dist, args, rank, world, local_rank, dev, coll_dev, backend, steps=5, warmup=2):
    """BASELINE configs[3]: ONE 1M x 100k pool matched by all ranks (strong scaling).  The carve is replicated (every rank
    runs the whole streaming launch), the pair sweep + claim run over the owned workers, one all-gather of the published
    rows per tick.  Returns the sub-object of the line (every rank takes part; rank 0's numbers are reported)."""
    import hashlib
    from protocol_amd.dist import EngineLocal, ShardedEngine
    from protocol_amd.swarm import baseline_config
    sw = baseline_config(3, seed=args.seed)                  # the SAME swarm on every rank
    eng = E.Engine(device=local_rank, group_id_seed=args.seed)
    host.load_swarm(eng, sw)
    sharded = ShardedEngine(EngineLocal(eng, dev), sw.address)
    sharded.time_exchanges = True

    def step():
        eng.reset_groups()
        return sharded.tick()

    for _ in range(warmup):
        step()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = [step() for _ in range(steps)]
    torch.cuda.synchronize()
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=coll_dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # every rank must hold the same groups and the same full table: compare a digest across the ranks
    _, groups, members = eng.get_groups()
    h = hashlib.sha256(groups.tobytes() + members.tobytes()).digest()[:8]
    mine = torch.tensor([int.from_bytes(h, "little") >> 1, len(groups)], dtype=torch.int64, device=coll_dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    assert all(bool((v == mine).all()) for v in allv), "ranks disagree on the groups they formed"
    n_ticks = steps + warmup
    exch_ms = sharded.exchange_ms / n_ticks
    sharded_ms = med(stats, "ms_sweep")
    T, W = sw.T, sw.W
    out = {"workload": ("BASELINE configs[3]: 1M tasks x 100k workers, Zipf-skewed topologies, ONE pool, worker ownership "
                        "hash-sharded across the ranks, one all-gather of the published rows per tick"),
           "scaling": "strong", "steps": steps, "ms_per_step": 1e3 * elapsed / steps,
           "value": float(T) * float(W) * steps / elapsed, "unit": "pair-evals/s",
           "sharding": "ownership = splitmix64(address) % n_gpus; carve replicated, pair sweep + claim over the owned workers",
           "workers_owned_by_rank0": int((sharded.shard == rank).sum()), "exchanges_per_tick": sharded.exchanges / n_ticks,
           "backend": backend, "identical_groups_on_all_ranks": True, "groups": int(stats[-1]["n_groups"]),
           "exchange_ms": exch_ms, "sharded_ms": sharded_ms,
           "replicated_ms": max(1e3 * elapsed / steps - sharded_ms - exch_ms, 0.0),
           "split_note": ("per tick on rank 0: exchange = device time inside the ONE all-gather of a tick (the published rows of "
                          "the owned workers; events around it); sharded = this rank's pair sweep + claim over the workers it "
                          "owns; replicated = the rest of the tick — compat sweep, the carve (one streaming launch, the same on "
                          "every rank: a chain of dependent steps that N GPUs do not divide), publish")}
    if rank == 0:
        # the SAME swarm on one GPU, unsharded (a second engine on rank 0's device): the honest one-GPU reference of this
        # sub-object's strong scaling
        solo = E.Engine(device=local_rank, group_id_seed=args.seed)
        host.load_swarm(solo, sw)
        ts = []
        for _k in range(4):
            solo.reset_groups()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            solo.tick()
            ts.append(time.perf_counter() - t1)
        solo.close()
        one = statistics.median(ts[1:])
        out["one_gpu_same_workload"] = {"ms_per_step": 1e3 * one, "value": float(T) * float(W) / one,
                                        "note": "rank 0, unsharded engine, p50 of 3 cold matches"}
        out["speedup_vs_one_gpu"] = one / (elapsed / steps)
    dist.barrier()   # (rank 0 may still be measuring its one-GPU reference)
    eng.close()
    return out


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=21)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=None,
                    help="BASELINE.json configs index (default: 1 on one GPU, 3 on several)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sweep-variant", type=int, default=0)
    ap.add_argument("--carve-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the configs2 / churn / per_task sub-objects")
    ap.add_argument("--check", action="store_true", help="also verify the groups against the oracle (slow)")
    args = ap.parse_args()

    # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 unless the environment
    # says otherwise) and runs two streams that share one in turn: with K engines (one stream each; two until round 5)
    # in one process, K = 4 pools matched one behind the other in pairs (1.4x the one-pool rate; 3.4x with 8 or more
    # queues — `pools_on_one_gpu`).  Read once, when the runtime starts: set before anything touches HIP.  One engine
    # is indifferent to it (1.31 vs 1.32 ms per match).  Only the N = 1 line runs several engines in one process; the
    # N > 1 ranks (one engine each, beside RCCL's own streams) keep the runtime's default, as they were measured.
    if int(os.environ.get("WORLD_SIZE", "1")) == 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", HIP_HW_QUEUES)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X (the engine has no CPU fallback)", file=sys.stderr)
        return 2
    # PM_BENCH_BACKEND=gloo + PM_BENCH_SHARE_DEVICE=1: plumbing check of the N>1 path on a 1-GPU box (all ranks
    # on device 0, collectives staged through the host).  The real path is nccl (= RCCL over xGMI), one GPU per rank.
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

    # N = 1: configs[1].  N > 1: REPLICAS — the same configuration, one independent pool per rank (seed + rank), no
    # data-path collective (weak scaling); the one-pool sharded tick of configs[3] is measured behind it (dist.one_pool_sharded)
    cfg_index = args.config if args.config is not None else 1
    seed_r = args.seed + (rank if world > 1 else 0)
    sw = baseline_config(cfg_index, seed=seed_r)
    names = {1: "100k tasks x 10000 workers, mixed gpu/mem/storage/cpu constraints (BASELINE configs[1])",
             2: "1M tasks x 100k workers, Zipf-skewed topologies (BASELINE configs[2])"}
    workload = names.get(cfg_index, f"BASELINE configs[{cfg_index}]")
    if world > 1:
        workload += f" — one such pool per rank (seed + rank), {world} independent engines"

    eng = E.Engine(device=local_rank, sweep_variant=args.sweep_variant, carve_variant=args.carve_variant,
                   group_id_seed=seed_r)
    host.load_swarm(eng, sw)

    def step():
        eng.reset_groups()
        return eng.tick()

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(step())
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([elapsed, 1.0], dtype=torch.float64, device=coll_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        ranks_seen = int(round(float(t[1].item())))

    T, W = sw.T, sw.W
    value = world * float(T) * float(W) * args.steps / elapsed   # whole job: every rank matched its own pool K times
    ms = [s["ms_total"] for s in stats]
    prop = proposer_split(E, host, sw, args.seed) if (world == 1 and not args.no_extras) else \
        {"ms": 0.0, "proposals": med(stats, "proposals"), "keys": med(stats, "propose_keys"),
         "ms_carve_kernel_with_events": 0.0}
    kt = kernel_table(sw, stats, T, W, prop)
    carve = kt["carve"]
    traffic, traffic_src = pmc_traffic("carve")
    out = {
        "metric": "task x worker pair-evals/sec (full-swarm match)", "value": value, "unit": "pair-evals/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": workload, "tasks": T, "workers": W, "workers_total": W * world, "configs": len(sw.configs),
                   "seed": args.seed,
                   "parallelism": ("one pool on one GPU" if world == 1 else
                                   f"{world} replicas: one independent pool per GPU, no data-path collective (the one-pool sharded "
                                   f"tick of configs[3] is in dist.one_pool_sharded)"),
                   "sweep_variant": args.sweep_variant, "carve_variant": args.carve_variant},
        "p50_match_latency_ms": statistics.median(ms),
        "match_latency_ms": {"min": min(ms), "p50": statistics.median(ms), "max": max(ms)},
        "match_latency_note": ("pm_stats.ms_total of the timed ticks: the host's clock from the tick's begin to its end (the claim "
                               "kernel writes the host's snapshot buffer itself, the tick's last event is queued behind it)"),
        "phase_ms_p50": {k: med(stats, k) for k in ("ms_compat", "ms_carve", "ms_merge", "ms_sweep", "ms_publish")},
        "phase_note": ("GPU time between hipEvents on the engine's stream; ms_sweep = match_prep + pair sweep + the claim that publishes "
                       "(rows straight into pinned host memory): ms_publish is 0 on this path"),
        "groups": int(stats[-1]["n_groups"]), "host_resolved_steps": int(stats[-1]["host_resolved_steps"]),
        "roofline": {"bound": "hbm", "binds": "latency-chain",
                     "kernel": ("carve_stream_kernel (+ its list preparation and carve_finish_kernel: one launch sequence, "
                                "one hipEvent pair)" if args.carve_variant == 0 else
                                "carve (preparation + carve_propose_kernel + carve_kernel launch sequence)"),
                     "achieved": carve["GB/s"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (carve["GB/s"] or 0.0) / HBM_PEAK_GBS,
                     "bound_note": ("bound / achieved / peak / frac are the contract's HBM accounting (algorithmic bytes over "
                                    "the launch sequence's duration: of the contract's two roofs, hbm is the one this byte / integer "
                                    "path belongs under) and are NOT what binds the sequence — `binds`: it is a chain of "
                                    "dependent steps, priced in `chain` (floor per step vs achieved)"),
                     "traffic": traffic, "traffic_source": (f"profiles/{traffic_src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                                            f"of this command, a committed constant — not measured in this run")
                     if traffic_src else None,
                     "note": ("SURVEY 8(d) accounting (W_remaining*20+8 bytes per step); the sequence is a dependent chain, "
                              "not a bandwidth-shaped scan: see `chain`"),
                     "chain": chain_model(carve["steps"], carve["ms"], kt["carve_propose_kernel"]["ms"])},
        "kernels": kt,
    }
    if world > 1:
        out["ranks_seen"] = ranks_seen
        out["cpu_baseline"] = None
        # The secondary measurement is a sequence of collectives: an exception on ONE rank leaves the others inside a
        # collective it never joins, and RCCL has not met this path on N > 1 GPUs before the driver's run.  A watchdog
        # thread (the main thread may be inside a collective, which releases the GIL) prints the headline as it stands —
        # it is complete at this point — and ends the process when the leg has not returned in time: never lose the line.
        import threading
        done = threading.Event()
        once = threading.Lock()   # whoever takes it first — the watchdog or the main thread — is the one that prints the line
        limit_s = float(os.environ.get("PM_BENCH_DIST_TIMEOUT", "120"))

        def watchdog():
            if done.wait(limit_s):
                return
            if not once.acquire(blocking=False):
                return   # (the main thread is through and printing)
            if rank == 0:
                line = dict(out)
                line["dist"] = {"one_pool_sharded": {"error": f"no result within {limit_s:.0f} s (a rank failed or a collective hung): "
                                                              f"the headline above is unaffected"}}
                print(json.dumps(line), flush=True)
            os._exit(0)

        wd = threading.Thread(target=watchdog, daemon=True)
        wd.start()
        try:
            out["dist"] = {"one_pool_sharded": run_one_pool_sharded(E, host, torch, dist, args, rank, world, local_rank, dev, coll_dev, backend)}
        except Exception as ex:   # never lose the line over the secondary measurement
            out["dist"] = {"one_pool_sharded": {"error": repr(ex)}}
            if rank == 0:   # (the other ranks may be inside a collective this one left: do not wait for them)
                done.set()
                if once.acquire(blocking=False):
                    print(json.dumps(out), flush=True)
                os._exit(0)
            # (another rank: stay — a peer that disappears can take rank 0's communicator down with it before the line is out;
            # the watchdog ends this process when rank 0's has printed)
            time.sleep(limit_s + 30.0)
            os._exit(0)
        done.set()
        if not once.acquire(blocking=False):   # (the watchdog has the line: it prints and ends the process)
            time.sleep(60.0)
            os._exit(0)
    single = rank == 0 and world == 1
    if single and not args.no_extras:
        try:
            out["hbm_triad_gbs_measured"] = eng.hbm_triad_gbs()
        except Exception as ex:  # never lose the line over the microbench
            out["hbm_triad_gbs_measured"] = None
            out["hbm_triad_error"] = repr(ex)
        pt = per_task_times(eng)
        out["per_task"] = {"ms": pt["per_task_ms"], "device_ms": pt["per_task_device_ms"],
                           "pair_evals_per_s": T * W / (pt["per_task_ms"] * 1e-3),
                           "note": "north_star orientation (pm_match_per_task): per task best bid + bidder count; ms "
                                   "includes the D2H copy of both columns, device_ms leaves them in HBM "
                                   "(pm_match_per_task_device); medians of 5 warm calls"}
    if single and not args.no_cpu_baseline:
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
        out["pcie_inclusive"] = {"ms_per_match": up * 1e3, "value": float(T) * float(W) / up, "unit": "pair-evals/s",
                                 "note": "host SoA columns -> HBM (workers + tasks) + match, p50 of 5"}
        out["cpu_baseline"] = cpu_baseline(sw)
    else:
        out["cpu_baseline"] = None
    if args.check and rank == 0:
        from oracle import oracle_ffi as orc
        nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
        st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False, group_id_seed=seed_r)
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        _, groups, members = eng.get_groups()
        got = [(int(g["id"]), int(g["config"]),
                members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])].tolist()) for g in groups]
        out["parity_vs_oracle"] = got == [(gid, c, m) for (_s, gid, c, m, _t) in st.groups()]
    eng.close()
    if single and not args.no_extras:
        for key, fn in (("configs2", run_extra_configs2), ("churn", run_extra_churn), ("merge", run_extra_merge),
                        ("pools_on_one_gpu", run_extra_pools)):
            try:
                out[key] = fn(E, host, args.seed)
            except Exception as ex:
                out[key] = {"error": repr(ex)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        # (the line is out: a rank that left the secondary leg early must not keep the others — and the driver's clock — waiting
        # in this barrier for the collective's own time-out)
        import threading
        threading.Thread(target=lambda: (time.sleep(30.0), os._exit(0)), daemon=True).start()
        dist.barrier()  # (rank 0 may still be measuring its one-GPU reference)
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
