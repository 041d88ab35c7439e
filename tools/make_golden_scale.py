#!/usr/bin/env python3
"""Golden digests of the CPU oracle at the full BASELINE sizes (tests/golden/scale_digests.json).

The oracle (oracle/pm_oracle.c, best-effort mode: masks evaluated once, distances cached — same results as
the reference-shaped mode, tests/test_oracle_groups.py) is run on the seeded swarms

    cfg1_seed1   baseline_config(1, seed=1)    100k tasks x  10k workers
    cfg2_seed1   baseline_config(2, seed=1)      1M tasks x 100k workers, Zipf-skewed topologies
    cfg2_seed2   the same with seed 2
    cfg1_seed1_seeded   cfg1_seed1 with the SEEDED chooser (chooser_seed 77): groups + per-worker task column only
    cfg2_seed1_seeded   cfg2_seed1 likewise

through try_form_new_groups + try_merge_solo_groups (mod.rs:478-628, 631-971) and one get_task_for_node per
worker (scheduler_impl.rs:11-110, chooser FIRST).  What is kept:

    groups_sha256   sha256 over  ids(u64) | configs(u32) | sizes(u32) | members(u32, BTreeSet order), creation order
    task_sha256     sha256 over the per-worker task column (u32, 0xFFFFFFFF = none)
    count_sha256    sha256 over the per-worker applicable-task counts (u32)
    table_sha256    sha256 over GROUP_INDEX | GROUP_SIZE | NEXT worker per worker (u32 each, 0 / 0 / NONE outside groups)
    first_groups / last_groups   the first and last 1000 groups in full, to localise a mismatch

The 1e11-pair sweep of cfg2 runs on every host core (orc_pair_sweep_per_worker_mt); 200 workers are
cross-checked against the oracle's own filter_tasks so the derived columns are tied to that function.

    python tools/make_golden_scale.py            # ~1 min for cfg1, ~5-10 min for cfg2 on 8 cores
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ffi as orc  # noqa: E402
from protocol_amd.swarm import baseline_config  # noqa: E402

NONE = 0xFFFFFFFF
OUT = os.path.join(ROOT, "tests", "golden", "scale_digests.json")


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def groups_digest(groups):
    """groups = [(id, config, members)] in creation order"""
    ids = np.array([g[0] for g in groups], dtype=np.uint64)
    cfg = np.array([g[1] for g in groups], dtype=np.uint32)
    n = np.array([len(g[2]) for g in groups], dtype=np.uint32)
    mem = np.array([m for g in groups for m in g[2]], dtype=np.uint32)
    return sha(ids, cfg, n, mem)


def table_columns(W: int, groups):
    """GROUP_INDEX, GROUP_SIZE, NEXT worker per worker from the member lists (BTreeSet order, mod.rs:424-434;
    next = (idx + 1) % size, scheduler_impl.rs:115-128)"""
    gi = np.zeros(W, dtype=np.uint32)
    gs = np.zeros(W, dtype=np.uint32)
    nx = np.full(W, NONE, dtype=np.uint32)
    for (_id, _c, mem) in groups:
        n = len(mem)
        for k, w in enumerate(mem):
            gi[w], gs[w], nx[w] = k, n, mem[(k + 1) % n]
    return gi, gs, nx


def digest_of(cfg_index: int, seed: int, threads: int) -> dict:
    t0 = time.time()
    sw = baseline_config(cfg_index, seed=seed)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False, group_id_seed=seed)
    n_formed = st.try_form_new_groups()
    n_merged = st.try_merge_solo_groups()
    groups = [(gid, c, mem) for (_s, gid, c, mem, _t) in st.groups()]
    print(f"  cfg{cfg_index} seed {seed}: {n_formed} formed, {n_merged} merged, {time.time() - t0:.1f} s", flush=True)
    cfg_of_node = np.full(sw.W, -1, dtype=np.int32)
    for (_id, c, mem) in groups:
        cfg_of_node[mem] = c
    t1 = time.time()
    first, count = orc.pair_sweep_per_worker(tasks, cfgs, cfg_of_node, threads=threads)
    print(f"  pair sweep {sw.T} x {sw.W} on {threads} threads: {time.time() - t1:.1f} s", flush=True)
    gi, gs, nx = table_columns(sw.W, groups)
    # tie the derived columns to the oracle's own filter_tasks (it also performs the claim)
    rng = np.random.default_rng(7)
    for w in rng.choice(sw.W, size=200, replace=False):
        t, i, s, n = st.filter_tasks(int(w))
        if cfg_of_node[w] < 0:
            assert t < 0, (w, t)
            continue
        assert (NONE if t < 0 else t) == first[w] and (i, s) == (gi[w], gs[w]) and n == nx[w], (w, t, i, s, n)
    return {
        "config": cfg_index, "seed": seed, "W": sw.W, "T": sw.T, "n_groups": len(groups), "n_formed": n_formed,
        "n_merged": n_merged, "groups_sha256": groups_digest(groups), "task_sha256": sha(first),
        "count_sha256": sha(count), "table_sha256": sha(gi, gs, nx),
        "first_groups": [[int(a), int(b), [int(x) for x in m]] for (a, b, m) in groups[:1000]],
        "last_groups": [[int(a), int(b), [int(x) for x in m]] for (a, b, m) in groups[-1000:]],
    }


def digest_seeded(cfg_index: int, seed: int, chooser_seed: int) -> dict:
    """the same swarm with the SEEDED chooser (the injected stand-in for rand::rng().choose, scheduler_impl.rs:66-70):
    groups and the per-worker task column, every worker through the oracle's own get_task_for_node"""
    t0 = time.time()
    sw = baseline_config(cfg_index, seed=seed)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False, group_id_seed=seed,
                   chooser=orc.CHOOSE_SEEDED, chooser_seed=chooser_seed)
    n_formed = st.try_form_new_groups()
    n_merged = st.try_merge_solo_groups()
    task = np.array([st.get_task_for_node(w) for w in range(sw.W)], dtype=np.int64)
    task = np.where(task < 0, NONE, task).astype(np.uint32)
    groups = [(gid, c, mem) for (_s, gid, c, mem, _t) in st.groups()]
    print(f"  cfg{cfg_index} seed {seed} seeded chooser {chooser_seed}: {len(groups)} groups, {time.time() - t0:.1f} s", flush=True)
    return {"config": cfg_index, "seed": seed, "chooser_seed": chooser_seed, "W": sw.W, "T": sw.T, "n_groups": len(groups),
            "n_formed": n_formed, "n_merged": n_merged, "groups_sha256": groups_digest(groups), "task_sha256": sha(task),
            "n_with_task": int((task != NONE).sum())}


def main():
    threads = os.cpu_count() or 1
    out = {}
    if os.path.exists(OUT):
        out = json.load(open(OUT))
    for name, ci in {"cfg1_seed1_seeded": 1, "cfg2_seed1_seeded": 2}.items():
        if len(sys.argv) < 2 or name in sys.argv[1:]:
            print(name, flush=True)
            out[name] = digest_seeded(ci, 1, 77)
            with open(OUT, "w") as f:
                json.dump(out, f, separators=(",", ":"))
                f.write("\n")
    for name, (ci, seed) in {"cfg1_seed1": (1, 1), "cfg2_seed1": (2, 1), "cfg2_seed2": (2, 2)}.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(name, flush=True)
        out[name] = digest_of(ci, seed, threads)
        with open(OUT, "w") as f:
            json.dump(out, f, separators=(",", ":"))
            f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
