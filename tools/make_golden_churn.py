#!/usr/bin/env python3
"""Golden digests of the CPU oracle for the workloads the bench line times but no live oracle run can follow on the
GPU box (tests/golden/churn_digests.json):

  churn      BASELINE configs[4] exactly as bench.py's `churn` sub-object drives it (protocol_amd/churn.py,
             seed 1, 8 ticks planned): the cold match on 100k workers x 10k tasks, then all eight ticks (the two warm-up ticks and the six bench.py times) —
             1000 deaths (groups dissolve, status_update_impl.rs:17-29), 1000 appended workers (mod.rs:487-497),
             10k tasks in front of the list, try_form_new_groups + try_merge_solo_groups, one get_task_for_node
             per worker.  Per tick: sha256 of the groups (sorted by id: ids | configs | sizes | members in BTreeSet
             order), of the per-worker task column, and of the group life-cycle feed in emission order.
  per_task   pm_match_per_task (north_star orientation) at BASELINE configs[1] and [2], cold state: for every task
             the first eligible compatible worker and the number of candidates, from the oracle's own compat masks
             (eligible = Healthy & p2p, mod.rs:492-497; compatible = some enabled configuration of the task's
             topologies the worker meets).

    python tools/make_golden_churn.py [churn] [per_task]      # ~3 min + ~1 min on 8 cores
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
from protocol_amd.churn import ChurnStream  # noqa: E402
from protocol_amd.swarm import baseline_config  # noqa: E402

NONE = 0xFFFFFFFF
OUT = os.path.join(ROOT, "tests", "golden", "churn_digests.json")
CHURN_SEED, CHURN_TICKS_PLANNED, CHURN_TICKS_PINNED = 1, 8, 8


def sha(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def groups_digest(groups) -> str:
    """groups = [(id, config, members)], any order: hashed sorted by id"""
    groups = sorted(groups)
    ids = np.array([g[0] for g in groups], dtype=np.uint64)
    cfg = np.array([g[1] for g in groups], dtype=np.uint32)
    n = np.array([len(g[2]) for g in groups], dtype=np.uint32)
    mem = np.array([m for g in groups for m in g[2]], dtype=np.uint32)
    return sha(ids, cfg, n, mem)


def events_digest(events) -> str:
    """[(kind, id, config, members)] in emission order"""
    kind = np.array([e[0] for e in events], dtype=np.uint32)
    ids = np.array([e[1] for e in events], dtype=np.uint64)
    cfg = np.array([e[2] for e in events], dtype=np.uint32)
    n = np.array([len(e[3]) for e in events], dtype=np.uint32)
    mem = np.array([m for e in events for m in e[3]], dtype=np.uint32)
    return sha(kind, ids, cfg, n, mem)


def churn() -> dict:
    cs = ChurnStream(CHURN_SEED, CHURN_TICKS_PLANNED)
    sw = cs.sw_all
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    status_all = nodes["status"].copy()
    nodes["status"][cs.W0:] = 0  # the oracle's table is fixed-size: not there yet
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False, group_id_seed=1)
    tasks0 = tasks.copy()  # (the stream picks the topologies of its new tasks from the ORIGINAL table)
    out = {"seed": CHURN_SEED, "ticks_planned": CHURN_TICKS_PLANNED, "W0": cs.W0, "T0": len(tasks), "ticks": []}

    def snapshot(W, tag, t0):
        n_formed = st.try_form_new_groups()
        n_merged = st.try_merge_solo_groups()
        task = np.array([st.get_task_for_node(w) for w in range(W)], dtype=np.int64)
        task = np.where(task < 0, NONE, task).astype(np.uint32)
        groups = [(gid, c, mem) for (_s, gid, c, mem, _t) in st.groups()]
        ev = st.drain_events()
        print(f"  {tag}: {n_formed} formed, {n_merged} merged, {len(groups)} groups, {len(ev)} events, "
              f"{time.time() - t0:.1f} s", flush=True)
        return {"W": W, "T": len(st.tasks) if st.tasks is not None else 0, "n_formed": n_formed, "n_merged": n_merged,
                "n_groups": len(groups), "groups_sha256": groups_digest(groups), "task_sha256": sha(task),
                "n_events": len(ev), "events_sha256": events_digest(ev)}

    t0 = time.time()
    out["cold"] = snapshot(cs.W0, "cold", t0)
    for k in range(CHURN_TICKS_PINNED):
        t0 = time.time()
        leave, idx_new, (m_new, c_new, u_new, pick) = cs.step()
        for w in leave:
            st.set_node_status(int(w), orc.ST_DEAD)
        for w in idx_new:
            st.set_node_status(int(w), int(status_all[w]))
        new_rows = tasks0[pick].copy()
        new_rows["created_at"] = c_new
        old_n = len(tasks)
        tasks = np.concatenate([new_rows, tasks])
        st.set_tasks(tasks)
        st.remap_tasks(np.arange(old_n) + len(new_rows))
        out["ticks"].append(snapshot(cs.W, f"tick {k}", t0))
    return out


def per_task(cfg_index: int, seed: int) -> dict:
    t0 = time.time()
    sw = baseline_config(cfg_index, seed=seed)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    masks = orc.compat_masks(nodes, cfgs)
    elig = (sw.status == 2) & sw.has_p2p
    col = np.where(elig, masks & np.uint64(sw.enabled_mask()), np.uint64(0))
    tm = sw.task_masks()
    # one pass per DISTINCT topology mask (a few hundred), not per task
    uniq, inv = np.unique(tm, return_inverse=True)
    best_u = np.full(len(uniq), NONE, dtype=np.uint32)
    count_u = np.zeros(len(uniq), dtype=np.uint32)
    for k, m in enumerate(uniq):
        hit = np.nonzero(col & m)[0]
        count_u[k] = len(hit)
        if len(hit):
            best_u[k] = hit[0]
    best, count = best_u[inv], count_u[inv]
    print(f"  per_task cfg{cfg_index} seed {seed}: {sw.T} tasks, {len(uniq)} distinct masks, {time.time() - t0:.1f} s", flush=True)
    return {"config": cfg_index, "seed": seed, "W": sw.W, "T": sw.T, "best_sha256": sha(best), "count_sha256": sha(count),
            "n_without_candidate": int((count == 0).sum())}


def main():
    out = json.load(open(OUT)) if os.path.exists(OUT) else {}
    what = sys.argv[1:] or ["churn", "per_task"]
    if "churn" in what:
        print("churn", flush=True)
        out["churn"] = churn()
    if "per_task" in what:
        print("per_task", flush=True)
        out["per_task"] = {"cfg1_seed1": per_task(1, 1), "cfg2_seed1": per_task(2, 1)}
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print("wrote", OUT)


if __name__ == "__main__":
    main()
