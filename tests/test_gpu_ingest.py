"""Incremental worker ingestion through the C ABI: pm_append_workers (truly new rows), pm_update_workers (row
deltas of known nodes), pm_set_addr_ranks — against the oracle, tick by tick.

The reference is incremental: a node that appears next tick pairs with the leftovers, existing groups stay as
they are (node_groups/mod.rs:487-497; tests.rs:993-1213); discovery sync rewrites specs / location of known
nodes without touching their groups (orchestrator/src/discovery/monitor.rs:236-420)."""
import copy

import numpy as np
import pytest

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import make_swarm
from helpers import engine_groups, oracle_groups

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF

_WORKER_FIELDS = ("address", "status", "has_p2p", "has_specs", "has_gpu", "gpu_count_some", "gpu_mem_some",
                  "gpu_model_some", "has_cpu", "cpu_cores_some", "ram_some", "storage_some", "gpu_count", "gpu_mem_mb",
                  "gpu_model_id", "cpu_cores", "ram_mb", "storage_gb", "price", "has_loc", "lat", "lon")


def _take(sw, idx):
    """the swarm restricted to worker rows idx (tasks and configurations unchanged)"""
    s = copy.copy(sw)
    for k in _WORKER_FIELDS:
        setattr(s, k, getattr(sw, k)[idx].copy())
    return s


def _rows(packed, idx):
    return {k: np.ascontiguousarray(v[idx]) for k, v in packed.items()}


def _tasks_of(eng, W):
    return [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(W)]


def test_a_fourth_node_joins_next_tick():
    """tests.rs:993-1213: three nodes, groups of two -> one group and a leftover; a FOURTH node that did not exist
    before pairs with the leftover on the next tick, the first group is untouched."""
    sw = make_swarm(41, 20, 4)
    sw.configs = [("pair", 2, 2, None)]
    sw.topo[:] = -2
    sw.topo[:, 0] = 0                                                 # every task allows the one configuration
    sw.restricted[:] = True
    sw.n_topo[:] = 1
    sw.status[:] = 2
    sw.has_p2p[:] = True
    first3 = _take(sw, np.arange(3))
    eng = E.Engine()
    host.load_swarm(eng, first3)
    eng.tick()
    g_before = engine_groups(eng)
    assert len(g_before) == 1 and len(g_before[0][2]) == 2
    leftover = ({0, 1, 2} - set(g_before[0][2])).pop()
    packed = host.pack_workers(sw)
    assert eng.append_workers(_rows(packed, np.array([3]))) == 3
    eng.set_addr_ranks(packed["addr_rank"])
    s = eng.tick()
    g_after = engine_groups(eng)
    assert s["n_formed"] == 1 and len(g_after) == 2
    assert g_after[0] == g_before[0]                                  # id, configuration, members, task: untouched
    assert sorted(g_after[1][2]) == sorted([leftover, 3])
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    nodes["status"][3] = 0                                            # not there yet
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks)
    st.try_form_new_groups()
    st.set_node_status(3, 2)
    st.try_form_new_groups()
    assert [g[:3] for g in oracle_groups(st)] == [g[:3] for g in g_after]
    assert eng.lookup(3).group_size == 2
    with pytest.raises(E.EngineError):
        eng.upload_workers(_rows(packed, np.arange(3)), keep_groups=True)   # keep_groups needs the same rows
    eng.close()


def test_appended_and_rewritten_rows_track_the_oracle():
    """2400 rows to start with, 3 x 200 appended, and every tick 60 known rows rewritten (new specs and location)
    and a dozen deaths; groups and every worker's task against the oracle after every tick."""
    rng = np.random.default_rng(11)
    W0, add, ticks = 2400, 200, 3
    sw_all = make_swarm(42, 1500, W0 + add * ticks)
    donor = make_swarm(43, 10, W0 + add * ticks)                      # a second draw of rows to rewrite from
    packed_all = host.pack_workers(sw_all)
    packed_join = {k: v.copy() for k, v in packed_all.items()}       # rows as they are when they first appear
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw_all)
    status_all = nodes["status"].copy()
    nodes["status"][W0:] = 0                                          # the oracle's table is fixed-size: not there yet
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    eng = E.Engine()
    sw0 = _take(sw_all, np.arange(W0))
    host.load_swarm(eng, sw0)
    eng.upload_workers(_rows(packed_all, np.arange(W0)))              # ranks over the final address set: stable
    W = W0

    def check(tag):
        s = eng.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        want = [st.get_task_for_node(w) for w in range(W)]          # (the oracle claims on this call)
        assert _tasks_of(eng, W) == want, tag
        assert sorted(oracle_groups(st)) == sorted(engine_groups(eng)), tag
        for w in rng.choice(W, size=100, replace=False):
            t, gi, gs, nxt = st.filter_tasks(int(w))
            a = eng.lookup(int(w))
            if t >= 0:
                assert (a.group_index, a.group_size, a.next_worker) == (gi, gs, nxt), (tag, w)
        assert s["host_resolved_steps"] == 0
        return s

    check("initial")
    for k in range(ticks):
        # ---- brand-new rows
        idx_new = np.arange(W, W + add)
        assert eng.append_workers(_rows(packed_join, idx_new)) == W
        for w in idx_new:
            st.set_node_status(int(w), int(status_all[w]))
        W += add
        # ---- known rows rewritten by the discovery sync: specs, location (grouped and ungrouped rows alike)
        idx_upd = rng.choice(W, size=60, replace=False)
        for f in _WORKER_FIELDS:
            if f not in ("address", "status"):
                getattr(sw_all, f)[idx_upd] = getattr(donor, f)[idx_upd]
        packed_all = host.pack_workers(sw_all)
        flags_now = packed_all["flags"].copy()
        cur_status = st.nodes["status"].copy()
        flags_now = np.where(cur_status == 2, flags_now | E.W_HEALTHY, flags_now & ~np.uint32(E.W_HEALTHY))
        packed_all["flags"] = flags_now.astype(np.uint32)
        eng.update_workers(idx_upd, _rows(packed_all, idx_upd))
        fresh = orc.from_swarm(sw_all)[0]
        for w in idx_upd:
            keep = int(st.nodes["status"][w])
            st.nodes[w] = fresh[w]
            st.nodes["status"][w] = keep
        # ---- deaths: the whole group dissolves
        alive = np.nonzero(st.nodes["status"][:W] == 2)[0]
        for w in rng.choice(alive, size=12, replace=False):
            st.set_node_status(int(w), 4)
            packed_all["flags"][w] &= ~np.uint32(E.W_HEALTHY)
            eng.on_worker_status(int(w), int(packed_all["flags"][w]), True)
        s = check(f"tick {k}")
        assert s["n_formed"] > 0
    eng.close()


def test_status_changes_then_rewrites_of_the_same_rows_then_new_rows_in_one_interval():
    """The order a busy interval has: status reports first (deaths, and dead nodes that come back), THEN the discovery sync
    rewrites known rows — some of them the very rows whose status has just changed — THEN brand-new rows join; one tick
    behind all of it.  pm_update_workers / pm_append_workers leave the pending status changes pending (they go up before the
    next kernel that reads the flags: what goes up is the host column's value at that time), so a row that is in both must end
    with what the LAST call said.  Groups and every worker's task against the oracle after every tick
    (status_update_impl.rs:8-39, discovery/monitor.rs:236-420, node_groups/mod.rs:487-497)."""
    rng = np.random.default_rng(23)
    W0, add, ticks = 2000, 150, 3
    sw_all = make_swarm(52, 1200, W0 + add * ticks)
    donor = make_swarm(53, 10, W0 + add * ticks)
    packed_all = host.pack_workers(sw_all)
    packed_join = {k: v.copy() for k, v in packed_all.items()}
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw_all)
    status_all = nodes["status"].copy()
    nodes["status"][W0:] = 0
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, _take(sw_all, np.arange(W0)))
    eng.upload_workers(_rows(packed_all, np.arange(W0)))
    W = W0

    def check(tag):
        eng.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        assert _tasks_of(eng, W) == [st.get_task_for_node(w) for w in range(W)], tag
        assert sorted(oracle_groups(st)) == sorted(engine_groups(eng)), tag

    check("initial")
    gone = np.zeros(0, dtype=np.int64)
    for k in range(ticks):
        # ---- status reports: deaths (the group dissolves), and last interval's dead reporting healthy again
        alive = np.nonzero(st.nodes["status"][:W] == 2)[0]
        victims = rng.choice(alive, size=40, replace=False)
        ws = np.concatenate([victims, gone])
        fl = np.concatenate([packed_all["flags"][victims] & ~np.uint32(E.W_HEALTHY), packed_all["flags"][gone] | np.uint32(E.W_HEALTHY)])
        dead = np.concatenate([np.ones(len(victims)), np.zeros(len(gone))]).astype(np.uint32)
        eng.on_worker_status_many(ws, fl.astype(np.uint32), dead)
        for w in victims:
            st.set_node_status(int(w), 4)
        for w in gone:
            st.set_node_status(int(w), 2)
        packed_all["flags"][victims] &= ~np.uint32(E.W_HEALTHY)
        packed_all["flags"][gone] |= np.uint32(E.W_HEALTHY)
        # ---- the discovery sync rewrites known rows: half of the rows whose status has just changed among them
        idx_upd = np.unique(np.concatenate([rng.choice(W, size=50, replace=False), victims[:20], gone[:10]])).astype(np.int64)
        for f in _WORKER_FIELDS:
            if f not in ("address", "status"):
                getattr(sw_all, f)[idx_upd] = getattr(donor, f)[idx_upd]
        packed_new = host.pack_workers(sw_all)
        cur_status = st.nodes["status"].copy()
        packed_new["flags"] = np.where(cur_status == 2, packed_new["flags"] | E.W_HEALTHY,
                                       packed_new["flags"] & ~np.uint32(E.W_HEALTHY)).astype(np.uint32)
        packed_all = packed_new
        eng.update_workers(idx_upd, _rows(packed_all, idx_upd))
        fresh = orc.from_swarm(sw_all)[0]
        for w in idx_upd:
            keep = int(st.nodes["status"][w])
            st.nodes[w] = fresh[w]
            st.nodes["status"][w] = keep
        # ---- brand-new rows behind both
        idx_new = np.arange(W, W + add)
        assert eng.append_workers(_rows(packed_join, idx_new)) == W
        for w in idx_new:
            st.set_node_status(int(w), int(status_all[w]))
        W += add
        packed_all["flags"][idx_new] = packed_join["flags"][idx_new]
        gone = victims
        check(f"tick {k}")
    eng.close()


def test_status_changes_in_one_call_equal_the_single_calls():
    """pm_on_worker_status_many == the same events through pm_on_worker_status, one by one (status_update_impl.rs:8-39:
    a death dissolves the whole group; the survivors re-group on the next tick)."""
    sw = make_swarm(11, 400, 1500)
    packed = host.pack_workers(sw)
    flags = packed["flags"].astype(np.int64)
    engines = []
    for _ in range(2):
        eng = E.Engine()
        host.load_swarm(eng, sw)
        eng.tick()
        engines.append(eng)
    a, b = engines
    assert engine_groups(a) == engine_groups(b)
    rng = np.random.default_rng(3)
    healthy = np.nonzero(sw.status == 2)[0]
    for tick in range(3):
        victims = rng.choice(healthy, size=40, replace=False)
        back = victims[:10]                                    # ten of them report healthy again in the same sweep
        ws = np.concatenate([victims, back])
        fl = np.concatenate([flags[victims] & ~E.W_HEALTHY, flags[back]])
        dead = np.concatenate([np.ones(len(victims)), np.zeros(len(back))]).astype(np.uint32)
        for w, f, d in zip(ws, fl, dead):
            a.on_worker_status(int(w), int(f), bool(d))
        b.on_worker_status_many(ws, fl, dead)
        assert engine_groups(a) == engine_groups(b)            # the dissolutions, before any tick
        a.tick()
        b.tick()
        assert engine_groups(a) == engine_groups(b)
        assert _tasks_of(a, sw.W) == _tasks_of(b, sw.W)
        for w in victims[10:]:                                 # the rest come back for the next round
            a.on_worker_status(int(w), int(flags[w]), False)
        b.on_worker_status_many(victims[10:], flags[victims[10:]])
    with pytest.raises(E.EngineError):
        b.on_worker_status_many([0, sw.W], [0, 0])             # out of range: nothing applied
    assert engine_groups(a) == engine_groups(b)
    a.close()
    b.close()
