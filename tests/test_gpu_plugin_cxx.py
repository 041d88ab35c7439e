"""The compiled host side (protocol_amd/plugin: the C++ GpuMatchPlugin + Scheduler, libpm_plugin.so over libpm_engine.so)
against the oracle, with the schedule tests/test_gpu_shim_replay.py drives the Python replay of the Rust shim with
(tests/plugin_cxx.py gives the C++ objects that interface).

Part of the default `-m gpu` run since round 5 (it was opt-in for the one round in which it had been written after
the GPU budget was spent)."""
import os

import numpy as np
import pytest

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import make_swarm
from helpers import engine_groups, oracle_groups
from plugin_cxx import PluginCxx
from test_gpu_ingest import _WORKER_FIELDS

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


def test_cxx_plugin_call_sequence_tracks_the_oracle():
    """tests/test_gpu_shim_replay.py::test_shim_call_sequence_tracks_the_oracle with the COMPILED host side in the shim's
    place: five management intervals (snapshots in a new order each, new / rewritten / departed nodes), task observers,
    deaths, ticks; every node's heartbeat through the C++ Scheduler, the engine's groups and the webhook feed against
    the oracle."""
    rng = np.random.default_rng(5)
    W_store, T0, ticks = 1600, 300, 5
    sw = make_swarm(31, T0, W_store)
    donor = make_swarm(32, 10, W_store)                    # a second draw of rows: what discovery rewrites rows to
    # ---- the schedule of the store, decided up front so that the oracle can be given its rows in first-seen order
    present = set(range(900))
    pool = list(range(900, W_store))
    snapshots, leavers, rewritten = [], [], []
    for k in range(ticks):
        if k:
            gone = rng.choice(sorted(present), size=25, replace=False)
            present.difference_update(int(x) for x in gone)
            leavers.append(gone)
            new = [pool.pop() for _ in range(100)]
            present.update(new)
            rewritten.append(rng.choice(sorted(present - set(new)), size=20, replace=False))
        else:
            leavers.append(np.zeros(0, dtype=np.int64))
            rewritten.append(np.zeros(0, dtype=np.int64))
        snap = np.array(sorted(present))
        rng.shuffle(snap)                                   # the store hands its nodes out in another order each time
        snapshots.append(snap)
    first_seen = []
    seen = set()
    for snap in snapshots:
        for node in snap:
            if int(node) not in seen:
                seen.add(int(node))
                first_seen.append(int(node))
    row_of = {node: i for i, node in enumerate(first_seen)}
    # ---- oracle: the same rows in first-seen order, absent until they are first seen
    nodes_all, cfgs, tasks, enabled = orc.from_swarm(sw)
    status_all = nodes_all["status"].copy()
    nodes = nodes_all[np.array(first_seen)].copy()
    nodes["status"] = 0
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks[:0], reference_shaped=False)
    # ---- the shim
    shim = PluginCxx(sw)                                    # GpuMatchPlugin::new + Scheduler::new(store, [plugin])
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    shim.sync_tasks(masks, created, uid, sw.enabled_mask())            # start-up: the full snapshot
    st.set_tasks(tasks)
    cur_tasks, cur_uid = tasks.copy(), [int(u) for u in uid]
    healthy = {n for n in range(W_store) if status_all[n] == orc.ST_HEALTHY}
    next_uid = 1 << 42
    t_max = int(created.max())
    n_created = n_destroyed = 0
    for k in range(ticks):
        # ---- discovery rewrote some rows (specs, location) since the last interval
        for node in rewritten[k]:
            for f in _WORKER_FIELDS:
                if f not in ("address", "status"):
                    getattr(sw, f)[node] = getattr(donor, f)[node]
        if len(rewritten[k]):
            shim.packed_all = host.pack_workers(sw)
            fresh = orc.from_swarm(sw)[0]
            for node in rewritten[k]:
                i = row_of[int(node)]
                keep = int(st.nodes["status"][i])
                st.nodes[i] = fresh[node]
                st.nodes["status"][i] = keep
        # ---- sync_nodes with today's snapshot
        shim.sync_nodes(snapshots[k], healthy)
        for i in sorted(row_of[int(node)] for node in leavers[k]):      # (the shim reports them in row order)
            st.set_node_status(i, orc.ST_DEAD)                         # a departed node: its group dissolves
        for node in snapshots[k]:
            i = row_of[int(node)]
            if st.nodes["status"][i] == 0 and int(node) in healthy:
                st.set_node_status(i, orc.ST_HEALTHY)
            elif st.nodes["status"][i] == 0:
                st.set_node_status(i, int(status_all[node]))
        # ---- task observers: two new tasks, from the third interval on one claimed task deleted
        for _ in range(2):
            src = int(rng.integers(0, len(masks)))
            t_max += 1
            shim.on_task_created(int(masks[src]), t_max, next_uid, sw.enabled_mask())
            row = tasks[src:src + 1].copy()
            row["created_at"] = t_max
            n_old = len(cur_tasks)
            cur_tasks = np.concatenate([row, cur_tasks])
            cur_uid.insert(0, next_uid)
            st.set_tasks(cur_tasks)
            st.remap_tasks(np.arange(n_old) + 1)
            next_uid += 1
        if k >= 2:
            claimed = [g[4] for g in st.groups() if g[4] >= 0]
            if claimed:
                victim = max(set(claimed), key=claimed.count)
                shim.on_task_deleted(cur_uid[victim], sw.enabled_mask())
                keep = np.ones(len(cur_tasks), dtype=bool)
                keep[victim] = False
                cur_tasks = cur_tasks[keep]
                del cur_uid[victim]
                st.set_tasks(cur_tasks)
                st.remap_tasks(np.where(keep, np.cumsum(keep) - 1, -1))
        # ---- a few deaths through handle_status_change
        alive_rows = np.nonzero(st.nodes["status"] == orc.ST_HEALTHY)[0]
        for i in rng.choice(alive_rows, size=8, replace=False):
            node = first_seen[int(i)]
            healthy.discard(node)
            shim.handle_status_change(node, healthy=False, dead=True)
            st.set_node_status(int(i), orc.ST_DEAD)
        # ---- the management interval, then every node's heartbeat
        shim.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        for i, node in enumerate(first_seen[:len(shim.rows)]):        # (the oracle claims on this call)
            t = st.get_task_for_node(i)
            assert shim.filter_tasks(node) == (None if t < 0 else cur_uid[t]), (k, node)   # Scheduler::get_task_for_node
        assert sorted(oracle_groups(st)) == sorted(engine_groups(shim.eng)), f"interval {k}"
        ev = st.drain_events()
        assert shim.events == ev, f"interval {k}: the webhook feed differs"
        n_created += sum(e[0] == E.GROUP_CREATED for e in ev)
        n_destroyed += sum(e[0] == E.GROUP_DESTROYED for e in ev)
        shim.events.clear()
    assert n_created > 100 and n_destroyed > 20 and len(shim.rows) == len(first_seen)
    assert shim.store_loads == 0, "a chain headed by the engine's plugin must not load the store's task list per heartbeat"
    shim.close()
