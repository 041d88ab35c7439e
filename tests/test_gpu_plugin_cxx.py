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


def _check_read_surface(shim, st, first_seen, sw, stamps, clock_now):
    """The plugin's read surface (what the API routes call, node_groups/mod.rs:324-434, :1002-1065) against the oracle's
    group state: get_all_groups (sorted by id text), get_group_by_id, get_node_group + get_idx_in_group for every known
    node, get_node_groups_batch, get_all_node_group_mappings, and the storage route's upload name rendered through them."""
    addr = sw.address_strings()
    names = shim.config_names
    og = st.groups()
    want = sorted((("%x" % gid, names[cfg], [addr[first_seen[i]] for i in mem]) for (_s, gid, cfg, mem, _t) in og),
                  key=lambda g: g[0])                                                # mod.rs:1040: by id TEXT
    got = shim.get_all_groups()
    assert [(g["id"], g["config"], g["nodes"]) for g in got] == want
    for g in got:   # created_at: a group keeps the stamp of the interval that formed it
        assert stamps.setdefault(g["id"], clock_now) == g["created_at"], g
    by_slot = {s: ("%x" % gid, names[cfg], [addr[first_seen[i]] for i in mem]) for (s, gid, cfg, mem, _t) in og}
    n2g = st.node_to_group
    known = list(first_seen[:len(shim.rows)])
    want_map, per_node = {}, {}
    for i, node in enumerate(known):
        r = shim.get_node_group(addr[node])
        if n2g[i] < 0:
            assert r is None, (i, r)
            per_node[addr[node]] = None
            continue
        gid, cfg, nodes = by_slot[int(n2g[i])]
        _t, gi, gs, _nxt = st.filter_tasks(i)                                        # the oracle's GROUP_INDEX / GROUP_SIZE
        assert r is not None and r[0] == gi == nodes.index(addr[node]) and len(nodes) == gs
        assert (r[1]["id"], r[1]["config"], r[1]["nodes"]) == (gid, cfg, nodes)
        want_map[addr[node]] = gid
        per_node[addr[node]] = (gid, cfg, nodes)
    assert shim.get_all_node_group_mappings() == want_map
    asked = [addr[n] for n in known] + ["0x" + "9" * 40]                             # + an address nobody has
    batch = shim.get_node_groups_batch(asked)
    assert set(batch) == set(asked) and batch[asked[-1]] is None
    for a in asked[:-1]:
        b = batch[a]
        assert (None if b is None else (b["id"], b["config"], b["nodes"])) == per_node[a], a
    for gid, cfg, nodes in list(by_slot.values())[:40]:
        g = shim.get_group_by_id(gid)
        assert g is not None and (g["id"], g["config"], g["nodes"]) == (gid, cfg, nodes)
    assert shim.get_group_by_id("f" * 16 if "f" * 16 not in stamps else "e" * 16) is None
    # the storage route (api/routes/storage.rs:147-207): the upload's name through get_node_group + get_idx_in_group
    tmpl = "m/${NODE_GROUP_ID}-${NODE_GROUP_SIZE}-${NODE_GROUP_INDEX}-${TOTAL_UPLOAD_COUNT_AFTER}-${CURRENT_FILE_INDEX}.parquet"
    shim.set_upload_count(3)
    for a in asked[:60]:
        name, key = shim.upload_file_name(tmpl, a)
        g = per_node[a]
        if g is None:
            assert (name, key) == (orc.upload_name_vars(tmpl, None, 0, 0, 3), "no-group")
        else:
            assert (name, key) == (orc.upload_name_vars(tmpl, g[0], len(g[2]), g[2].index(a), 3), g[0])


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
    stamps: dict = {}
    assert [c[0] for c in shim.get_configurations(False)] == [shim.config_names[c] for c in host.config_order(host.pack_configs(sw.configs)[0], (1 << len(sw.configs)) - 1)]
    for k in range(ticks):
        shim.set_clock(1000 * (k + 1))
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
        _check_read_surface(shim, st, first_seen, sw, stamps, 1000 * (k + 1))
        # ---- the admin route: DELETE /groups/{id} (api/routes/groups.rs:126) on three groups, by id text
        victims = st.groups()[1::max(len(st.groups()) // 3, 1)][:3]
        for (slot, gid, _cfg, _mem, _t) in victims:
            shim.dissolve_group("%x" % gid)
            st.dissolve_group(slot)
        shim.dissolve_group("%x" % victims[0][1])                      # already gone: Ok(()), nothing happens
        assert shim.events == st.drain_events() and len(shim.events) == len(victims)
        n_destroyed += len(victims)
        shim.events.clear()
        assert sorted(oracle_groups(st)) == sorted(engine_groups(shim.eng))
        for (_slot, gid, _cfg, mem, _t) in victims:
            assert shim.get_group_by_id("%x" % gid) is None
            assert shim.get_node_group(sw.address_strings()[first_seen[mem[0]]]) is None
            assert shim.filter_tasks(first_seen[mem[0]]) is None     # the heartbeat of a freed node: no group, no task
    avail = [c[0] for c in shim.get_configurations(True)]
    assert avail and set(avail) <= set(shim.config_names)
    assert n_created > 100 and n_destroyed > 20 and len(shim.rows) == len(first_seen)
    assert shim.store_loads == 0, "a chain headed by the engine's plugin must not load the store's task list per heartbeat"
    shim.close()


def test_cxx_scheduler_renders_the_task_every_worker_receives():
    """SURVEY 8(f)3 end to end on the GPU path: the task a worker is handed by Scheduler::get_task_for_node on the real
    engine — env values, cmd arguments and volume mounts with ${GROUP_INDEX} ${GROUP_SIZE} ${NEXT_P2P_ADDRESS} ${GROUP_ID}
    ${TOTAL_UPLOAD_COUNT} ${LAST_FILE_IDX} (scheduler_impl.rs:112-205), then ${TASK_ID} ${NODE_ADDRESS} ${TIMESTAMP}
    (scheduler/mod.rs:34-70, shared/models/task.rs:75-98) — string for string against the oracle's orc_group_vars /
    orc_volume_vars fed with the oracle's own filter_tasks numbers, for every worker of a 1,600-node swarm, over two
    intervals with deaths in between."""
    W, T = 1600, 300
    sw = make_swarm(41, T, W)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    shim = PluginCxx(sw)
    env = {"IDX": "${GROUP_INDEX}/${GROUP_SIZE}", "NEXT": "peer=${NEXT_P2P_ADDRESS}", "WHO": "${TASK_ID}@${NODE_ADDRESS}",
           "GID": "${GROUP_ID}-${GROUP_ID}", "UP": "${TOTAL_UPLOAD_COUNT}:${LAST_FILE_IDX}", "PLAIN": "no variables", "GROUP_INDEX": "overwritten"}
    cmd = ["--rank=${GROUP_INDEX}", "--world=${GROUP_SIZE}", "--next", "${NEXT_P2P_ADDRESS}", "--out=/data/${GROUP_ID}/${TASK_ID}/${LAST_FILE_IDX}",
           "--node=${NODE_ADDRESS}", "${GROUP_SIZE}${GROUP_SIZE}"]
    mounts = [("/host/${GROUP_ID}/${TASK_ID}", "/mnt/${NODE_ADDRESS}/${TIMESTAMP}"), ("/scratch/${GROUP_INDEX}", "/g/${GROUP_ID}")]
    shim.sync_tasks(sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy(), sw.enabled_mask(), env=env, cmd=cmd, mounts=mounts)
    shim.set_upload_count(7)
    healthy = {n for n in range(W) if sw.status[n] == 2}
    addr = sw.address_strings()
    from plugin_cxx import uuid_of
    rng = np.random.default_rng(7)
    served = 0
    for interval in range(2):
        shim.sync_nodes(np.arange(W), healthy)
        shim.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        gid_of_slot = {s: "%x" % gid for (s, gid, _c, _m, _t) in st.groups()}
        now = 1_700_000_000 + interval
        for w in range(W):
            t, gi, gs, nxt = st.filter_tasks(w)                                     # (the oracle claims on this call)
            got = shim.task_for_node(w, now=now)
            if t < 0:
                assert got is None, w
                continue
            served += 1
            gid = gid_of_slot[int(st.node_to_group[w])]
            tid = uuid_of(int(sw.task_uid[t]))
            assert got["id"] == tid
            gv = lambda s: orc.group_vars(s, gi, gs, "p2p-%d" % nxt, gid, "7")      # scheduler_impl.rs:160-183
            sv = lambda s: s.replace("${TASK_ID}", tid).replace("${NODE_ADDRESS}", addr[w])   # scheduler/mod.rs:38-54
            want_env = dict(env)
            want_env["GROUP_INDEX"] = str(gi)                                        # scheduler_impl.rs:161: inserted, THEN expanded
            assert got["env"] == {k: sv(gv(v)) for k, v in want_env.items()}, w
            assert got["cmd"] == [sv(gv(a)) for a in cmd], w
            assert got["mounts"] == [tuple(sv(orc.volume_vars(p, gid)).replace("${TIMESTAMP}", str(now)) for p in m) for m in mounts], w
        if interval == 0:   # a few deaths: their groups dissolve, the second interval re-forms from the leftovers
            for w in rng.choice(sorted(healthy), size=40, replace=False):
                healthy.discard(int(w))
                shim.handle_status_change(int(w), healthy=False, dead=True)
                st.set_node_status(int(w), orc.ST_DEAD)
    assert served > 1500
    assert shim.store_loads == 0
    shim.close()
