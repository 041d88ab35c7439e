"""The call sequence of rust/gpu_match_plugin.rs (GpuMatchPlugin) against the C ABI, statement for statement — the
executable stand-in for the Rust binding, which cannot be compiled in this image.  Every method names the Rust
function it replays; the engine calls are made in the same order with the same arguments (ctypes through
protocol_amd.engine).  The "store" is a protocol_amd.swarm.Swarm; a node is a row id of it.

  ShimReplay(sw)                         GpuMatchPlugin::new: pm_engine_create, pm_set_configs (+ model table with no
                                         spec models yet), EMPTY pm_upload_workers, pm_enable_group_events(1), EMPTY
                                         pm_upload_tasks
  sync_nodes(snapshot)                   one get_nodes() snapshot, in ANY order: model table (new spec models),
                                         tombstones (pm_on_worker_status_many), rewritten rows (pm_update_workers with
                                         their current ranks), new rows (pm_append_workers + pm_set_addr_ranks), then
                                         the two-call pm_drain_group_events
  sync_tasks / on_task_created / on_task_deleted / handle_status_change / tick     as named
  get_all_groups / get_group_by_id / get_node_group (+ get_idx_in_group) / get_node_groups_batch /
  get_all_node_group_mappings / dissolve_group                                     the read surface the API routes call
                                         (node_groups/mod.rs:324-434, :1002-1065): pm_get_groups, pm_get_group_by_id,
                                         pm_get_group_of_worker, pm_dissolve_group_by_id + the drain
  Scheduler(store, [shim])               Scheduler::get_task_for_node with the INTEGRATION.md edit: a chain headed by
                                         the engine's plugin does not load the store's task list

Locks: the Rust shim's `tasks` RwLock is modelled by a flag — the observers hold it for writing from BEFORE the engine
call until the Vec has changed (lock order: nodes, tasks, engine mutex), and filter_tasks, which takes it for reading
before the look-up, raises WouldBlock while it is held (in Rust the heartbeat waits there).  `mid_observer` is called
between the two halves of an observer: the place where another thread's heartbeat could land.
"""
from __future__ import annotations

import bisect
import ctypes as C

import numpy as np

from protocol_amd import engine as E
from protocol_amd import host

_ROW_FIELDS = ("flags", "gpu_count", "gpu_mem_mb", "gpu_model_class", "cpu_cores", "ram_mb", "storage_gb", "price",
               "lat", "lon")


class WouldBlock(Exception):
    """filter_tasks met the `tasks` write lock: the Rust heartbeat waits here until the observer is through"""


class TaskStore:
    """the store side of Scheduler::get_task_for_node: hands out the task list, counts how often it was asked"""

    def __init__(self):
        self.tasks: list = []
        self.loads = 0

    def get_all_tasks(self):
        self.loads += 1
        return list(self.tasks)


class Scheduler:
    """scheduler/mod.rs:9-36 with the edit of INTEGRATION.md ("The task list per heartbeat")"""

    def __init__(self, store: TaskStore, plugins: list):
        self.store, self.plugins = store, plugins

    def get_task_for_node(self, node: int):
        head = self.plugins[0] if self.plugins else None
        all_tasks = [] if isinstance(head, ShimReplay) else self.store.get_all_tasks()
        for plugin in self.plugins:
            all_tasks = plugin.filter_tasks(all_tasks, node)
        return all_tasks[0] if all_tasks else None


class ShimReplay:
    def __init__(self, sw, **engine_kw):
        self.sw = sw
        self.tasks_write_locked = False                        # the `tasks` RwLock, write side
        self.mid_observer = None                               # called between the halves of a task observer
        self.hold_tasks_lock = True                            # False = the round-3 order (engine call, THEN the lock)
        self.republish_on_insert = False                       # GpuMatchPlugin::republish_on_insert
        self.eng = E.Engine(**engine_kw)                       # pm_engine_config_default + pm_engine_create
        cfg_rows, alt_rows, self.req_models = host.pack_configs(sw.configs)
        self.eng.set_configs(cfg_rows, alt_rows)               # set_configs
        self.spec_models: list = []                            # interned in first-seen order (project)
        self.spec_index: dict = {}
        self._push_model_table()                               # push_model_table(&NodeTable::default())
        self.packed_all = host.pack_workers(sw)                # the projection of every store row (project)
        self.flags_all = self.packed_all["flags"].copy()
        # NodeTable: index (address -> row), rows, present
        self.index: dict = {}
        self.node_of_row: list = []                            # engine row -> store row
        self.rows: list = []                                   # projected row (tuple) per engine row
        self.present: list = []
        self.sorted_rows: list = []                            # engine rows in address-string order (incremental ranks)
        self.sorted_keys: list = []
        self.events: list = []                                 # what send_group_created / _destroyed were called with
        self.engine_rows_stale = False                         # NodeTable::engine_rows_stale
        self.clock = lambda: 0                                 # chrono::Utc::now() (NodeGroup.created_at)
        self.group_created_at: dict = {}                       # GpuMatchPlugin::group_created_at
        self.addr_strings = sw.address_strings()
        self.tasks: list = []                                  # the shim's Vec<Task>: uids in list order
        empty = {f: np.zeros(0, dtype=self.packed_all[f].dtype) for f in _ROW_FIELDS}
        empty["addr_rank"] = np.zeros(0, dtype=np.uint32)
        self.eng.upload_workers(empty)                         # pm_upload_workers(&empty, 0)
        self.eng.enable_group_events(True)                     # pm_enable_group_events(1)
        self.eng.upload_tasks(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.uint64))

    # ---- helpers of the shim
    def _push_model_table(self):
        bits = host.build_model_table(self.req_models, self.spec_models)
        self.eng.set_model_table(bits, len(self.req_models), len(self.spec_models))

    def _project(self, node: int, status_healthy: bool):
        """project(): one OrchestratorNode -> Row; interns its GPU model in first-seen order"""
        new_model = False
        row = {f: self.packed_all[f][node] for f in _ROW_FIELDS}
        flags = int(row["flags"]) & ~E.W_HEALTHY
        if status_healthy:
            flags |= E.W_HEALTHY
        row["flags"] = np.uint32(flags)
        if flags & E.W_GPU_MODEL:
            name = self.sw.model_names[int(row["gpu_model_class"])]
            if name not in self.spec_index:
                self.spec_index[name] = len(self.spec_models)
                self.spec_models.append(name)
                new_model = True
            row["gpu_model_class"] = np.uint32(self.spec_index[name])
        return row, new_model

    def _ranks(self) -> np.ndarray:
        """address_ranks(): rank of address.to_string() among the known rows (kept sorted incrementally)"""
        rank = np.zeros(len(self.rows), dtype=np.uint32)
        rank[np.array(self.sorted_rows, dtype=np.int64)] = np.arange(len(self.sorted_rows), dtype=np.uint32)
        return rank

    @staticmethod
    def _columns(rows: list, ranks) -> dict:
        out = {f: np.array([r[f] for r in rows]) for f in _ROW_FIELDS}
        out["addr_rank"] = np.asarray(ranks, dtype=np.uint32)
        return out

    def _emit_group_webhooks(self):
        """emit_group_webhooks(): size query, then the drain (Engine.drain_group_events makes exactly these two calls)"""
        ev = self.eng.drain_group_events()
        now = self.clock()
        for kind, gid, _cfg, _mem in ev:
            if kind == E.GROUP_CREATED:
                self.group_created_at[gid] = now
            else:
                self.group_created_at.pop(gid, None)
        self.events.extend(ev)

    # ---- the plugin surface
    def sync_nodes(self, snapshot, healthy):
        """snapshot: store rows present now, in the store's order of the day; healthy: set of store rows with
        status Healthy"""
        seen = [False] * len(self.rows)
        new_model = False
        appended, upd_idx, updated = [], [], []
        for node in snapshot:
            row, nm = self._project(int(node), int(node) in healthy)
            new_model |= nm
            i = self.index.get(int(node))
            if i is not None:
                seen[i] = True
                if any(self.rows[i][f] != row[f] for f in _ROW_FIELDS) or not self.present[i]:
                    self.rows[i] = row
                    self.present[i] = True
                    upd_idx.append(i)
                    updated.append(row)
            else:
                i = len(self.rows)
                self.index[int(node)] = i
                self.node_of_row.append(int(node))
                self.rows.append(row)
                self.present.append(True)
                key = int(self.sw.address[int(node)])          # (digit-only address strings of one length: the
                k = bisect.bisect_left(self.sorted_keys, key)  # integer order is the byte order of the strings)
                self.sorted_keys.insert(k, key)
                self.sorted_rows.insert(k, i)
                appended.append(row)
        if self.engine_rows_stale:                             # every row again, groups kept, then the deaths the engine missed
            for i in range(len(seen)):
                if not seen[i]:
                    self.present[i] = False
            gone, gone_flags = [], []
            for i in range(len(self.rows)):
                if not self.present[i]:
                    self.rows[i]["flags"] = np.uint32(int(self.rows[i]["flags"]) & ~E.W_HEALTHY)
                    gone.append(i)
                    gone_flags.append(int(self.rows[i]["flags"]))
            self._push_model_table()
            self.eng.upload_workers(self._columns(self.rows, self._ranks()), keep_groups=True)
            if gone:
                self.eng.on_worker_status_many(np.array(gone), np.array(gone_flags, dtype=np.uint32), np.ones(len(gone), dtype=np.uint32))
            self.engine_rows_stale = False
            self._emit_group_webhooks()
            return
        self.engine_rows_stale = True                          # (cleared behind the last engine call below)
        if new_model:
            self._push_model_table()
        gone, gone_flags = [], []
        for i in range(len(seen)):
            if not seen[i] and self.present[i]:
                self.present[i] = False
                self.rows[i]["flags"] = np.uint32(int(self.rows[i]["flags"]) & ~E.W_HEALTHY)
                gone.append(i)
                gone_flags.append(int(self.rows[i]["flags"]))
        if gone:
            self.eng.on_worker_status_many(np.array(gone), np.array(gone_flags, dtype=np.uint32),
                                           np.ones(len(gone), dtype=np.uint32))
        if upd_idx:
            ranks_known = np.zeros(len(seen), dtype=np.uint32)     # ranks among the rows the engine already has
            order = [i for i in self.sorted_rows if i < len(seen)]
            ranks_known[np.array(order, dtype=np.int64)] = np.arange(len(order), dtype=np.uint32)
            self.eng.update_workers(np.array(upd_idx), self._columns(updated, ranks_known[np.array(upd_idx)]))
        if appended:
            first = self.eng.append_workers(self._columns(appended, np.zeros(len(appended), dtype=np.uint32)))
            if first != len(seen):
                raise RuntimeError("the engine's worker table and the plugin's row map disagree")
            self.eng.set_addr_ranks(self._ranks())
        self.engine_rows_stale = False
        self._emit_group_webhooks()

    def _push_enabled(self):
        self.eng.set_enabled_mask(self._enabled)

    def _between_halves(self):
        if self.mid_observer is not None:
            self.mid_observer()

    def sync_tasks(self, masks, created, uid, enabled):
        self.tasks_write_locked = self.hold_tasks_lock         # let mut guard = self.tasks.write();
        try:
            self.eng.upload_tasks(masks, created, uid)         # sync_tasks_locked
            self._enabled = enabled
            self._push_enabled()
            self._between_halves()
            self.tasks = [int(u) for u in uid]
        finally:
            self.tasks_write_locked = False

    def on_task_created(self, mask, created, uid, enabled):
        self.tasks_write_locked = self.hold_tasks_lock         # let mut tasks = self.tasks.write();  (before the engine)
        try:
            self.eng.tasks_insert_front(np.array([mask], dtype=np.uint64), np.array([created], dtype=np.int64),
                                        np.array([uid], dtype=np.uint64), republish=self.republish_on_insert)
            self._between_halves()
            self.tasks.insert(0, int(uid))
            self._enabled = enabled
            self._push_enabled()
        finally:
            self.tasks_write_locked = False

    def on_task_deleted(self, uid, enabled):
        self.tasks_write_locked = self.hold_tasks_lock         # let mut tasks = self.tasks.write();  (before the engine)
        try:
            assert self.eng.tasks_delete(np.array([uid], dtype=np.uint64)) == 1
            self._between_halves()
            self.tasks.remove(int(uid))
            self._enabled = enabled
            self._push_enabled()
        finally:
            self.tasks_write_locked = False                    # drop(tasks)
        self._emit_group_webhooks()

    def handle_status_change(self, node: int, healthy: bool, dead: bool):
        w = self.index.get(int(node))
        if w is None:
            return
        flags = int(self.rows[w]["flags"]) & ~E.W_HEALTHY
        if healthy:
            flags |= E.W_HEALTHY
        self.rows[w]["flags"] = np.uint32(flags)
        self.eng.on_worker_status(w, flags, dead)
        if dead:
            self._emit_group_webhooks()

    def tick(self):
        s = self.eng.tick()
        self._emit_group_webhooks()
        return s

    def filter_tasks(self, *args):
        """filter_tasks(node) -> uid of the task the node gets, or None; filter_tasks(tasks, node) -> [uid] or [] (the
        plugin-chain form: `tasks` is ignored, as in the Rust).  tasks.read() is taken BEFORE pm_lookup_task_for_worker
        and kept until the Vec<Task> has been indexed."""
        node = args[-1]
        as_list = len(args) == 2
        w = self.index.get(int(node))
        if w is None:
            return [] if as_list else None
        if self.tasks_write_locked:                            # let tasks = self.tasks.read();
            raise WouldBlock
        a = self.eng.lookup(w)
        uid = None if (a.task == 0xFFFFFFFF or a.task >= len(self.tasks)) else self.tasks[a.task]
        if as_list:
            return [] if uid is None else [uid]
        return uid

    # ---- the read surface (same record as tests/plugin_cxx.PluginCxx: {"id", "config", "created_at", "nodes"})
    def _make_group(self, gid: int, cfg: int, members) -> dict:
        return {"id": "%x" % gid, "config": self.sw.configs[cfg][0],
                "created_at": self.group_created_at.get(gid, self.clock()),
                "nodes": [self.addr_strings[self.node_of_row[int(w)]] for w in members]}

    @staticmethod
    def _parse_group_id(text: str):
        if not text or len(text) > 16 or (len(text) > 1 and text[0] == "0") or any(c not in "0123456789abcdef" for c in text):
            return None
        return int(text, 16)

    def _row_of_address_text(self, text: str):
        node = next((n for n in self.index if self.addr_strings[n] == text), None)   # (the Rust: a binary search)
        return None if node is None else self.index[node]

    def get_all_groups(self):
        _gow, groups, members = self.eng.get_groups()
        out = [self._make_group(int(g["id"]), int(g["config"]), members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])])
               for g in groups]
        return sorted(out, key=lambda g: g["id"])                                     # mod.rs:1040

    def get_group_by_id(self, group_id: str):
        gid = self._parse_group_id(group_id)
        if gid is None:
            return None
        g = self.eng.get_group_by_id(gid)
        return None if g is None else self._make_group(g["id"], g["config"], g["members"])

    def get_node_group(self, address: str):
        """-> None or (get_idx_in_group, group)"""
        row = self._row_of_address_text(address)
        if row is None:
            return None
        try:
            g = self.eng.get_group_of_worker(row)
        except E.EngineError as ex:
            if ex.code == E.PM_ERANGE:                                                # a row the engine was never sent
                return None
            raise
        if g is None:
            return None
        group = self._make_group(g["id"], g["config"], g["members"])
        return group["nodes"].index(address), group

    def get_node_groups_batch(self, addresses):
        if not addresses:
            return {}
        gow, groups, members = self.eng.get_groups()
        out = {}
        for a in addresses:
            row = self._row_of_address_text(a)
            gi = -1 if row is None or row >= len(gow) else int(gow[row])
            if gi < 0:
                out[a] = None
            else:
                g = groups[gi]
                out[a] = self._make_group(int(g["id"]), int(g["config"]), members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])])
        return out

    def get_all_node_group_mappings(self):
        return {n: g["id"] for g in self.get_all_groups() for n in g["nodes"]}

    def dissolve_group(self, group_id: str):
        gid = self._parse_group_id(group_id)
        if gid is None:
            return
        if self.eng.dissolve_group_by_id(gid):
            self._emit_group_webhooks()

    def set_clock(self, now_ms: int):
        self.clock = lambda: int(now_ms)

    def close(self):
        self.eng.close()
