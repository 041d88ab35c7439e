"""ctypes binding of include/pm_engine.h (libpm_engine.so).

This is plumbing for tests and bench.py: every call goes straight through the C ABI.  Loading fails
loudly if the library has not been built, and Engine() fails with PM_ENODEV when no MI355X is
visible — there is no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

PM_NONE = 0xFFFFFFFF
PM_MAX_CONFIGS = 64
PM_OK, PM_EINVAL, PM_ENODEV, PM_ENOMEM, PM_ESTATE, PM_ERANGE, PM_EPARSE, PM_EPANIC = 0, -1, -2, -3, -4, -5, -6, -7
CHOOSE_FIRST, CHOOSE_SEEDED = 0, 1

# worker flag bits
(W_HAS_SPECS, W_HAS_GPU, W_GPU_COUNT, W_GPU_MEM, W_GPU_MODEL, W_HAS_CPU, W_CPU_CORES, W_RAM, W_STORAGE,
 W_HEALTHY, W_HAS_P2P, W_HAS_LOC) = (1 << i for i in range(12))
R_HAS_REQ, R_CPU, R_CPU_CORES, R_RAM, R_STORAGE = (1 << i for i in range(5))
G_COUNT, G_MODEL, G_MEM, G_MEM_MIN, G_MEM_MAX, G_TOT_MIN, G_TOT_MAX = (1 << i for i in range(7))

config_row_dt = np.dtype([("flags", "<u4"), ("cpu_cores", "<u4"), ("ram_mb", "<u4"), ("storage_gb", "<u4"),
                          ("alt_begin", "<u4"), ("alt_count", "<u4"), ("min_group_size", "<u4"),
                          ("max_group_size", "<u4")], align=True)
alt_row_dt = np.dtype([("flags", "<u4"), ("count", "<u4"), ("memory_mb", "<u4"), ("memory_mb_min", "<u4"),
                       ("memory_mb_max", "<u4"), ("total_memory_min", "<u4"), ("total_memory_max", "<u4"),
                       ("model_row", "<u4")], align=True)
group_dt = np.dtype([("id", "<u8"), ("config", "<u4"), ("n_members", "<u4"), ("member_begin", "<u4"),
                     ("task", "<u4")], align=True)
GROUP_EVENT = np.dtype([("group_id", "<u8"), ("kind", "<u4"), ("config", "<u4"), ("member_begin", "<u4"),
                        ("n_members", "<u4")])   # pm_group_event
GROUP_CREATED, GROUP_DESTROYED = 1, 2
assignment_dt = np.dtype([("task", "<u4"), ("group_slot", "<u4"), ("group_index", "<u4"), ("group_size", "<u4"),
                          ("next_worker", "<u4"), ("group_id", "<u8")], align=True)
assert config_row_dt.itemsize == 32 and alt_row_dt.itemsize == 32 and assignment_dt.itemsize == 32


class WorkerSoa(C.Structure):
    _fields_ = [("n", C.c_uint32)] + [(k, C.c_void_p) for k in
                                      ("flags", "gpu_count", "gpu_mem_mb", "gpu_model_class", "cpu_cores", "ram_mb",
                                       "storage_gb", "price", "addr_rank", "lat", "lon")]


class TaskSoa(C.Structure):
    _fields_ = [("n", C.c_uint32), ("topo_mask", C.c_void_p), ("created_at", C.c_void_p), ("uid", C.c_void_p)]


class EngineConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("proximity_enabled", C.c_uint32),
                ("switching_enabled", C.c_uint32), ("prefer_larger_groups", C.c_uint32), ("chooser", C.c_uint32),
                ("chooser_seed", C.c_uint64), ("group_id_seed", C.c_uint64), ("debug_uncertain_every", C.c_uint32),
                ("sweep_variant", C.c_uint32), ("carve_variant", C.c_uint32), ("time_proposer", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("ms_compat", C.c_float), ("ms_carve", C.c_float), ("ms_merge", C.c_float), ("ms_sweep", C.c_float),
                ("ms_publish", C.c_float), ("ms_total", C.c_float), ("ms_compat_kernel", C.c_float),
                ("ms_carve_kernel", C.c_float), ("ms_sweep_kernel", C.c_float), ("n_groups", C.c_uint32),
                ("n_formed", C.c_uint32), ("n_merged", C.c_uint32), ("carve_steps", C.c_uint32),
                ("carve_fast_steps", C.c_uint32),
                ("host_resolved_steps", C.c_uint32), ("carve_launches", C.c_uint32), ("pair_evals", C.c_uint64),
                ("carve_cand_sum", C.c_uint64), ("ms_propose_kernel", C.c_float), ("proposals", C.c_uint32),
                ("propose_keys", C.c_uint64)]

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k, _ in self._fields_}


class GroupVars(C.Structure):
    _fields_ = [("group_index", C.c_uint32), ("group_size", C.c_uint32), ("next_p2p_address", C.c_char_p),
                ("group_id", C.c_char_p), ("total_upload_count", C.c_char_p)]


class DistXfer(C.Structure):
    """pm_dist_xfer: device pointers + size of one all-gather (recv = [world][bytes_per_rank])."""
    _fields_ = [("send_ptr", C.c_uint64), ("recv_ptr", C.c_uint64), ("bytes_per_rank", C.c_uint64)]


class Assignment(C.Structure):
    _fields_ = [("task", C.c_uint32), ("group_slot", C.c_uint32), ("group_index", C.c_uint32),
                ("group_size", C.c_uint32), ("next_worker", C.c_uint32), ("group_id", C.c_uint64)]


# every symbol include/pm_engine.h declares (tests check the library exports all of them)
EXPORTS = [
    "pm_engine_config_default", "pm_engine_create", "pm_engine_destroy", "pm_last_error", "pm_set_configs",
    "pm_set_model_table", "pm_set_enabled_mask", "pm_upload_workers", "pm_update_workers", "pm_upload_tasks",
    "pm_on_worker_status", "pm_on_worker_status_many", "pm_enable_group_events", "pm_drain_group_events",
    "pm_dissolve_group", "pm_reset_groups", "pm_compat_masks", "pm_form_groups",
    "pm_merge_solo_groups", "pm_get_groups", "pm_match", "pm_match_per_task", "pm_newest_task", "pm_tick",
    "pm_last_stats", "pm_lookup_task_for_worker", "pm_device_task_column", "pm_host_parse_requirements", "pm_host_model_matches",
    "pm_host_build_model_table", "pm_host_config_order", "pm_host_group_vars", "pm_host_volume_vars",
    "pm_host_upload_name_vars", "pm_host_last_file_idx", "pm_abi_version",
    "pm_append_workers", "pm_set_addr_ranks", "pm_tasks_insert_front", "pm_tasks_insert_front_ex", "pm_tasks_delete", "pm_set_stream", "pm_set_carve_workgroups", "pm_tick_many", "pm_dist_configure", "pm_dist_tick_begin", "pm_dist_carve_wait",
    "pm_dist_match_begin", "pm_dist_tick_end", "pm_match_per_task_device",
    "pm_dissolve_group_by_id", "pm_get_group_by_id", "pm_get_group_of_worker", "pm_host_to_lowercase",
]

_lib = None


class EngineError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"pm_engine error {code}: {msg}")
        self.code = code


def lib() -> C.CDLL:
    """Load libpm_engine.so (building it in-tree first if the sources are newer)."""
    global _lib
    if _lib is None:
        path = _build.LIB_PATH
        if _build.needs_build():
            path = _build.build()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m protocol_amd.build` (needs hipcc)")
        L = C.CDLL(path)
        vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
        L.pm_last_error.restype = C.c_char_p
        L.pm_abi_version.restype = u32
        L.pm_engine_config_default.argtypes = [C.POINTER(EngineConfig)]
        L.pm_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(vp)]
        L.pm_engine_destroy.argtypes = [vp]
        L.pm_engine_destroy.restype = None
        L.pm_set_configs.argtypes = [vp, vp, u32, vp, u32]
        L.pm_set_model_table.argtypes = [vp, vp, u32, u32]
        L.pm_set_enabled_mask.argtypes = [vp, u64]
        L.pm_upload_workers.argtypes = [vp, C.POINTER(WorkerSoa), u32]
        L.pm_update_workers.argtypes = [vp, vp, C.POINTER(WorkerSoa)]
        L.pm_upload_tasks.argtypes = [vp, C.POINTER(TaskSoa)]
        L.pm_on_worker_status.argtypes = [vp, u32, u32, u32]
        L.pm_on_worker_status_many.argtypes = [vp, vp, vp, vp, u32]
        L.pm_enable_group_events.argtypes = [vp, u32]
        L.pm_drain_group_events.argtypes = [vp, vp, u32, vp, u32, C.POINTER(u32), C.POINTER(u32)]
        L.pm_dissolve_group.argtypes = [vp, u32]
        L.pm_reset_groups.argtypes = [vp]
        L.pm_compat_masks.argtypes = [vp, vp]
        L.pm_form_groups.argtypes = [vp, C.POINTER(u32)]
        L.pm_merge_solo_groups.argtypes = [vp, C.POINTER(u32)]
        L.pm_get_groups.argtypes = [vp, vp, vp, u32, C.POINTER(u32), vp, u32, C.POINTER(u32)]
        L.pm_match.argtypes = [vp, vp, vp]
        L.pm_match_per_task.argtypes = [vp, vp, vp]
        L.pm_newest_task.argtypes = [vp, C.POINTER(u32)]
        L.pm_tick.argtypes = [vp, C.POINTER(Stats)]
        L.pm_last_stats.argtypes = [vp, C.POINTER(Stats)]
        L.pm_tick_many.argtypes = [C.POINTER(vp), u32, C.POINTER(Stats), u32]
        L.pm_lookup_task_for_worker.argtypes = [vp, u32, C.POINTER(Assignment)]
        L.pm_device_task_column.argtypes = [vp, C.POINTER(u64), C.POINTER(u32)]
        L.pm_append_workers.argtypes = [vp, C.POINTER(WorkerSoa), C.POINTER(u32)]
        L.pm_set_addr_ranks.argtypes = [vp, vp, u32]
        L.pm_tasks_insert_front.argtypes = [vp, C.POINTER(TaskSoa)]
        L.pm_tasks_delete.argtypes = [vp, vp, u32, C.POINTER(u32)]
        L.pm_set_stream.argtypes = [vp, vp]
        L.pm_dist_configure.argtypes = [vp, u32, u32, vp]
        L.pm_dist_tick_begin.argtypes = [vp]
        L.pm_dist_carve_wait.argtypes = [vp]
        L.pm_dissolve_group_by_id.argtypes = [vp, u64, C.POINTER(u32)]
        L.pm_get_group_by_id.argtypes = [vp, u64, vp, vp, u32, C.POINTER(u32)]
        L.pm_get_group_of_worker.argtypes = [vp, u32, vp, vp, u32, C.POINTER(u32)]
        L.pm_dist_match_begin.argtypes = [vp, C.POINTER(DistXfer)]
        L.pm_dist_tick_end.argtypes = [vp, C.POINTER(Stats)]
        L.pm_match_per_task_device.argtypes = [vp, C.POINTER(u64), C.POINTER(u64), C.POINTER(u32)]
        L.pm_host_parse_requirements.argtypes = [C.c_char_p, vp, vp, u32, C.c_char_p, C.c_size_t]
        L.pm_host_model_matches.argtypes = [C.c_char_p, C.c_char_p]
        L.pm_host_build_model_table.argtypes = [C.POINTER(C.c_char_p), u32, C.POINTER(C.c_char_p), u32, vp]
        L.pm_host_config_order.argtypes = [vp, u32, u64, vp, C.POINTER(u32)]
        sz = C.c_size_t
        L.pm_host_group_vars.argtypes = [C.c_char_p, C.POINTER(GroupVars), C.c_char_p, sz, C.POINTER(sz)]
        L.pm_host_volume_vars.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, sz, C.POINTER(sz)]
        L.pm_host_upload_name_vars.argtypes = [C.c_char_p, C.c_char_p, u32, u32, u64, C.c_char_p, sz, C.POINTER(sz)]
        L.pm_host_to_lowercase.argtypes = [C.c_char_p, C.c_char_p, sz, C.POINTER(sz)]
        L.pm_host_last_file_idx.argtypes = [C.c_char_p]
        L.pm_host_last_file_idx.restype = u32
        for name in EXPORTS:
            fn = getattr(L, name)
            if name not in ("pm_last_error", "pm_abi_version", "pm_engine_destroy", "pm_engine_config_default",
                            "pm_host_last_file_idx"):
                fn.restype = i32
        _lib = L
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise EngineError(rc, lib().pm_last_error().decode(errors="replace"))
    return rc


def _arr(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


TICK_MANY_THREADS = 1


def tick_many(engines, threads: bool = False) -> list:
    """pm_tick_many: one match per engine (pool), all of them in one call — the carves started before the first is
    waited for, from one host thread (threads=True: one host thread per engine inside the library, for comparison).
    Returns the engines' stats in order."""
    n = len(engines)
    hs = (C.c_void_p * n)(*[e._h for e in engines])
    st = (Stats * n)()
    check(lib().pm_tick_many(hs, n, st, TICK_MANY_THREADS if threads else 0))
    return [s.as_dict() for s in st]


class Engine:
    """Owns a pm_engine*.  Keeps nothing but the handle: all state lives behind the C ABI."""

    def __init__(self, *, device: int = 0, proximity=True, switching=True, prefer_larger=True,
                 chooser=CHOOSE_FIRST, chooser_seed=0, group_id_seed=1, debug_uncertain_every=0, sweep_variant=0,
                 carve_variant=0, time_proposer=False):
        L = lib()
        cfg = EngineConfig()
        L.pm_engine_config_default(C.byref(cfg))
        cfg.device = device
        cfg.proximity_enabled = int(proximity)
        cfg.switching_enabled = int(switching)
        cfg.prefer_larger_groups = int(prefer_larger)
        cfg.chooser = chooser
        cfg.chooser_seed = chooser_seed
        cfg.group_id_seed = group_id_seed
        cfg.debug_uncertain_every = debug_uncertain_every
        cfg.sweep_variant = sweep_variant
        cfg.carve_variant = carve_variant
        cfg.time_proposer = int(time_proposer)
        self._h = C.c_void_p()
        check(L.pm_engine_create(C.byref(cfg), C.byref(self._h)))
        self.W = 0
        self.T = 0

    def close(self):
        if getattr(self, "_h", None):
            lib().pm_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- tables
    def set_configs(self, cfg_rows: np.ndarray, alt_rows: np.ndarray):
        cfg_rows = _arr(cfg_rows, config_row_dt)
        alt_rows = _arr(alt_rows, alt_row_dt)
        check(lib().pm_set_configs(self._h, cfg_rows.ctypes.data, len(cfg_rows),
                                   alt_rows.ctypes.data if len(alt_rows) else None, len(alt_rows)))
        self.C = len(cfg_rows)

    def set_model_table(self, bits: np.ndarray, n_rows: int, n_classes: int):
        bits = _arr(bits, np.uint32)
        check(lib().pm_set_model_table(self._h, bits.ctypes.data if bits.size else None, n_rows, n_classes))

    def set_enabled_mask(self, mask: int):
        check(lib().pm_set_enabled_mask(self._h, mask & 0xFFFFFFFFFFFFFFFF))

    @staticmethod
    def _worker_soa(cols: dict):
        keep = {}
        soa = WorkerSoa()
        n = len(cols["flags"])
        soa.n = n
        for k, dt in (("flags", np.uint32), ("gpu_count", np.uint32), ("gpu_mem_mb", np.uint32),
                      ("gpu_model_class", np.uint32), ("cpu_cores", np.uint32), ("ram_mb", np.uint32),
                      ("storage_gb", np.uint32), ("price", np.uint32), ("addr_rank", np.uint32),
                      ("lat", np.float64), ("lon", np.float64)):
            v = cols.get(k)
            if v is None:
                setattr(soa, k, None)
            else:
                keep[k] = _arr(v, dt)
                assert len(keep[k]) == n, k
                setattr(soa, k, keep[k].ctypes.data)
        return soa, keep

    def upload_workers(self, cols: dict, keep_groups: bool = False):
        soa, keep = self._worker_soa(cols)
        check(lib().pm_upload_workers(self._h, C.byref(soa), int(keep_groups)))
        self.W = soa.n

    def update_workers(self, idx, cols: dict):
        idx = _arr(idx, np.uint32)
        soa, keep = self._worker_soa(cols)
        assert soa.n == len(idx)
        check(lib().pm_update_workers(self._h, idx.ctypes.data, C.byref(soa)))

    def append_workers(self, cols: dict) -> int:
        """-> index of the first appended row"""
        soa, keep = self._worker_soa(cols)
        first = C.c_uint32(0)
        check(lib().pm_append_workers(self._h, C.byref(soa), C.byref(first)))
        self.W += soa.n
        return first.value

    def set_addr_ranks(self, ranks):
        r = _arr(ranks, np.uint32)
        check(lib().pm_set_addr_ranks(self._h, r.ctypes.data if len(r) else None, len(r)))

    @staticmethod
    def _task_soa(topo_mask, created_at, uid):
        tm, ca = _arr(topo_mask, np.uint64), _arr(created_at, np.int64)
        soa = TaskSoa()
        soa.n = len(tm)
        soa.topo_mask = tm.ctypes.data if len(tm) else None
        soa.created_at = ca.ctypes.data if len(ca) else None
        u = None
        if uid is not None:
            u = _arr(uid, np.uint64)
            soa.uid = u.ctypes.data if len(u) else None
        return soa, (tm, ca, u)

    def upload_tasks(self, topo_mask, created_at, uid=None):
        soa, keep = self._task_soa(topo_mask, created_at, uid)
        check(lib().pm_upload_tasks(self._h, C.byref(soa)))
        self.T = soa.n

    def tasks_insert_front(self, topo_mask, created_at, uid=None, republish=False):
        """new tasks, newest first; they sort in front of the table (on_task_created).  republish: also run pm_match's
        pair sweep + claim + publish on the standing groups (pm_tasks_insert_front_ex), so that a group holding no task
        is served the new one before the next tick"""
        soa, keep = self._task_soa(topo_mask, created_at, uid)
        if republish:
            check(lib().pm_tasks_insert_front_ex(self._h, C.byref(soa), 1))
        else:
            check(lib().pm_tasks_insert_front(self._h, C.byref(soa)))
        self.T += soa.n

    def set_carve_workgroups(self, n: int):
        """row-making workgroups of the carve's launch (0 = by the size of the swarm): the share of the GPU this engine
        takes when several pools match on it at the same time"""
        check(lib().pm_set_carve_workgroups(self._h, int(n)))

    def tasks_delete(self, uids) -> int:
        u = _arr(uids, np.uint64)
        n = C.c_uint32(0)
        check(lib().pm_tasks_delete(self._h, u.ctypes.data if len(u) else None, len(u), C.byref(n)))
        self.T -= n.value
        return n.value

    # ---- events
    def on_worker_status(self, worker: int, flags_new: int, dead: bool):
        check(lib().pm_on_worker_status(self._h, worker, flags_new, int(dead)))

    def on_worker_status_many(self, workers, flags_new, dead=None):
        """pm_on_worker_status for a batch, in array order; `dead`: per-entry flags (None = nobody)."""
        w, f = _arr(workers, np.uint32), _arr(flags_new, np.uint32)
        assert len(w) == len(f)
        d = _arr(dead, np.uint32) if dead is not None else None
        assert d is None or len(d) == len(w)
        check(lib().pm_on_worker_status_many(self._h, w.ctypes.data if len(w) else None, f.ctypes.data if len(f) else None,
                                             d.ctypes.data if d is not None and len(d) else None, len(w)))

    def enable_group_events(self, on: bool = True):
        check(lib().pm_enable_group_events(self._h, int(on)))

    def drain_group_events(self):
        """[(kind, group id, config, members in BTreeSet order)] since the last drain (kind 1 = created, 2 = destroyed):
        the webhook feed, in the order the reference emits it"""
        ne, nm = C.c_uint32(0), C.c_uint32(0)
        rc = lib().pm_drain_group_events(self._h, None, 0, None, 0, C.byref(ne), C.byref(nm))
        if rc == 0:
            return []                                          # the log is empty
        ev = np.zeros(max(ne.value, 1), dtype=GROUP_EVENT)
        mem = np.zeros(max(nm.value, 1), dtype=np.uint32)
        check(lib().pm_drain_group_events(self._h, ev.ctypes.data, len(ev), mem.ctypes.data, len(mem), C.byref(ne),
                                          C.byref(nm)))
        return [(int(e["kind"]), int(e["group_id"]), int(e["config"]),
                 mem[int(e["member_begin"]):int(e["member_begin"]) + int(e["n_members"])].tolist()) for e in ev[:ne.value]]

    def dissolve_group(self, slot: int):
        check(lib().pm_dissolve_group(self._h, slot))

    def dissolve_group_by_id(self, group_id: int) -> bool:
        """dissolve_group(&group_id) (mod.rs:1002-1004): False = no group has this id (not an error)"""
        n = C.c_uint32(0)
        check(lib().pm_dissolve_group_by_id(self._h, int(group_id), C.byref(n)))
        return bool(n.value)

    def _one_group(self, call):
        g = np.zeros(1, dtype=group_dt)
        slot = C.c_uint32(0)
        members = np.zeros(64, dtype=np.uint32)
        rc = call(g.ctypes.data, members.ctypes.data, len(members), C.byref(slot))
        if rc == PM_ERANGE and slot.value != PM_NONE:            # a group of more than 64: its size is in the record
            members = np.zeros(int(g[0]["n_members"]), dtype=np.uint32)
            rc = call(g.ctypes.data, members.ctypes.data, len(members), C.byref(slot))
        check(rc)
        if slot.value == PM_NONE:
            return None
        n = int(g[0]["n_members"])
        return {"slot": slot.value, "id": int(g[0]["id"]), "config": int(g[0]["config"]), "task": int(g[0]["task"]),
                "members": members[:n].tolist()}

    def get_group_by_id(self, group_id: int):
        """get_group_by_id (mod.rs:1046-1055) -> None or {slot, id, config, task, members (BTreeSet order)}"""
        return self._one_group(lambda g, m, cap, s: lib().pm_get_group_by_id(self._h, int(group_id), g, m, cap, s))

    def get_group_of_worker(self, worker: int):
        """get_node_group (mod.rs:324-337) by row index -> None or the same record"""
        return self._one_group(lambda g, m, cap, s: lib().pm_get_group_of_worker(self._h, int(worker), g, m, cap, s))

    def reset_groups(self):
        check(lib().pm_reset_groups(self._h))

    # ---- phases
    def compat_masks(self) -> np.ndarray:
        out = np.zeros(self.W, dtype=np.uint64)
        check(lib().pm_compat_masks(self._h, out.ctypes.data if self.W else None))
        return out

    def form_groups(self) -> int:
        n = C.c_uint32(0)
        check(lib().pm_form_groups(self._h, C.byref(n)))
        return n.value

    def merge_solo_groups(self) -> int:
        n = C.c_uint32(0)
        check(lib().pm_merge_solo_groups(self._h, C.byref(n)))
        return n.value

    def get_groups(self):
        """-> (group_of_worker int32[W], groups structured array, members uint32[] in BTreeSet order)"""
        ng, nm = C.c_uint32(0), C.c_uint32(0)
        check(lib().pm_get_groups(self._h, None, None, 0, C.byref(ng), None, 0, C.byref(nm)))
        gow = np.zeros(self.W, dtype=np.int32)
        groups = np.zeros(ng.value, dtype=group_dt)
        members = np.zeros(nm.value, dtype=np.uint32)
        check(lib().pm_get_groups(self._h, gow.ctypes.data if self.W else None,
                                  groups.ctypes.data if ng.value else None, ng.value, C.byref(ng),
                                  members.ctypes.data if nm.value else None, nm.value, C.byref(nm)))
        return gow, groups, members

    def match(self):
        task = np.zeros(self.W, dtype=np.uint32)
        count = np.zeros(self.W, dtype=np.uint32)
        check(lib().pm_match(self._h, task.ctypes.data if self.W else None, count.ctypes.data if self.W else None))
        return task, count

    def match_per_task(self):
        best = np.zeros(self.T, dtype=np.uint32)
        count = np.zeros(self.T, dtype=np.uint32)
        check(lib().pm_match_per_task(self._h, best.ctypes.data if self.T else None,
                                      count.ctypes.data if self.T else None))
        return best, count

    def newest_task(self) -> int:
        t = C.c_uint32(0)
        check(lib().pm_newest_task(self._h, C.byref(t)))
        return t.value

    def tick(self) -> dict:
        s = Stats()
        check(lib().pm_tick(self._h, C.byref(s)))
        return s.as_dict()

    def last_stats(self) -> dict:
        s = Stats()
        check(lib().pm_last_stats(self._h, C.byref(s)))
        return s.as_dict()

    def device_task_column(self):
        """(device pointer, n) of the published per-worker task column — for device-side consumers."""
        p, n = C.c_uint64(0), C.c_uint32(0)
        check(lib().pm_device_task_column(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- multi-GPU: ownership + the stepwise tick (the exchanges are the caller's, protocol_amd/dist.py)
    def set_stream(self, hip_stream: int | None):
        check(lib().pm_set_stream(self._h, C.c_void_p(hip_stream or 0)))

    def dist_configure(self, rank: int, world: int, shard_of_worker=None):
        sh = None if shard_of_worker is None else _arr(shard_of_worker, np.uint8)
        if sh is not None:
            assert len(sh) == self.W
        check(lib().pm_dist_configure(self._h, rank, world, sh.ctypes.data if sh is not None and len(sh) else None))

    def dist_tick_begin(self):
        check(lib().pm_dist_tick_begin(self._h))

    def dist_carve_wait(self):
        check(lib().pm_dist_carve_wait(self._h))

    def dist_match_begin(self) -> DistXfer:
        x = DistXfer()
        check(lib().pm_dist_match_begin(self._h, C.byref(x)))
        return x

    def dist_tick_end(self) -> dict:
        s = Stats()
        check(lib().pm_dist_tick_end(self._h, C.byref(s)))
        return s.as_dict()

    def match_per_task_device(self):
        """-> (device pointer of best u32[T], device pointer of count u32[T], T)"""
        b, c, n = C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        check(lib().pm_match_per_task_device(self._h, C.byref(b), C.byref(c), C.byref(n)))
        return b.value, c.value, n.value

    def hbm_triad_gbs(self, n_doubles: int = 1 << 27, reps: int = 5) -> float:
        """measured HBM rate (stream triad over 3 x n_doubles f64) — a debug export, not part of the ABI header"""
        L = lib()
        L.pm_debug_hbm_triad.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
        L.pm_debug_hbm_triad.restype = C.c_int32
        out = C.c_double(0)
        check(L.pm_debug_hbm_triad(self._h, n_doubles, reps, C.byref(out)))
        return out.value

    def debug_mem_lists_above(self, n: int):
        """test hook (pm_internal.h): candidate lists longer than n slots take the all-in-HBM carve path; 0 = off"""
        L = lib()
        L.pm_debug_mem_lists_above.argtypes = [C.c_void_p, C.c_uint32]
        L.pm_debug_mem_lists_above.restype = C.c_int32
        check(L.pm_debug_mem_lists_above(self._h, n))

    def debug_stream_abort_after(self, n: int):
        """test hook (include/pm_engine_debug.h): the streaming carve gives its launch up (CARVE_STATE_ABORTED) once n
        steps are committed, and the engine continues on the batch pipeline; 0 = off"""
        L = lib()
        L.pm_debug_stream_abort_after.argtypes = [C.c_void_p, C.c_uint32]
        L.pm_debug_stream_abort_after.restype = C.c_int32
        check(L.pm_debug_stream_abort_after(self._h, n))

    def debug_merge_streamed(self) -> int:
        """merge configurations whose selections went through the streaming carve since the engine was created"""
        L = lib()
        L.pm_debug_merge_streamed.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.pm_debug_merge_streamed.restype = C.c_int32
        n = C.c_uint32(0)
        check(L.pm_debug_merge_streamed(self._h, C.byref(n)))
        return n.value

    def debug_delta_pushes(self) -> int:
        """times the device's group state went up as a delta instead of the whole list since the engine was created"""
        L = lib()
        L.pm_debug_delta_pushes.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.pm_debug_delta_pushes.restype = C.c_int32
        n = C.c_uint32(0)
        check(L.pm_debug_delta_pushes(self._h, C.byref(n)))
        return n.value

    def debug_row_networks(self, keys, sites, n_per_wave: int, slot_bits: int, ulps: int, upto: int):
        """test hook (include/pm_engine_debug.h): rows from keys[n_waves * n_per_wave] by the insertion and by the sorting
        networks, compared on the device -> (mismatch bits, rows that differ, the networks' rows as u64[n_waves, 64])"""
        import numpy as np
        L = lib()
        keys = np.ascontiguousarray(keys, dtype=np.uint64)
        sites = np.ascontiguousarray(sites, dtype=np.uint32)
        assert keys.size % n_per_wave == 0 and sites.size == 1 << slot_bits
        n_waves = keys.size // n_per_wave
        rows = np.zeros((n_waves, 64), dtype=np.uint64)
        mism = (C.c_uint32 * 2)()
        L.pm_debug_row_networks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64,
                                            C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]
        L.pm_debug_row_networks.restype = C.c_int32
        check(L.pm_debug_row_networks(self._h, keys.ctypes.data, sites.ctypes.data, n_waves, n_per_wave, slot_bits, ulps, upto,
                                      rows.ctypes.data, mism))
        return int(mism[0]), int(mism[1]), rows

    def debug_carve_counters(self) -> dict:
        """how the last carve went (pm_internal.h, pm_debug_carve_prof words 32..45): how its validation launches ended,
        and what the proposer's spatial index did"""
        L = lib()
        L.pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
        L.pm_debug_carve_prof.restype = C.c_int32
        out = (C.c_ulonglong * 56)()
        check(L.pm_debug_carve_prof(self._h, out, 56))
        w = [int(v) for v in out[32:56]]
        return {"why": w[:8], "batches": w[8], "void_launches": w[9], "pruned_batches": w[10], "prune_fallbacks": w[11],
                "cell_g": w[12], "n_indexed": w[13],
                # the streaming carve (carve_variant 0): did the last carve stream, seeds handed to the proposers, rows the
                # validator gave up waiting for, configurations that switched from the index walk to a candidate list,
                # configurations entered with a list, proposer workgroups, launches that gave up (-> batch pipeline),
                # steps that took the exact sweep
                "stream": w[14], "stream_tickets": w[15], "stream_timeouts": w[16], "stream_switches": w[17],
                "stream_listed": w[18], "stream_wgs": w[19], "stream_aborts": w[20], "slow_steps": w[21],
                "stream_pre_used": w[22], "stream_pre_lost": w[23]}

    def debug_prune_mode(self, mode: int):
        """test hook (pm_internal.h): 0 the proposer always sweeps the whole candidate list, 1 it walks the spatial index
        when that pays (default), 2 whenever the carve has an index, 3 = 2 with every seed through the fallback"""
        L = lib()
        L.pm_debug_prune_mode.argtypes = [C.c_void_p, C.c_uint32]
        L.pm_debug_prune_mode.restype = C.c_int32
        check(L.pm_debug_prune_mode(self._h, mode))

    def lookup(self, worker: int) -> Assignment:
        a = Assignment()
        check(lib().pm_lookup_task_for_worker(self._h, worker, C.byref(a)))
        return a
