"""The compiled host side (protocol_amd/plugin: GpuMatchPlugin + Scheduler in C++, libpm_plugin.so) behind the interface
tests/shim_replay.py gives the Python replay of the Rust shim — so the same scenario can drive either: against the
oracle on a GPU (tests/test_gpu_shim_replay.py), and the two against each other over tests/cpp/mock_engine.cpp
without one (tests/plugin_diff_driver.py).  ctypes over protocol_amd/plugin/pm_plugin_c.h.

A store row of the Swarm becomes an OrchestratorNode: the Options that are Some are the PM_W_* bits host.pack_workers
derives from the Swarm's *_some columns, the address string is Swarm.address_strings()'s, the p2p id is "p2p-<row>".  A
task is identified by a Uuid whose low 64 bits (as_u64_pair().1) are the uid the harness uses; its allowed_topologies
are the names of the configurations its mask has set — plus, for a mask without a bit (a task that names only
topologies no configuration has), one such name.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from protocol_amd import build as _build
from protocol_amd import engine as E
from protocol_amd import host

ALL = 0xFFFFFFFFFFFFFFFF


class PmxConfig(C.Structure):
    _fields_ = [("name", C.c_char_p), ("min_group_size", C.c_uint32), ("max_group_size", C.c_uint32),
                ("compute_requirements", C.c_char_p)]


class PmxNode(C.Structure):
    _fields_ = [("address", C.c_char_p), ("status", C.c_uint32), ("has", C.c_uint32), ("p2p_id", C.c_char_p),
                ("gpu_count", C.c_uint32), ("gpu_memory_mb", C.c_uint32), ("gpu_model", C.c_char_p),
                ("cpu_cores", C.c_uint32), ("ram_mb", C.c_uint32), ("storage_gb", C.c_uint32),
                ("latitude", C.c_double), ("longitude", C.c_double)]


class PmxTask(C.Structure):
    _fields_ = [("id", C.c_char_p), ("name", C.c_char_p), ("created_at", C.c_int64), ("n_topologies", C.c_int32),
                ("topologies", C.POINTER(C.c_char_p)), ("n_env", C.c_uint32), ("env_keys", C.POINTER(C.c_char_p)),
                ("env_values", C.POINTER(C.c_char_p)), ("n_cmd", C.c_int32), ("cmd", C.POINTER(C.c_char_p)),
                ("n_mounts", C.c_int32), ("mount_host", C.POINTER(C.c_char_p)), ("mount_container", C.POINTER(C.c_char_p))]


PMX_HEALTHY, PMX_UNHEALTHY, PMX_DEAD = 2, 3, 4
_plib = None


def plugin_lib(path: str | None = None) -> C.CDLL:
    """libpm_plugin.so (its pm_* symbols bind to the engine library already loaded into the process — the real one
    through its DT_NEEDED entry, or whatever E.lib() was pointed at before: load that first, RTLD_GLOBAL)"""
    global _plib
    if _plib is None:
        L = C.CDLL(path or _build.build_plugin())
        vp, u32 = C.c_void_p, C.c_uint32
        L.pmx_last_error.restype = C.c_char_p
        L.pmx_create.argtypes = [C.POINTER(PmxConfig), u32, C.c_int32, C.POINTER(vp)]
        L.pmx_destroy.argtypes = [vp]
        L.pmx_destroy.restype = None
        L.pmx_engine.argtypes = [vp]
        L.pmx_engine.restype = vp
        L.pmx_set_upload_count.argtypes = [vp, C.c_uint64]
        L.pmx_set_upload_count.restype = None
        L.pmx_set_republish_on_insert.argtypes = [vp, u32]
        L.pmx_set_republish_on_insert.restype = None
        L.pmx_sync_nodes.argtypes = [vp, C.POINTER(PmxNode), u32]
        L.pmx_sync_tasks.argtypes = [vp, C.POINTER(PmxTask), u32]
        L.pmx_on_task_created.argtypes = [vp, C.POINTER(PmxTask)]
        L.pmx_on_task_deleted.argtypes = [vp, C.c_char_p]
        L.pmx_handle_status_change.argtypes = [vp, C.c_char_p, u32]
        L.pmx_tick.argtypes = [vp, C.POINTER(E.Stats)]
        L.pmx_row_of.argtypes = [vp, C.c_char_p, C.POINTER(u32)]
        L.pmx_known_nodes.argtypes = [vp]
        L.pmx_known_nodes.restype = u32
        L.pmx_store_loads.argtypes = [vp]
        L.pmx_store_loads.restype = u32
        sz = C.c_size_t
        L.pmx_get_task_for_node.argtypes = [vp, C.c_char_p, C.c_int64, C.c_char_p, sz, C.POINTER(sz)]
        L.pmx_take_webhooks.argtypes = [vp, C.c_char_p, sz, C.POINTER(sz)]
        text = [C.c_char_p, sz, C.POINTER(sz)]
        L.pmx_set_clock.argtypes = [vp, C.c_int64]
        L.pmx_set_clock.restype = None
        L.pmx_get_all_groups.argtypes = [vp] + text
        L.pmx_get_group_by_id.argtypes = [vp, C.c_char_p] + text
        L.pmx_get_node_group.argtypes = [vp, C.c_char_p] + text
        L.pmx_get_node_groups_batch.argtypes = [vp, C.POINTER(C.c_char_p), u32] + text
        L.pmx_get_all_node_group_mappings.argtypes = [vp] + text
        L.pmx_get_configurations.argtypes = [vp, u32] + text
        L.pmx_dissolve_group.argtypes = [vp, C.c_char_p]
        L.pmx_upload_file_name.argtypes = [vp, C.c_char_p, C.c_char_p] + text
        _plib = L
    return _plib


_dlib = None


class PmxCommError(RuntimeError):
    pass


def dist_lib() -> C.CDLL:
    """libpm_plugin_dist.so: GpuMatchPlugin::tick_dist's communicators (protocol_amd/plugin/pm_plugin_dist_c.h)"""
    global _dlib
    if _dlib is None:
        plugin_lib()
        L = C.CDLL(_build.build_plugin_dist())
        vp, u32 = C.c_void_p, C.c_uint32
        L.pmx_last_error_dist.restype = C.c_char_p
        L.pmx_rccl_create.argtypes = [u32, u32, C.c_int32, C.c_char_p, C.POINTER(vp)]
        L.pmx_local_world_create.argtypes = [u32, C.c_int32, C.POINTER(vp)]
        L.pmx_comm_destroy.argtypes = [vp]
        L.pmx_comm_destroy.restype = None
        L.pmx_tick_dist.argtypes = [vp, vp, C.POINTER(E.Stats)]
        L.pmx_rccl_self_test.argtypes = [C.c_int32, u32, C.c_char_p]
        _dlib = L
    return _dlib


def _check_dist(rc: int):
    if rc != 0:
        raise PmxCommError("pm_plugin_dist: " + dist_lib().pmx_last_error_dist().decode(errors="replace"))


def local_world(n: int, device: int = 0) -> list:
    """n communicator handles of ranks that share this process (one thread each)"""
    out = (C.c_void_p * n)()
    _check_dist(dist_lib().pmx_local_world_create(n, device, out))
    return [C.c_void_p(out[i]) for i in range(n)]


def rccl_comm(rank: int, world: int, device: int, id_file: str):
    out = C.c_void_p()
    _check_dist(dist_lib().pmx_rccl_create(rank, world, device, id_file.encode(), C.byref(out)))
    return out


def comm_destroy(comm):
    dist_lib().pmx_comm_destroy(comm)


def _check(rc: int):
    if rc != 0:
        raise RuntimeError("pm_plugin: " + plugin_lib().pmx_last_error().decode(errors="replace"))


def _text(call) -> str:
    need = C.c_size_t(0)
    rc = call(None, 0, C.byref(need))
    if rc not in (0, -2):
        _check(rc)
    buf = C.create_string_buffer(max(need.value, 1))
    _check(call(buf, len(buf), C.byref(need)))
    return buf.value.decode()


def _unesc(s: str) -> str:
    return s.replace("\\t", "\t").replace("\\n", "\n").replace("\\\\", "\\")


def uuid_of(uid: int) -> str:
    """a Uuid text whose as_u64_pair().1 is `uid`"""
    return "00000000-0000-4000-%04x-%012x" % ((uid >> 48) & 0xFFFF, uid & 0xFFFFFFFFFFFF)


def uid_of(uuid: str) -> int:
    return int(uuid.replace("-", "")[-16:], 16)


class _BorrowedEngine(E.Engine):
    """E.Engine over the plugin's own pm_engine* (get_groups & co. for the parity checks); never destroys it"""

    def __init__(self, handle):  # noqa: super().__init__ would create an engine
        self._h = C.c_void_p(handle)
        self.W = 0
        self.T = 0

    def close(self):
        self._h = None


class PluginCxx:
    """the interface of tests/shim_replay.ShimReplay over the C++ GpuMatchPlugin (+ its Scheduler: filter_tasks goes
    through Scheduler::get_task_for_node)"""

    def __init__(self, sw, device: int = 0):
        self.sw = sw
        self.L = plugin_lib()
        self.config_names = [c[0] for c in sw.configs]
        cfgs = (PmxConfig * len(sw.configs))()
        for i, (name, mn, mx, req) in enumerate(sw.configs):
            cfgs[i] = PmxConfig(name.encode(), mn, mx, None if req is None else req.encode())
        self._p = C.c_void_p()
        _check(self.L.pmx_create(cfgs, len(sw.configs), device, C.byref(self._p)))
        self.eng = _BorrowedEngine(self.L.pmx_engine(self._p))
        self.packed_all = host.pack_workers(sw)
        self.addr = [s.encode() for s in sw.address_strings()]
        self.node_of_addr = {s.decode(): i for i, s in enumerate(self.addr)}
        self.events: list = []
        self.tasks: list = []          # uids in the store's order (what the harness indexes)
        self.mid_observer = None       # (ShimReplay's hook between the halves of an observer: the C++ holds a real lock there)

    # ---- conversions
    def _node(self, node: int, healthy: bool) -> PmxNode:
        p = self.packed_all
        flags = int(p["flags"][node]) & ~E.W_HEALTHY
        model = self.sw.model_names[int(p["gpu_model_class"][node])].encode() if flags & E.W_GPU_MODEL else None
        return PmxNode(self.addr[node], PMX_HEALTHY if healthy else PMX_UNHEALTHY, flags, b"p2p-%d" % node,
                       int(p["gpu_count"][node]), int(p["gpu_mem_mb"][node]), model, int(p["cpu_cores"][node]),
                       int(p["ram_mb"][node]), int(p["storage_gb"][node]), float(p["lat"][node]), float(p["lon"][node]))

    def _task(self, mask: int, created: int, uid: int, keep: list, env=None, cmd=None, mounts=None) -> PmxTask:
        mask = int(mask) & ALL
        if mask == ALL:
            n_topo, names = -1, []
        else:
            names = [self.config_names[c].encode() for c in range(len(self.config_names)) if (mask >> c) & 1] or [b"no-such-topology"]
            n_topo = len(names)
        arr = (C.c_char_p * max(len(names), 1))(*names)
        ident = uuid_of(int(uid)).encode()
        keep += [arr, ident]
        def strings(items):
            a = (C.c_char_p * max(len(items), 1))(*[x.encode() for x in items])
            keep.append(a)
            return a
        env = env or {}
        return PmxTask(ident, b"task", int(created), n_topo, arr, len(env), strings(list(env.keys())), strings(list(env.values())),
                       -1 if cmd is None else len(cmd), strings(cmd or []), -1 if mounts is None else len(mounts),
                       strings([m[0] for m in (mounts or [])]), strings([m[1] for m in (mounts or [])]))

    def _take_webhooks(self):
        for line in _text(lambda o, c, n: self.L.pmx_take_webhooks(self._p, o, c, n)).splitlines():
            f = line.split("\t")
            kind = E.GROUP_CREATED if f[0] == "created" else E.GROUP_DESTROYED
            members = []
            for a in f[3:]:
                row = C.c_uint32(0)
                _check(self.L.pmx_row_of(self._p, a.encode(), C.byref(row)))
                members.append(row.value)
            self.events.append((kind, int(f[1], 16), self.config_names.index(f[2]), members))

    # ---- the plugin surface, as ShimReplay spells it
    @property
    def rows(self):
        return range(self.L.pmx_known_nodes(self._p))

    def sync_nodes(self, snapshot, healthy):
        nodes = (PmxNode * max(len(snapshot), 1))()
        for k, node in enumerate(snapshot):
            nodes[k] = self._node(int(node), int(node) in healthy)
        _check(self.L.pmx_sync_nodes(self._p, nodes, len(snapshot)))
        self._take_webhooks()

    def sync_tasks(self, masks, created, uid, enabled=None, env=None, cmd=None, mounts=None):
        """env / cmd / mounts: the same templates on every task ({key: value}, [arg], [(host, container)])"""
        keep: list = []
        tasks = (PmxTask * max(len(uid), 1))()
        for i in range(len(uid)):
            tasks[i] = self._task(masks[i], created[i], uid[i], keep, env, cmd, mounts)
        _check(self.L.pmx_sync_tasks(self._p, tasks, len(uid)))
        self.tasks = [int(u) for u in uid]

    def on_task_created(self, mask, created, uid, enabled=None):
        keep: list = []
        t = self._task(mask, created, uid, keep)
        _check(self.L.pmx_on_task_created(self._p, C.byref(t)))
        self.tasks.insert(0, int(uid))

    def on_task_deleted(self, uid, enabled=None):
        _check(self.L.pmx_on_task_deleted(self._p, uuid_of(int(uid)).encode()))
        self.tasks.remove(int(uid))
        self._take_webhooks()

    def handle_status_change(self, node: int, healthy: bool, dead: bool):
        _check(self.L.pmx_handle_status_change(self._p, self.addr[int(node)], PMX_HEALTHY if healthy else (PMX_DEAD if dead else PMX_UNHEALTHY)))
        self._take_webhooks()

    def tick(self):
        s = E.Stats()
        _check(self.L.pmx_tick(self._p, C.byref(s)))
        self._take_webhooks()
        return s.as_dict()

    def tick_dist(self, comm):
        """GpuMatchPlugin::tick_dist(comm): every rank calls it at the same point (its own thread or process)"""
        s = E.Stats()
        _check_dist(dist_lib().pmx_tick_dist(self._p, comm, C.byref(s)))
        self._take_webhooks()
        return s.as_dict()

    def task_for_node(self, node: int, now: int = 0):
        """Scheduler::get_task_for_node -> None or {"id", "uid", "name", "env", "cmd", "mounts"}"""
        text = _text(lambda o, c, n: self.L.pmx_get_task_for_node(self._p, self.addr[int(node)], now, o, c, n))
        if not text:
            return None
        out = {"env": {}, "cmd": [], "mounts": []}
        for line in text.splitlines():
            f = [_unesc(x) for x in line.split("\t")]
            if f[0] in ("id", "name"):
                out[f[0]] = f[1]
            elif f[0] == "env":
                out["env"][f[1]] = f[2]
            elif f[0] == "cmd":
                out["cmd"].append(f[1])
            elif f[0] == "mount":
                out["mounts"].append((f[1], f[2]))
        out["uid"] = uid_of(out["id"])
        return out

    def filter_tasks(self, *args):
        """filter_tasks(node) -> uid or None; filter_tasks(tasks, node) -> [uid] or [] (ShimReplay's two forms)"""
        t = self.task_for_node(args[-1])
        uid = None if t is None else t["uid"]
        if len(args) == 2:
            return [] if uid is None else [uid]
        return uid

    # ---- the read surface (what the API routes call on the plugin)
    @staticmethod
    def _group(fields):
        """-> {"id", "config", "created_at", "nodes"} from the fields of a group line"""
        return {"id": fields[0], "config": fields[1], "created_at": int(fields[2]), "nodes": fields[3:]}

    def set_clock(self, now_ms: int):
        self.L.pmx_set_clock(self._p, int(now_ms))

    def set_upload_count(self, n: int):
        self.L.pmx_set_upload_count(self._p, int(n))

    def get_all_groups(self):
        text = _text(lambda o, c, n: self.L.pmx_get_all_groups(self._p, o, c, n))
        return [self._group([_unesc(x) for x in line.split("\t")]) for line in text.splitlines()]

    def get_group_by_id(self, group_id: str):
        text = _text(lambda o, c, n: self.L.pmx_get_group_by_id(self._p, group_id.encode(), o, c, n))
        return self._group([_unesc(x) for x in text.splitlines()[0].split("\t")]) if text else None

    def get_node_group(self, address: str):
        """-> None or (get_idx_in_group, group)"""
        text = _text(lambda o, c, n: self.L.pmx_get_node_group(self._p, address.encode(), o, c, n))
        if not text:
            return None
        f = [_unesc(x) for x in text.splitlines()[0].split("\t")]
        return int(f[0]), self._group(f[1:])

    def get_node_groups_batch(self, addresses):
        arr = (C.c_char_p * max(len(addresses), 1))(*[a.encode() for a in addresses])
        text = _text(lambda o, c, n: self.L.pmx_get_node_groups_batch(self._p, arr, len(addresses), o, c, n))
        out = {}
        for line in text.splitlines():
            f = [_unesc(x) for x in line.split("\t")]
            out[f[0]] = None if f[1:] == ["-"] else self._group(f[1:])
        return out

    def get_all_node_group_mappings(self):
        text = _text(lambda o, c, n: self.L.pmx_get_all_node_group_mappings(self._p, o, c, n))
        return dict(tuple(_unesc(x) for x in line.split("\t")) for line in text.splitlines())

    def get_configurations(self, available_only: bool):
        text = _text(lambda o, c, n: self.L.pmx_get_configurations(self._p, int(available_only), o, c, n))
        out = []
        for line in text.splitlines():
            name, mn, mx, req = [_unesc(x) for x in line.split("\t")]
            out.append((name, int(mn), int(mx), None if req == "-" else req))
        return out

    def dissolve_group(self, group_id: str):
        _check(self.L.pmx_dissolve_group(self._p, group_id.encode()))
        self._take_webhooks()

    def upload_file_name(self, file_name: str, address: str):
        """-> (the rendered name, the group the route keys its upload counter by)"""
        text = _text(lambda o, c, n: self.L.pmx_upload_file_name(self._p, file_name.encode(), address.encode(), o, c, n))
        name, key = text.split("\n")[:2]
        return _unesc(name), _unesc(key)

    @property
    def store_loads(self) -> int:
        return self.L.pmx_store_loads(self._p)

    def set_republish_on_insert(self, on: bool):
        self.L.pmx_set_republish_on_insert(self._p, int(on))

    def close(self):
        if self._p:
            self.L.pmx_destroy(self._p)
            self._p = None
