"""ctypes/numpy binding for oracle/libpm_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(the product package protocol_amd never does).  The numpy structured dtypes below mirror the C
structs of oracle/pm_oracle.h field for field (align=True == the C ABI layout).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpm_oracle.so")

MAX_ALTS, MODEL_LEN, NAME_LEN, ADDR_LEN, MAX_TOPO = 16, 64, 32, 48, 4

# flag bits (pm_oracle.h)
G_COUNT, G_MODEL, G_MEM, G_MEM_MIN, G_MEM_MAX, G_TOT_MIN, G_TOT_MAX = (1 << i for i in range(7))
R_CPU, R_CPU_CORES, R_RAM, R_STORAGE = (1 << i for i in range(4))
S_GPU, S_G_COUNT, S_G_MODEL, S_G_MEM, S_CPU, S_CPU_CORES, S_RAM, S_STORAGE = (1 << i for i in range(8))
(ST_DISCOVERED, ST_WAITING, ST_HEALTHY, ST_UNHEALTHY, ST_DEAD, ST_EJECTED, ST_BANNED,
 ST_LOWBALANCE) = range(8)
CHOOSE_FIRST, CHOOSE_SEEDED = 0, 1

gpu_req_dt = np.dtype([("flags", "<u4"), ("count", "<u4"), ("memory_mb", "<u4"),
                       ("memory_mb_min", "<u4"), ("memory_mb_max", "<u4"),
                       ("total_memory_min", "<u4"), ("total_memory_max", "<u4"),
                       ("model", f"S{MODEL_LEN}")], align=True)
req_dt = np.dtype([("flags", "<u4"), ("cpu_cores", "<u4"), ("ram_mb", "<u4"), ("storage_gb", "<u4"),
                   ("n_gpu", "<u4"), ("gpu", gpu_req_dt, (MAX_ALTS,))], align=True)
specs_dt = np.dtype([("flags", "<u4"), ("gpu_count", "<u4"), ("gpu_memory_mb", "<u4"),
                     ("cpu_cores", "<u4"), ("ram_mb", "<u4"), ("storage_gb", "<u4"),
                     ("gpu_model", f"S{MODEL_LEN}")], align=True)
node_dt = np.dtype([("address", f"S{ADDR_LEN}"), ("status", "<u4"), ("has_p2p", "<u4"),
                    ("has_specs", "<u4"), ("has_location", "<u4"), ("specs", specs_dt),
                    ("latitude", "<f8"), ("longitude", "<f8")], align=True)
config_dt = np.dtype([("name", f"S{NAME_LEN}"), ("min_group_size", "<u8"), ("max_group_size", "<u8"),
                      ("has_requirements", "<u4"), ("_pad", "<u4"), ("req", req_dt)], align=True)
task_dt = np.dtype([("created_at", "<i8"), ("restricted", "<u4"), ("n_topologies", "<u4"),
                    ("topologies", f"S{NAME_LEN}", (MAX_TOPO,))], align=True)
policy_dt = np.dtype([("proximity_enabled", "<u4"), ("switching_enabled", "<u4"),
                      ("prefer_larger_groups", "<u4"), ("chooser", "<u4"), ("chooser_seed", "<u8"),
                      ("group_id_seed", "<u8"), ("reference_shaped", "<u4"), ("_pad", "<u4")], align=True)


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile).  Building the checker is not using it."""
    src = os.path.join(HERE, "pm_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(HERE, "pm_oracle.h"))):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B", "libpm_oracle.so"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        vp, sz, u32p = C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)
        L.orc_parse_requirements.argtypes = [C.c_char_p, vp, C.c_char_p, sz]
        L.orc_parse_requirements.restype = C.c_int
        L.orc_meets.argtypes = [vp, vp]
        L.orc_meets.restype = C.c_int
        L.orc_model_matches.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_model_matches.restype = C.c_int
        L.orc_to_lowercase_str.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_to_lowercase_str.restype = None
        L.orc_is_node_compatible_with_config.argtypes = [vp, vp]
        L.orc_is_node_compatible_with_config.restype = C.c_int
        L.orc_calculate_distance.argtypes = [C.c_double] * 4
        L.orc_calculate_distance.restype = C.c_double
        L.orc_distance_column.argtypes = [C.c_double, C.c_double, vp, vp, sz, vp]
        L.orc_distance_column.restype = None
        L.orc_sort_configs.argtypes = [vp, sz, vp]
        L.orc_sort_configs.restype = C.c_int
        L.orc_available_configs.argtypes = [vp, vp, sz, vp, vp]
        L.orc_available_configs.restype = sz
        L.orc_newest_task.argtypes = [vp, sz]
        L.orc_newest_task.restype = C.c_int64
        L.orc_task_applicable.argtypes = [vp, C.c_char_p]
        L.orc_task_applicable.restype = C.c_int
        L.orc_state_new.argtypes = [vp, sz, vp, sz, vp]
        L.orc_state_new.restype = vp
        L.orc_state_free.argtypes = [vp]
        L.orc_state_set_enabled.argtypes = [vp, vp]
        L.orc_state_set_tasks.argtypes = [vp, vp, sz]
        L.orc_state_set_node_status.argtypes = [vp, vp, sz, C.c_uint32]
        L.orc_state_remap_tasks.argtypes = [vp, vp, sz]
        L.orc_try_form_new_groups.argtypes = [vp]
        L.orc_try_form_new_groups.restype = sz
        L.orc_try_merge_solo_groups.argtypes = [vp]
        L.orc_try_merge_solo_groups.restype = sz
        L.orc_dissolve_group.argtypes = [vp, C.c_uint32]
        L.orc_filter_tasks_node_groups.argtypes = [vp, sz, u32p, u32p, u32p]
        L.orc_filter_tasks_node_groups.restype = C.c_int64
        L.orc_get_task_for_node.argtypes = [vp, sz, C.c_int]
        L.orc_get_task_for_node.restype = C.c_int64
        L.orc_n_groups.argtypes = [vp]
        L.orc_n_groups.restype = sz
        L.orc_group_slots.argtypes = [vp]
        L.orc_group_slots.restype = sz
        L.orc_node_to_group.argtypes = [vp]
        L.orc_node_to_group.restype = C.POINTER(C.c_int32)
        L.orc_group_info.argtypes = [vp, C.c_uint32, C.POINTER(C.c_uint64), u32p, u32p, vp, sz,
                                     C.POINTER(C.c_int64)]
        L.orc_group_info.restype = C.c_int
        L.orc_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_events.argtypes = [vp, vp, C.c_size_t, vp, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_events.restype = C.c_size_t
        L.orc_events_clear.argtypes = [vp]
        L.orc_events_clear.restype = None
        L.orc_group_vars.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_group_vars.restype = C.c_void_p
        L.orc_volume_vars.argtypes = [C.c_char_p, C.c_char_p]
        L.orc_volume_vars.restype = C.c_void_p
        L.orc_upload_name_vars.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_upload_name_vars.restype = C.c_void_p
        L.orc_last_file_idx.argtypes = [C.c_char_p]
        L.orc_last_file_idx.restype = C.c_uint32
        L.orc_free_string.argtypes = [C.c_void_p]
        L.orc_free_string.restype = None
        L.orc_splitmix64.argtypes = [C.POINTER(C.c_uint64)]
        L.orc_splitmix64.restype = C.c_uint64
        L.orc_compat_masks.argtypes = [vp, sz, vp, sz, vp]
        L.orc_pair_sweep_per_worker.argtypes = [vp, sz, vp, vp, sz, vp, vp]
        L.orc_pair_sweep_per_worker_mt.argtypes = [vp, sz, vp, vp, sz, vp, vp, C.c_uint32]
        L.orc_pair_sweep_per_worker_mt.restype = C.c_int
        _lib = L
    return _lib


GROUP_EVENT = np.dtype([("group_id", np.uint64), ("kind", np.uint32), ("config", np.uint32),
                        ("member_begin", np.uint32), ("n_members", np.uint32)])


def _p(a: np.ndarray) -> int:
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


# ------------------------------------------------------------------ pure functions

def parse_requirements(s: str):
    """-> (code, req_row(np.void) or None, err). code 0 ok / 1 Err / 2 panic / 3 oracle capacity."""
    out = np.zeros(1, dtype=req_dt)
    err = C.create_string_buffer(128)
    code = lib().orc_parse_requirements(s.encode(), _p(out), err, 128)
    return code, (out[0] if code == 0 else None), err.value.decode()


def make_specs(gpu_count=None, gpu_model=None, gpu_mem=None, cpu_cores=None, ram=None, storage=None,
               *, has_gpu=None, has_cpu=None) -> np.ndarray:
    """Mirror of the reference test helper create_compute_specs (shared/src/models/node.rs:625-657):
    gpu is Some iff any gpu field is given; cpu is Some iff cores is given (overridable)."""
    s = np.zeros(1, dtype=specs_dt)
    f = 0
    if has_gpu is None:
        has_gpu = gpu_count is not None or gpu_model is not None or gpu_mem is not None
    if has_cpu is None:
        has_cpu = cpu_cores is not None
    if has_gpu:
        f |= S_GPU
        if gpu_count is not None:
            f |= S_G_COUNT
            s["gpu_count"] = gpu_count
        if gpu_model is not None:
            f |= S_G_MODEL
            s["gpu_model"] = gpu_model.encode()
        if gpu_mem is not None:
            f |= S_G_MEM
            s["gpu_memory_mb"] = gpu_mem
    if has_cpu:
        f |= S_CPU
        if cpu_cores is not None:
            f |= S_CPU_CORES
            s["cpu_cores"] = cpu_cores
    if ram is not None:
        f |= S_RAM
        s["ram_mb"] = ram
    if storage is not None:
        f |= S_STORAGE
        s["storage_gb"] = storage
    s["flags"] = f
    return s


def meets(specs: np.ndarray, req) -> bool:
    r = np.zeros(1, dtype=req_dt)
    r[0] = req
    return bool(lib().orc_meets(_p(specs), _p(r)))


def model_matches(spec_model: str, req_model: str) -> bool:
    return bool(lib().orc_model_matches(spec_model.encode(), req_model.encode()))


def to_lowercase(text: str) -> str:
    """str::to_lowercase as the oracle restates it"""
    raw = text.encode()
    buf = C.create_string_buffer(4 * len(raw) + 8)
    lib().orc_to_lowercase_str(raw, buf, len(buf))
    return buf.value.decode()


def calculate_distance(lat1, lon1, lat2, lon2) -> float:
    return lib().orc_calculate_distance(lat1, lon1, lat2, lon2)


def distance_column(lat0: float, lon0: float, lat: np.ndarray, lon: np.ndarray) -> np.ndarray:
    lat = np.ascontiguousarray(lat, dtype=np.float64)
    lon = np.ascontiguousarray(lon, dtype=np.float64)
    out = np.zeros(len(lat), dtype=np.float64)
    lib().orc_distance_column(lat0, lon0, _p(lat), _p(lon), len(lat), _p(out))
    return out


def make_config(name: str, min_size: int, max_size: int, requirements: str | None) -> np.ndarray:
    c = np.zeros(1, dtype=config_dt)
    c["name"] = name.encode()
    c["min_group_size"] = min_size
    c["max_group_size"] = max_size
    if requirements is not None:
        code, row, err = parse_requirements(requirements)
        if code != 0:
            raise ValueError(f"requirements {requirements!r}: {err}")
        c["has_requirements"] = 1
        c["req"] = row
    return c


def sort_configs(cfgs: np.ndarray):
    order = np.zeros(max(len(cfgs), 1), dtype=np.uint32)
    code = lib().orc_sort_configs(_p(cfgs), len(cfgs), _p(order))
    return code, order[:len(cfgs)]


def newest_task(tasks: np.ndarray) -> int:
    return int(lib().orc_newest_task(_p(tasks), len(tasks)))


def make_task(created_at: int, topologies=None) -> np.ndarray:
    """topologies None => no allowed_topologies key (unrestricted); list => restricted."""
    t = np.zeros(1, dtype=task_dt)
    t["created_at"] = created_at
    if topologies is not None:
        assert len(topologies) <= MAX_TOPO
        t["restricted"] = 1
        t["n_topologies"] = len(topologies)
        for i, name in enumerate(topologies):
            t["topologies"][0][i] = name.encode()
    return t


def compat_masks(nodes: np.ndarray, cfgs: np.ndarray) -> np.ndarray:
    out = np.zeros(len(nodes), dtype=np.uint64)
    lib().orc_compat_masks(_p(nodes), len(nodes), _p(cfgs), len(cfgs), _p(out))
    return out


def pair_sweep_per_worker(tasks: np.ndarray, cfgs: np.ndarray, cfg_of_node: np.ndarray, threads: int = 1):
    """threads > 1: the nodes are split over POSIX threads (independent heartbeats)"""
    cfg_of_node = np.ascontiguousarray(cfg_of_node, dtype=np.int32)
    first = np.zeros(len(cfg_of_node), dtype=np.uint32)
    count = np.zeros(len(cfg_of_node), dtype=np.uint32)
    if threads > 1:
        lib().orc_pair_sweep_per_worker_mt(_p(tasks), len(tasks), _p(cfgs), _p(cfg_of_node), len(cfg_of_node),
                                           _p(first), _p(count), threads)
    else:
        lib().orc_pair_sweep_per_worker(_p(tasks), len(tasks), _p(cfgs), _p(cfg_of_node), len(cfg_of_node),
                                        _p(first), _p(count))
    return first, count


def splitmix64_stream(seed: int, n: int) -> np.ndarray:
    s = C.c_uint64(seed)
    return np.array([lib().orc_splitmix64(C.byref(s)) for _ in range(n)], dtype=np.uint64)


# ------------------------------------------------------------------ swarm state

class State:
    """Owns an orc_state; keeps the borrowed numpy tables alive."""

    def __init__(self, nodes: np.ndarray, cfgs: np.ndarray, *, proximity=True, switching=True,
                 prefer_larger=True, chooser=CHOOSE_FIRST, chooser_seed=0, group_id_seed=1,
                 reference_shaped=True, enabled=None, tasks: np.ndarray | None = None):
        self.nodes = np.ascontiguousarray(nodes)
        self.cfgs = np.ascontiguousarray(cfgs)
        pol = np.zeros(1, dtype=policy_dt)
        pol["proximity_enabled"] = int(proximity)
        pol["switching_enabled"] = int(switching)
        pol["prefer_larger_groups"] = int(prefer_larger)
        pol["chooser"] = chooser
        pol["chooser_seed"] = chooser_seed
        pol["group_id_seed"] = group_id_seed
        pol["reference_shaped"] = int(reference_shaped)
        self._h = lib().orc_state_new(_p(self.nodes), len(self.nodes), _p(self.cfgs), len(self.cfgs), _p(pol))
        if not self._h:
            raise ValueError("reference constructor would panic (duplicate names or max<min)")
        self.tasks = None
        self.set_enabled(np.ones(len(cfgs), dtype=np.uint8) if enabled is None else enabled)
        if tasks is not None:
            self.set_tasks(tasks)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_state_free(self._h)
            self._h = None

    def set_enabled(self, enabled):
        self.enabled = np.ascontiguousarray(enabled, dtype=np.uint8)
        lib().orc_state_set_enabled(self._h, _p(self.enabled))

    def set_tasks(self, tasks: np.ndarray):
        self.tasks = np.ascontiguousarray(tasks)
        lib().orc_state_set_tasks(self._h, _p(self.tasks), len(self.tasks))

    def remap_tasks(self, old_to_new):
        m = np.ascontiguousarray(old_to_new, dtype=np.int64)
        lib().orc_state_remap_tasks(self._h, _p(m), len(m))

    def set_node_status(self, idx: int, status: int):
        lib().orc_state_set_node_status(self._h, _p(self.nodes), idx, status)

    def try_form_new_groups(self) -> int:
        return lib().orc_try_form_new_groups(self._h)

    def try_merge_solo_groups(self) -> int:
        return lib().orc_try_merge_solo_groups(self._h)

    def dissolve_group(self, slot: int):
        lib().orc_dissolve_group(self._h, slot)

    def filter_tasks(self, node_idx: int):
        gi, gs, nn = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        t = lib().orc_filter_tasks_node_groups(self._h, node_idx, C.byref(gi), C.byref(gs), C.byref(nn))
        return int(t), gi.value, gs.value, nn.value

    def get_task_for_node(self, node_idx: int, use_node_groups=True) -> int:
        return int(lib().orc_get_task_for_node(self._h, node_idx, int(use_node_groups)))

    @property
    def n_groups(self) -> int:
        return lib().orc_n_groups(self._h)

    @property
    def node_to_group(self) -> np.ndarray:
        p = lib().orc_node_to_group(self._h)
        return np.ctypeslib.as_array(p, shape=(len(self.nodes),)).copy()

    def groups(self):
        """[(slot, id, config_idx, members(list of node idx in address order), task_idx)] live only."""
        out = []
        n_slots = lib().orc_group_slots(self._h)
        buf = np.zeros(max(len(self.nodes), 1), dtype=np.uint32)
        for s in range(n_slots):
            gid, cfg, n, task = C.c_uint64(0), C.c_uint32(0), C.c_uint32(0), C.c_int64(0)
            if lib().orc_group_info(self._h, s, C.byref(gid), C.byref(cfg), C.byref(n), _p(buf), len(buf),
                                    C.byref(task)):
                out.append((s, gid.value, cfg.value, buf[:n.value].tolist(), task.value))
        return out

    def drain_events(self):
        """[(kind, group id, config idx, members in BTreeSet order)] since the last drain: what the webhook plugins
        would have been called with, in call order (kind 1 = created, 2 = destroyed)"""
        nm = C.c_size_t(0)
        ne = lib().orc_events(self._h, None, 0, None, 0, C.byref(nm))
        ev = np.zeros(max(ne, 1), dtype=GROUP_EVENT)
        mem = np.zeros(max(nm.value, 1), dtype=np.uint32)
        lib().orc_events(self._h, _p(ev), ne, _p(mem), nm.value, C.byref(nm))
        lib().orc_events_clear(self._h)
        return [(int(e["kind"]), int(e["group_id"]), int(e["config"]),
                 mem[int(e["member_begin"]):int(e["member_begin"]) + int(e["n_members"])].tolist()) for e in ev[:ne]]

    def counters(self):
        a, b = C.c_uint64(0), C.c_uint64(0)
        lib().orc_counters(self._h, C.byref(a), C.byref(b))
        return a.value, b.value


# ------------------------------------------------------------------ swarm -> oracle AoS rows

def from_swarm(sw):
    """protocol_amd.swarm.Swarm (string-level data) -> (nodes, cfgs, tasks, enabled) oracle tables."""
    W, T = sw.W, sw.T
    nodes = np.zeros(W, dtype=node_dt)
    nodes["address"] = np.array([("0x%040d" % int(v)).encode() for v in sw.address], dtype=f"S{ADDR_LEN}") \
        if W else np.zeros(0, dtype=f"S{ADDR_LEN}")
    nodes["status"] = sw.status
    nodes["has_p2p"] = sw.has_p2p
    nodes["has_specs"] = sw.has_specs
    nodes["has_location"] = sw.has_loc
    nodes["latitude"] = sw.lat
    nodes["longitude"] = sw.lon
    sp = nodes["specs"]
    f = np.zeros(W, dtype=np.uint32)
    gpu, cpu = sw.has_gpu, sw.has_cpu
    f |= np.where(gpu, S_GPU, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_count_some, S_G_COUNT, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_model_some, S_G_MODEL, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_mem_some, S_G_MEM, 0).astype(np.uint32)
    f |= np.where(cpu, S_CPU, 0).astype(np.uint32)
    f |= np.where(cpu & sw.cpu_cores_some, S_CPU_CORES, 0).astype(np.uint32)
    f |= np.where(sw.ram_some, S_RAM, 0).astype(np.uint32)
    f |= np.where(sw.storage_some, S_STORAGE, 0).astype(np.uint32)
    sp["flags"] = f
    sp["gpu_count"] = sw.gpu_count
    sp["gpu_memory_mb"] = sw.gpu_mem_mb
    sp["cpu_cores"] = sw.cpu_cores
    sp["ram_mb"] = sw.ram_mb
    sp["storage_gb"] = sw.storage_gb
    names = np.array([m.encode() for m in sw.model_names], dtype=f"S{MODEL_LEN}")
    sp["gpu_model"] = names[sw.gpu_model_id] if W else names[:0]
    nodes["specs"] = sp

    cfgs = np.concatenate([make_config(n, mn, mx, r) for (n, mn, mx, r) in sw.configs]) \
        if sw.configs else np.zeros(0, dtype=config_dt)

    tasks = np.zeros(T, dtype=task_dt)
    tasks["created_at"] = sw.created_at
    tasks["restricted"] = sw.restricted
    tasks["n_topologies"] = sw.n_topo
    cname = np.array([c[0].encode() for c in sw.configs] + [b"ghost-topology", b""], dtype=f"S{NAME_LEN}")
    topo = sw.topo.astype(np.int64)
    topo = np.where(topo == -1, len(sw.configs), np.where(topo == -2, len(sw.configs) + 1, topo))
    tp = np.zeros((T, MAX_TOPO), dtype=f"S{NAME_LEN}")
    if T:
        tp[:, :topo.shape[1]] = cname[topo]
    tasks["topologies"] = tp
    enabled = np.array([(sw.enabled_mask() >> i) & 1 for i in range(len(sw.configs))], dtype=np.uint8)
    return nodes, cfgs, tasks, enabled


# ---- group variables (scheduler_impl.rs:155-200, storage.rs:150-215)

def _take(ptr) -> str:
    try:
        return C.string_at(ptr).decode()
    finally:
        lib().orc_free_string(ptr)


def group_vars(text, group_index, group_size, next_p2p_address, group_id, total_upload_count) -> str:
    return _take(lib().orc_group_vars(text.encode(), group_index, group_size, next_p2p_address.encode(),
                                      group_id.encode(), total_upload_count.encode()))


def volume_vars(text, group_id) -> str:
    return _take(lib().orc_volume_vars(text.encode(), group_id.encode()))


def upload_name_vars(text, group_id, group_size, group_index, upload_count) -> str:
    gid = None if group_id is None else group_id.encode()
    return _take(lib().orc_upload_name_vars(text.encode(), gid, group_size, group_index, upload_count))


def last_file_idx(total_upload_count) -> int:
    return int(lib().orc_last_file_idx(total_upload_count.encode()))
