"""Host-side packing: string-level swarm data -> the engine's SoA views.

This is what the Rust shim does in production: intern GPU model strings, parse the
`k=v;...` requirement strings with ComputeRequirements::from_str semantics, evaluate the model
substring rule once per (requirement model, spec model class) pair, and project
OrchestratorNode / Task rows into columns.  All rule evaluation goes through the product's own C
helpers (pm_host_* in libpm_engine.so) — nothing here touches oracle/.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import engine as E
from .swarm import ST_HEALTHY, Swarm


def parse_requirements(s: str, alt_cap: int = 64):
    """ComputeRequirements::from_str via the C ABI -> (config_row, alt_rows, model strings per alt)."""
    cfg = np.zeros(1, dtype=E.config_row_dt)
    alts = np.zeros(alt_cap, dtype=E.alt_row_dt)
    models = C.create_string_buffer(4096)
    E.check(E.lib().pm_host_parse_requirements(s.encode(), cfg.ctypes.data, alts.ctypes.data, alt_cap, models, 4096))
    n = int(cfg["alt_count"][0])
    alts = alts[:n].copy()
    names = []
    for a in alts:
        if int(a["flags"]) & E.G_MODEL:
            off = int(a["model_row"])
            end = models.raw.index(b"\0", off)
            names.append(models.raw[off:end].decode())
        else:
            names.append(None)
    return cfg[0], alts, names


def model_matches(spec_model: str, req_model: str) -> bool:
    return bool(E.check(E.lib().pm_host_model_matches(spec_model.encode(), req_model.encode())))


def pack_configs(configs: list):
    """[(name, min, max, requirement string | None)] -> (cfg_rows, alt_rows, req_models)

    req_models[i] is the requirement model string of model-table row i; alt_rows[*].model_row points
    into it."""
    cfg_rows = np.zeros(len(configs), dtype=E.config_row_dt)
    alt_list, req_models = [], []
    for i, (_name, mn, mx, req) in enumerate(configs):
        row = np.zeros(1, dtype=E.config_row_dt)[0]
        if req is not None:
            row, alts, names = parse_requirements(req)
            row = row.copy()
            row["alt_begin"] = len(alt_list)
            for a, nm in zip(alts, names):
                a = a.copy()
                if nm is not None:
                    a["model_row"] = len(req_models)
                    req_models.append(nm)
                alt_list.append(a)
        row["min_group_size"] = mn
        row["max_group_size"] = mx
        cfg_rows[i] = row
    alt_rows = np.array(alt_list, dtype=E.alt_row_dt) if alt_list else np.zeros(0, dtype=E.alt_row_dt)
    return cfg_rows, alt_rows, req_models


def build_model_table(req_models: list, spec_models: list) -> np.ndarray:
    n_rows, n_cls = len(req_models), len(spec_models)
    words = (n_cls + 31) // 32
    bits = np.zeros(max(n_rows * words, 1), dtype=np.uint32)
    ra = (C.c_char_p * max(n_rows, 1))(*[m.encode() for m in req_models])
    sa = (C.c_char_p * max(n_cls, 1))(*[m.encode() for m in spec_models])
    E.check(E.lib().pm_host_build_model_table(ra, n_rows, sa, n_cls, bits.ctypes.data))
    return bits[:n_rows * words]


def to_lowercase(text: str) -> str:
    """str::to_lowercase as the product evaluates it (pm_host_to_lowercase)"""
    raw = text.encode()
    need = C.c_size_t(0)
    E.check(E.lib().pm_host_to_lowercase(raw, None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    E.check(E.lib().pm_host_to_lowercase(raw, buf, len(buf), C.byref(need)))
    return buf.value.decode()


def config_order(cfg_rows: np.ndarray, enabled: int) -> list:
    out = np.zeros(max(len(cfg_rows), 1), dtype=np.uint32)
    n = C.c_uint32(0)
    cfg_rows = np.ascontiguousarray(cfg_rows, dtype=E.config_row_dt)
    E.check(E.lib().pm_host_config_order(cfg_rows.ctypes.data, len(cfg_rows), enabled & 0xFFFFFFFFFFFFFFFF,
                                         out.ctypes.data, C.byref(n)))
    return out[:n.value].tolist()


def worker_flags(sw: Swarm) -> np.ndarray:
    f = np.zeros(sw.W, dtype=np.uint32)
    spec = sw.has_specs
    gpu = spec & sw.has_gpu
    cpu = spec & sw.has_cpu
    f |= np.where(spec, E.W_HAS_SPECS, 0).astype(np.uint32)
    f |= np.where(gpu, E.W_HAS_GPU, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_count_some, E.W_GPU_COUNT, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_mem_some, E.W_GPU_MEM, 0).astype(np.uint32)
    f |= np.where(gpu & sw.gpu_model_some, E.W_GPU_MODEL, 0).astype(np.uint32)
    f |= np.where(cpu, E.W_HAS_CPU, 0).astype(np.uint32)
    f |= np.where(cpu & sw.cpu_cores_some, E.W_CPU_CORES, 0).astype(np.uint32)
    f |= np.where(spec & sw.ram_some, E.W_RAM, 0).astype(np.uint32)
    f |= np.where(spec & sw.storage_some, E.W_STORAGE, 0).astype(np.uint32)
    f |= np.where(sw.status == ST_HEALTHY, E.W_HEALTHY, 0).astype(np.uint32)
    f |= np.where(sw.has_p2p, E.W_HAS_P2P, 0).astype(np.uint32)
    f |= np.where(sw.has_loc, E.W_HAS_LOC, 0).astype(np.uint32)
    return f


def pack_workers(sw: Swarm) -> dict:
    return dict(flags=worker_flags(sw), gpu_count=sw.gpu_count, gpu_mem_mb=sw.gpu_mem_mb,
                gpu_model_class=sw.gpu_model_id, cpu_cores=sw.cpu_cores, ram_mb=sw.ram_mb,
                storage_gb=sw.storage_gb, price=sw.price, addr_rank=sw.addr_rank(), lat=sw.lat, lon=sw.lon)


def load_swarm(eng: E.Engine, sw: Swarm, *, enabled: int | None = None):
    """Upload a whole swarm: configs + model table + workers + tasks + enabled set."""
    cfg_rows, alt_rows, req_models = pack_configs(sw.configs)
    eng.set_configs(cfg_rows, alt_rows)
    bits = build_model_table(req_models, sw.model_names)
    eng.set_model_table(bits, len(req_models), len(sw.model_names))
    eng.upload_workers(pack_workers(sw))
    eng.upload_tasks(sw.task_masks(), sw.created_at, sw.task_uid)
    eng.set_enabled_mask(sw.enabled_mask() if enabled is None else enabled)
    return cfg_rows, alt_rows, req_models


# ---- group variables of the assigned task

def _render(call) -> str:
    need = C.c_size_t(0)
    E.check(call(None, 0, C.byref(need)))
    buf = C.create_string_buffer(need.value)
    E.check(call(buf, need.value, C.byref(need)))
    return buf.value.decode()


def group_vars(text: str, group_index: int, group_size: int, next_p2p_address: str, group_id: str,
               total_upload_count: str) -> str:
    """scheduler_impl.rs:160-183 for one env-var value / cmd argument."""
    v = E.GroupVars(group_index, group_size, next_p2p_address.encode(), group_id.encode(), total_upload_count.encode())
    return _render(lambda o, c, n: E.lib().pm_host_group_vars(text.encode(), C.byref(v), o, c, n))


def volume_vars(text: str, group_id: str) -> str:
    """scheduler_impl.rs:185-200 for host_path / container_path."""
    return _render(lambda o, c, n: E.lib().pm_host_volume_vars(text.encode(), group_id.encode(), o, c, n))


def upload_name_vars(text: str, group_id, group_size: int, group_index: int, upload_count: int) -> str:
    """storage.rs:150-215 for the upload file-name template (group_id None: the node is in no group)."""
    gid = None if group_id is None else group_id.encode()
    return _render(lambda o, c, n: E.lib().pm_host_upload_name_vars(text.encode(), gid, group_size, group_index,
                                                                    upload_count, o, c, n))


def last_file_idx(total_upload_count: str) -> int:
    return int(E.lib().pm_host_last_file_idx(total_upload_count.encode()))
