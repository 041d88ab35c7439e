"""Seeded synthetic swarms for the five BASELINE.json configurations.

Everything derives from one 64-bit seed through splitmix64, written out below so a Rust host can
replicate the generator bit for bit:

    state += 0x9E3779B97F4A7C15; z = state
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9; z = (z ^ (z >> 27)) * 0x94D049BB133111EB
    out = z ^ (z >> 31)

Field k of the swarm uses the independent stream whose start state is mix(seed ^ mix(k)); the i-th
draw of a stream is therefore mix(state0 + (i+1) * GAMMA) and is computed vectorised here.  Uniform
floats are (x >> 11) * 2^-53.

A swarm is string-level data, like what the reference's stores return (model strings, config names,
requirement strings, address strings); protocol_amd.host packs it into the engine's SoA views and
oracle/oracle_ffi.py converts it into the oracle's AoS rows.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def mix64(x):
    """splitmix64 output function applied to state+GAMMA (works on numpy uint64 arrays)."""
    with np.errstate(over="ignore"):
        z = (np.asarray(x, dtype=np.uint64) + GAMMA).astype(np.uint64)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class Stream:
    """Vectorised splitmix64 stream."""

    def __init__(self, seed: int, salt: int):
        self.state = int(mix64(np.uint64(seed) ^ mix64(np.uint64(salt))))

    def u64(self, n: int) -> np.ndarray:
        with np.errstate(over="ignore"):
            idx = np.arange(0, n, dtype=np.uint64)
            st = np.uint64(self.state) + idx * GAMMA  # state before the (i+1)-th increment
            out = mix64(st)
            self.state = int((np.uint64(self.state) + np.uint64(n) * GAMMA).astype(np.uint64)) if n else self.state
        return out

    def uniform(self, n: int) -> np.ndarray:
        return (self.u64(n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)

    def randint(self, n: int, lo: int, hi: int) -> np.ndarray:
        """integers in [lo, hi] (inclusive); modulo bias is irrelevant for test data."""
        return (self.u64(n) % np.uint64(hi - lo + 1)).astype(np.int64) + lo

    def choice(self, n: int, probs) -> np.ndarray:
        cdf = np.cumsum(np.asarray(probs, dtype=np.float64))
        cdf[-1] = 1.0
        return np.searchsorted(cdf, self.uniform(n), side="right").astype(np.int64)


# 12 real-world model strings (first four are quoted in SURVEY.md §8d) and their per-card memory.
GPU_MODELS = [
    "NVIDIA H100 80GB HBM3", "NVIDIA A100-SXM4-80GB", "nvidia rtx a6000", "RTX 4090",
    "NVIDIA H100 PCIe", "NVIDIA A100-SXM4-40GB", "NVIDIA A100 80GB PCIe", "NVIDIA GeForce RTX 3090",
    "NVIDIA L40S", "NVIDIA H200", "Tesla V100-SXM2-32GB", "AMD Instinct MI300X",
]
GPU_MEMORY_MB = [81559, 81559, 49140, 24564, 81559, 40960, 81559, 24564, 49140, 143771, 32768, 196592]

# (name, min, max, requirements) — the mixed cfg2..cfg5 table: gpu:count, gpu:model, memory ranges,
# total memory, ram/storage/cpu, K in {1,2,3} alternatives, and unconstrained configurations.
MIXED_CONFIGS = [
    ("h100x8-quad", 4, 4, "gpu:count=8;gpu:model=h100"),
    ("big-ram-a100-h100", 2, 4, "gpu:count=8;gpu:model=a100,h100;ram_mb=256000"),
    ("h200-pair", 2, 2, "gpu:count=8;gpu:model=H200"),
    ("a100-80g-x4", 4, 8, "gpu:count=4;gpu:model=a100;gpu:memory_mb_min=80000"),
    ("dense-mem", 8, 8, "gpu:count=8;gpu:memory_mb_min=40000;gpu:count=4;gpu:memory_mb_min=80000"),
    ("consumer-4090", 2, 4,
     "gpu:count=1;gpu:model=rtx4090,rtx_3090;gpu:count=2;gpu:model=rtx4090;gpu:count=4;gpu:model=rtx4090"),
    ("total-300g", 3, 6, "gpu:total_memory_min=300000"),
    ("total-100-200g", 2, 3, "gpu:total_memory_min=100000;gpu:total_memory_max=200000"),
    ("workstation", 1, 2, "gpu:model=a6000,l40s;cpu:cores=32"),
    ("dual-storage", 2, 2, "gpu:count=2;storage_gb=2000"),
    ("fat-host", 2, 4, "ram_mb=512000;cpu:cores=64"),
    ("any-x4", 4, 4, "gpu:count=4"),
    ("any-x2", 2, 2, "gpu:count=2"),
    ("any-x1", 1, 1, "gpu:count=1"),
    ("open-wide", 2, 16, None),
    ("open-small", 1, 4, None),
    ("v100-solo", 1, 1, "gpu:model=v100;gpu:memory_mb=32000"),
    ("mi300x-ring", 1, 8, "gpu:model=mi300x"),
    ("mid-mem-5", 5, 5, "gpu:memory_mb_min=24000;gpu:memory_mb_max=50000;storage_gb=500"),
    ("x8-tier1", 8, 16, "gpu:count=8;gpu:model=a100;gpu:count=8;gpu:model=h100;gpu:count=8;gpu:model=h200"),
    ("cpu-monster", 1, 1, "cpu:cores=128;ram_mb=1000000"),
    ("archive", 2, 2, "storage_gb=20000"),
    ("single-80g", 3, 3, "gpu:count=1;gpu:memory_mb=80000"),
    ("headless", 1, 1, "gpu:count=0"),
]

# cfg1: "uniform GPU-count constraint"
UNIFORM_CONFIGS = [
    ("count-8", 8, 8, "gpu:count=8"),
    ("count-4", 4, 4, "gpu:count=4"),
    ("count-2", 2, 2, "gpu:count=2"),
    ("count-1", 1, 1, "gpu:count=1"),
]

# NodeStatus declaration order (crates/orchestrator/src/models/node.rs:75-85)
ST_DISCOVERED, ST_WAITING, ST_HEALTHY, ST_UNHEALTHY, ST_DEAD, ST_EJECTED, ST_BANNED, ST_LOWBALANCE = range(8)


@dataclass
class Swarm:
    seed: int
    # ---- workers (NodeStore::get_nodes order)
    address: np.ndarray        # (W,) uint64 — the address string is "0x%040d" % value (digit-only)
    status: np.ndarray         # (W,) uint8 NodeStatus
    has_p2p: np.ndarray        # (W,) bool
    has_specs: np.ndarray      # (W,) bool   compute_specs.is_some()
    has_gpu: np.ndarray        # (W,) bool   specs.gpu.is_some()
    gpu_count_some: np.ndarray
    gpu_mem_some: np.ndarray
    gpu_model_some: np.ndarray
    has_cpu: np.ndarray
    cpu_cores_some: np.ndarray
    ram_some: np.ndarray
    storage_some: np.ndarray
    gpu_count: np.ndarray      # (W,) uint32
    gpu_mem_mb: np.ndarray
    gpu_model_id: np.ndarray   # (W,) index into model_names
    cpu_cores: np.ndarray
    ram_mb: np.ndarray
    storage_gb: np.ndarray
    price: np.ndarray          # extension column, all zero
    has_loc: np.ndarray
    lat: np.ndarray            # (W,) float64
    lon: np.ndarray
    model_names: list
    # ---- configurations (NODE_GROUP_CONFIGS order)
    configs: list              # [(name, min, max, requirement string | None)]
    # ---- tasks (TaskStore::get_all_tasks order: created_at desc, stable)
    created_at: np.ndarray     # (T,) int64
    task_uid: np.ndarray       # (T,) uint64
    restricted: np.ndarray     # (T,) bool — allowed_topologies key present
    n_topo: np.ndarray         # (T,) uint8
    topo: np.ndarray           # (T, 3) int16 config index, -1 = a name no configuration has, -2 = unused
    meta: dict = field(default_factory=dict)

    @property
    def W(self) -> int:
        return len(self.address)

    @property
    def T(self) -> int:
        return len(self.created_at)

    def address_strings(self) -> list:
        return ["0x%040d" % int(v) for v in self.address]

    def addr_rank(self) -> np.ndarray:
        """rank of the address string in byte order == rank of the integer (equal-length digit strings)"""
        order = np.argsort(self.address, kind="stable")
        rank = np.empty(self.W, dtype=np.uint32)
        rank[order] = np.arange(self.W, dtype=np.uint32)
        return rank

    def task_masks(self) -> np.ndarray:
        """u64 topology masks exactly as a host would derive them from the topology name lists."""
        m = np.zeros(self.T, dtype=np.uint64)
        for k in range(self.topo.shape[1]):
            col = self.topo[:, k].astype(np.int64)
            ok = col >= 0
            m[ok] |= np.uint64(1) << col[ok].astype(np.uint64)
        m[~self.restricted] = np.uint64(0xFFFFFFFFFFFFFFFF)
        return m

    def enabled_mask(self) -> int:
        """available_node_group_configs after on_task_created for every task (mod.rs:1224-1243)."""
        m = 0
        cols = self.topo[self.restricted]
        for c in np.unique(cols[cols >= 0]):
            m |= 1 << int(c)
        return m


def _gen_workers(seed: int, W: int, uniform_gpu: bool):
    s = lambda salt: Stream(seed, salt)
    address = s(1).u64(W) >> np.uint64(1)
    # uniqueness: re-draw collisions deterministically (practically never triggers)
    _, first = np.unique(address, return_index=True)
    if len(first) != W:
        dup = np.ones(W, dtype=bool)
        dup[first] = False
        address[dup] = mix64(address[dup] ^ np.uint64(0xA5A5A5A5)) >> np.uint64(1)
    status = np.full(W, ST_HEALTHY, dtype=np.uint8)
    u = s(2).uniform(W)
    other = np.array([ST_DISCOVERED, ST_WAITING, ST_UNHEALTHY, ST_DEAD, ST_EJECTED, ST_BANNED, ST_LOWBALANCE],
                     dtype=np.uint8)
    not_healthy = u >= 0.97
    status[not_healthy] = other[s(3).randint(W, 0, 6)[not_healthy]]
    has_p2p = s(4).uniform(W) < 0.99
    gpu_count = np.array([1, 2, 4, 8], dtype=np.uint32)[s(5).choice(W, [0.35, 0.2, 0.2, 0.25])]
    model_id = s(6).randint(W, 0, len(GPU_MODELS) - 1).astype(np.uint32)
    if uniform_gpu:
        model_id[:] = 0
    gpu_mem = np.array(GPU_MEMORY_MB, dtype=np.uint32)[model_id]
    cpu_cores = np.array([8, 16, 32, 48, 64, 96, 128, 192, 256], dtype=np.uint32)[s(7).randint(W, 0, 8)]
    ram = s(8).randint(W, 16000, 2000000).astype(np.uint32)
    storage = s(9).randint(W, 100, 30000).astype(np.uint32)
    # 5 % of the rows lose one Option field (exercise every None branch of ComputeSpecs::meets)
    drop = s(10).uniform(W) < 0.05
    which = s(11).randint(W, 0, 8)
    flags = {k: np.ones(W, dtype=bool) for k in
             ("has_specs", "has_gpu", "gpu_count_some", "gpu_mem_some", "gpu_model_some", "has_cpu",
              "cpu_cores_some", "ram_some", "storage_some")}
    for i, k in enumerate(flags):
        flags[k][drop & (which == i)] = False
    has_loc = s(12).uniform(W) < 0.90
    # lat in [25,60]; lon in [-125,-65] U [-10,40]; 0.0001 degree grid
    lat = np.round(25.0 + 35.0 * s(13).uniform(W), 4)
    ul = s(14).uniform(W)
    lon = np.where(ul < 60.0 / 110.0, -125.0 + ul * 110.0, -10.0 + (ul * 110.0 - 60.0))
    lon = np.round(lon, 4)
    # 40 % snapped to 32 "city" points (exact distance ties)
    n_city = 32
    city_lat = np.round(25.0 + 35.0 * Stream(seed, 100).uniform(n_city), 4)
    cu = Stream(seed, 101).uniform(n_city)
    city_lon = np.round(np.where(cu < 60.0 / 110.0, -125.0 + cu * 110.0, -10.0 + (cu * 110.0 - 60.0)), 4)
    snap = s(15).uniform(W) < 0.40
    city = s(16).randint(W, 0, n_city - 1)
    lat = np.where(snap, city_lat[city], lat)
    lon = np.where(snap, city_lon[city], lon)
    return dict(address=address, status=status, has_p2p=has_p2p, gpu_count=gpu_count, gpu_mem_mb=gpu_mem,
                gpu_model_id=model_id, cpu_cores=cpu_cores, ram_mb=ram, storage_gb=storage,
                price=np.zeros(W, dtype=np.uint32), has_loc=has_loc, lat=lat, lon=lon, **flags)


def _gen_tasks(seed: int, T: int, configs: list, zipf: bool, t0_ms: int = 1_754_000_000_000):
    s = lambda salt: Stream(seed, 1000 + salt)
    C = len(configs)
    created = t0_ms + s(1).randint(T, 0, 86_400_000)  # ms timestamps over one day
    dupe = s(2).uniform(T) < 0.02                      # 2 % exact duplicates (tie-break test)
    src = s(3).randint(T, 0, max(T - 1, 0))
    created = np.where(dupe, created[src], created).astype(np.int64)
    n_topo = (1 + s(4).choice(T, [0.6, 0.3, 0.1])).astype(np.uint8)
    if zipf:
        # topology of task i ~ Zipf(s=1.1) over configs ranked by min_group_size (desc)
        rank = np.argsort([-c[1] for c in configs], kind="stable")
        w = 1.0 / np.power(np.arange(1, C + 1, dtype=np.float64), 1.1)
        w /= w.sum()
        pick = lambda salt: rank[s(salt).choice(T, w)]
    else:
        pick = lambda salt: s(salt).randint(T, 0, C - 1)
    topo = np.stack([pick(5), pick(6), pick(7)], axis=1).astype(np.int16)
    for k in range(3):
        topo[n_topo <= k, k] = -2
    kind = s(8).uniform(T)
    restricted = kind >= 0.02           # 2 % unrestricted (scheduling_config None)
    ghost = (kind >= 0.02) & (kind < 0.03)  # 1 % name a topology no configuration has
    topo[ghost, 0] = -1
    topo[ghost, 1:] = -2
    n_topo = np.where(ghost, 1, n_topo).astype(np.uint8)
    n_topo = np.where(restricted, n_topo, 0).astype(np.uint8)
    topo[~restricted] = -2
    uid = mix64(np.arange(T, dtype=np.uint64) ^ np.uint64(seed * 2654435761 & 0xFFFFFFFFFFFFFFFF))
    # TaskStore::get_all_tasks: RPUSH order then stable sort created_at desc (task_store.rs:61,79)
    order = np.argsort(-created, kind="stable")
    return dict(created_at=created[order], task_uid=uid[order], restricted=restricted[order],
                n_topo=n_topo[order], topo=topo[order])


def make_swarm(seed: int, n_tasks: int, n_workers: int, *, configs: str = "mixed", n_configs: int | None = None,
               zipf: bool = False) -> Swarm:
    cfgs = list(UNIFORM_CONFIGS if configs == "uniform" else MIXED_CONFIGS)
    if n_configs is not None:
        cfgs = cfgs[:n_configs]
    w = _gen_workers(seed, n_workers, uniform_gpu=(configs == "uniform"))
    t = _gen_tasks(seed, n_tasks, cfgs, zipf)
    return Swarm(seed=seed, model_names=list(GPU_MODELS), configs=cfgs, **w, **t,
                 meta=dict(n_tasks=n_tasks, n_workers=n_workers, configs=configs, zipf=zipf))


# The five BASELINE.json configurations.
def baseline_config(i: int, seed: int = 1, scale: float = 1.0) -> Swarm:
    sz = lambda n: max(1, int(n * scale))
    if i == 0:
        return make_swarm(seed, sz(1000), sz(256), configs="uniform")
    if i == 1:
        return make_swarm(seed, sz(100_000), sz(10_000))
    if i in (2, 3):
        return make_swarm(seed, sz(1_000_000), sz(100_000), zipf=True)
    if i == 4:
        return make_swarm(seed, sz(10_000), sz(100_000))
    raise ValueError(i)


def solo_merge_swarm(seed: int = 1, W: int = 100000, solos: int = 5000, T: int = 1000) -> Swarm:
    """The merge pass's own workload (try_merge_solo_groups, mod.rs:631-971; bench.py `merge`, tests/test_gpu_scale.py):
    W workers of which exactly `solos` healthy 1-GPU nodes match a (1, 1) configuration — formed with only that
    configuration enabled they are `solos` single-node groups — and a (2, 8) configuration over the same nodes that,
    once enabled, merges them.  Enable masks: 0b01 to form the solos, 0b11 to merge."""
    sw = make_swarm(seed, T, W, zipf=(W >= 100000))
    sw.configs = [("solo-1gpu", 1, 1, "gpu:count=1"), ("octet-1gpu", 2, 8, "gpu:count=1")]
    sw.topo[:] = -2
    sw.restricted[:] = False
    sw.n_topo[:] = 0
    one = np.nonzero((sw.gpu_count == 1) & (sw.status == ST_HEALTHY))[0]
    if len(one) < solos:
        raise ValueError(f"only {len(one)} healthy 1-GPU nodes among {W} workers")
    sw.status[one[solos:]] = ST_UNHEALTHY
    return sw


def events_digest(events) -> str:
    """digest of a life-cycle feed [(kind, group id, config, members)] in the order it was emitted"""
    import hashlib
    return hashlib.sha256(repr([(int(k), int(g), int(c), [int(w) for w in m]) for k, g, c, m in events]).encode()).hexdigest()[:16]
