"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs,
plus the reference's known-answer vectors pushed through the GPU predicate.  Bit-exact throughout
(integer / index work; the proximity order is proven equal or settled exactly on the host)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import GPU_MODELS, Swarm, baseline_config, make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for

pytestmark = pytest.mark.gpu

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "node_rs_kats.json")))
NONE = 0xFFFFFFFF


def _one_worker_cols(gc, gm, gmem, cores, ram, sto, model_class):
    f = E.W_HAS_SPECS | E.W_HEALTHY | E.W_HAS_P2P
    if gc is not None or gm is not None or gmem is not None:
        f |= E.W_HAS_GPU
    if gc is not None:
        f |= E.W_GPU_COUNT
    if gm is not None:
        f |= E.W_GPU_MODEL
    if gmem is not None:
        f |= E.W_GPU_MEM
    if cores is not None:
        f |= E.W_HAS_CPU | E.W_CPU_CORES
    if ram is not None:
        f |= E.W_RAM
    if sto is not None:
        f |= E.W_STORAGE
    z = lambda v: np.array([v or 0], dtype=np.uint32)
    return dict(flags=np.array([f], dtype=np.uint32), gpu_count=z(gc), gpu_mem_mb=z(gmem),
                gpu_model_class=z(model_class), cpu_cores=z(cores), ram_mb=z(ram), storage_gb=z(sto),
                lat=np.zeros(1), lon=np.zeros(1))


def test_meets_kats_through_the_gpu_predicate():
    """crates/shared/src/models/node.rs:740-1227 — all 29 `meets` vectors, one launch per vector."""
    eng = E.Engine()
    for kat in KATS["meets"]:
        gc, gm, gmem, cores, ram, sto = kat["specs"]
        cfg_rows, alt_rows, req_models = host.pack_configs([("k", 1, 1, kat["req"])])
        spec_models = [gm] if gm is not None else ["unused"]
        eng.set_configs(cfg_rows, alt_rows)
        eng.set_model_table(host.build_model_table(req_models, spec_models), len(req_models), len(spec_models))
        eng.upload_workers(_one_worker_cols(gc, gm, gmem, cores, ram, sto, 0))
        assert bool(eng.compat_masks()[0] & 1) is kat["expect"], kat["name"]
    eng.close()


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("cfg", [0, 1])
def test_compat_masks_bit_exact(cfg, seed):
    sw = baseline_config(cfg, seed=seed)
    nodes, cfgs, _, _ = orc.from_swarm(sw)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    assert np.array_equal(eng.compat_masks(), orc.compat_masks(nodes, cfgs))
    eng.close()


def test_compat_masks_edge_rows():
    # (None req, _) => true; (Some req, None specs) => false; empty tables
    sw = make_swarm(9, 10, 300)
    sw.has_specs[:100] = False
    nodes, cfgs, _, _ = orc.from_swarm(sw)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    assert np.array_equal(eng.compat_masks(), orc.compat_masks(nodes, cfgs))
    empty = make_swarm(9, 0, 0)
    host.load_swarm(eng, empty)
    assert eng.compat_masks().shape == (0,)
    eng.close()


def _form_both(sw, **kw):
    st = oracle_state_for(sw, reference_shaped=(sw.W <= 2048), **{k: v for k, v in kw.items()
                                                                   if k in ("proximity", "group_id_seed")})
    kw = dict(kw)
    eng = E.Engine(**kw)
    host.load_swarm(eng, sw)
    return st, eng


@pytest.mark.parametrize("carve_variant", [0, 1, 3])
@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5])
def test_form_groups_cfg1_bit_exact(seed, carve_variant):
    sw = baseline_config(0, seed=seed)
    st, eng = _form_both(sw, group_id_seed=seed, carve_variant=carve_variant)
    n_o, n_e = st.try_form_new_groups(), eng.form_groups()
    assert n_o == n_e and oracle_groups(st) == engine_groups(eng)
    assert np.array_equal(eng.get_groups()[0] >= 0, st.node_to_group >= 0)
    assert eng.last_stats()["host_resolved_steps"] == 0
    eng.close()


@pytest.mark.parametrize("carve_variant", [0, 1, 3])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_form_groups_cfg2_bit_exact(seed, carve_variant):
    sw = baseline_config(1, seed=seed)
    st, eng = _form_both(sw, group_id_seed=seed, carve_variant=carve_variant)
    n_o, n_e = st.try_form_new_groups(), eng.form_groups()
    assert n_o == n_e
    assert oracle_groups(st) == engine_groups(eng)
    stats = eng.last_stats()
    assert stats["host_resolved_steps"] == 0
    if carve_variant in (0, 3):   # most steps must come straight from the neighbour-list proposals
        assert stats["carve_fast_steps"] > 0.5 * stats["carve_steps"]
    eng.close()


def test_form_groups_wide_and_huge_groups():
    """max_group_size beyond the proposal row (63) and beyond the per-wave partial lists (64)."""
    sw = make_swarm(21, 100, 2500)
    sw.configs = [("huge", 100, 300, "gpu:count=8"), ("wide", 40, 60, "gpu:count=4"), ("rest", 2, 70, None)]
    sw.topo = (sw.topo.astype(np.int64) % 3).astype(np.int16)
    sw.topo[sw.n_topo[:, None] <= np.arange(3)[None, :]] = -2
    sw.topo[~sw.restricted] = -2
    for variant in (0, 1, 3):
        st = oracle_state_for(sw)
        eng = E.Engine(carve_variant=variant)
        host.load_swarm(eng, sw)
        st.try_form_new_groups()
        eng.form_groups()
        assert oracle_groups(st) == engine_groups(eng)
        eng.close()


@pytest.mark.parametrize("W", [300, 900, 2500])
@pytest.mark.parametrize("seed", [31, 32, 33])
def test_configurations_that_end_or_start_in_registers(seed, W):
    """Configurations of a wave's worth of located candidates or fewer are carved on sorted rows made by helper waves
    (stream_small_rows), 65..128 start two a lane and finish there, groups of ONE node are written in one go
    (stream_first_come in two parts: located first) — small swarms where every configuration is such a one, the
    co-located workers of the 32 cities in them (ties: the certificate's second look), with and without every second
    step forced through the host."""
    sw = make_swarm(seed, 400, W)
    sw.configs = [("quad8", 4, 4, "gpu:count=8"), ("pairs4", 2, 2, "gpu:count=4"), ("wide2", 3, 7, "gpu:count=2"),
                  ("singles-h100", 1, 1, "gpu:count=1;gpu:model=h100"), ("rest1", 2, 6, "gpu:count=1"), ("solo-any", 1, 1, None)]
    sw.topo = (sw.topo.astype(np.int64) % len(sw.configs)).astype(np.int16)
    sw.topo[sw.n_topo[:, None] <= np.arange(3)[None, :]] = -2
    sw.topo[~sw.restricted] = -2
    for every in (0, 2):
        st = oracle_state_for(sw)
        st.try_form_new_groups()
        eng = E.Engine(debug_uncertain_every=every)
        host.load_swarm(eng, sw)
        eng.form_groups()
        stats = eng.last_stats()
        assert oracle_groups(st) == engine_groups(eng), (seed, W, every)
        assert (stats["host_resolved_steps"] > 0) == (every != 0)
        assert eng.debug_carve_counters()["stream_aborts"] == 0
        eng.close()


def test_form_groups_without_proximity_and_with_partial_enable():
    sw = make_swarm(11, 500, 3000)
    for kw, enabled in ((dict(proximity=False), None), (dict(), 0b1010_1010_0110_0101_0011_0001)):
        st = oracle_state_for(sw, **kw)
        eng = E.Engine(**kw)
        host.load_swarm(eng, sw, enabled=enabled)
        if enabled is not None:
            st.set_enabled(np.array([(enabled >> i) & 1 for i in range(len(sw.configs))], dtype=np.uint8))
        st.try_form_new_groups()
        eng.form_groups()
        assert oracle_groups(st) == engine_groups(eng)
        eng.close()


@pytest.mark.parametrize("carve_variant", [0, 1, 3])
def test_host_resolve_path_gives_identical_groups(carve_variant):
    """debug_uncertain_every forces the exact host path (glibc distances) on every 3rd step."""
    sw = make_swarm(4, 200, 1500)
    st = oracle_state_for(sw)
    st.try_form_new_groups()
    eng = E.Engine(debug_uncertain_every=3, carve_variant=carve_variant)
    host.load_swarm(eng, sw)
    eng.form_groups()
    stats = eng.last_stats()
    assert stats["host_resolved_steps"] > 10
    assert oracle_groups(st) == engine_groups(eng)
    eng.close()


def test_incremental_ticks_and_death_reformation():
    """The reference is incremental (Appendix A): existing groups are sticky, a dead node dissolves its
    whole group (status_update_impl.rs:17-29) and survivors re-enter the next carve."""
    sw = make_swarm(5, 300, 800)
    late = np.arange(sw.W) % 7 == 0
    status0 = sw.status.copy()
    sw.status = np.where(late, 0, sw.status).astype(np.uint8)      # join later
    st = oracle_state_for(sw)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    st.try_form_new_groups()
    eng.form_groups()
    assert oracle_groups(st) == engine_groups(eng)
    flags = host.worker_flags(sw)
    for w in np.nonzero(late & (status0 == 2))[0]:
        st.set_node_status(int(w), 2)
        eng.on_worker_status(int(w), int(flags[w]) | E.W_HEALTHY, False)
    st.try_form_new_groups()
    eng.form_groups()
    assert oracle_groups(st) == engine_groups(eng)
    victims = [int(g[2][0]) for g in engine_groups(eng)[::9]]
    for w in victims:
        st.set_node_status(w, 4)
        eng.on_worker_status(w, int(flags[w]) & ~E.W_HEALTHY, True)
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    st.try_form_new_groups()
    eng.form_groups()
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    eng.close()


def _solo_scenario(seed, tasks_only_for=None, **kw):
    """Nodes trickle in one per tick against (1, k) configurations => many solo groups, then merge."""
    sw = make_swarm(seed, 50, 60, n_configs=24)
    sw.configs = [("pairs-gpu8", 1, 2, "gpu:count=8"), ("trio", 1, 3, None), ("solo-only", 1, 1, "gpu:count=1")]
    sw.topo = (sw.topo.astype(np.int64) % 3).astype(np.int16)
    sw.topo[sw.n_topo[:, None] <= np.arange(3)[None, :]] = -2
    sw.topo[~sw.restricted] = -2
    if tasks_only_for is not None:      # every task names exactly one configuration
        sw.restricted[:] = True
        sw.n_topo[:] = 1
        sw.topo[:, 0] = tasks_only_for
        sw.topo[:, 1:] = -2
    status0 = sw.status.copy()
    sw.status[:] = 0
    st = oracle_state_for(sw, **{k: v for k, v in kw.items() if k in ("switching", "prefer_larger", "chooser", "chooser_seed")})
    eng = E.Engine(**kw)
    host.load_swarm(eng, sw)
    flags = host.worker_flags(sw)
    for w in range(sw.W):
        if status0[w] != 2:
            continue
        st.set_node_status(w, 2)
        eng.on_worker_status(w, int(flags[w]) | E.W_HEALTHY, False)
        st.try_form_new_groups()
        eng.form_groups()
    return sw, st, eng


@pytest.mark.parametrize("stream_min", [None, "8"])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_merge_solo_groups_bit_exact(seed, stream_min, monkeypatch):
    """stream_min: from how many compatible solo groups a merge configuration's selections go through the streaming carve
    (512 in production; 8 sends these small lists — location-less nodes, lists shorter than a group, configurations of
    every size — down that path too)"""
    if stream_min:
        monkeypatch.setenv("PM_MERGE_STREAM_MIN", stream_min)
    sw, st, eng = _solo_scenario(seed)
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    assert sum(len(g[2]) == 1 for g in engine_groups(eng)) > 10
    # the webhook feed of the merge pass (send_merge_webhooks, mod.rs:974-1000): per merge, destroyed for every solo
    # group of the batch in batch order, then created for the merged group
    eng.enable_group_events()
    st.drain_events()
    n_merged = eng.merge_solo_groups()
    assert st.try_merge_solo_groups() == n_merged > 0
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    ev = eng.drain_group_events()
    assert ev == st.drain_events()
    assert sum(k == E.GROUP_CREATED for k, *_ in ev) == n_merged
    assert sum(k == E.GROUP_DESTROYED for k, *_ in ev) == sum(len(m) for k, _i, _c, m in ev if k == E.GROUP_CREATED)
    assert eng.drain_group_events() == []                                  # drained
    eng.close()


@pytest.mark.parametrize("stream_min", [None, "8"])
def test_merge_policies_and_host_resolve(stream_min, monkeypatch):
    if stream_min:
        monkeypatch.setenv("PM_MERGE_STREAM_MIN", stream_min)
    sw, st, eng = _solo_scenario(2, switching=False)
    assert st.try_merge_solo_groups() == eng.merge_solo_groups() == 0
    eng.close()
    # prefer_larger_groups = false: a batch containing a group that already holds a task is refused
    # (mod.rs:277-287).  Only "solo-only" groups can claim a task here, the others stay idle and merge.
    sw, st, eng = _solo_scenario(2, tasks_only_for=2, prefer_larger=False)
    t_o = [st.get_task_for_node(w) for w in range(sw.W)]
    t_e, _ = eng.match()
    assert t_o == [(-1 if t == NONE else int(t)) for t in t_e] and 0 < sum(t >= 0 for t in t_o) < sw.W
    assert st.try_merge_solo_groups() == eng.merge_solo_groups()
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    eng.close()
    sw, st, eng = _solo_scenario(3, debug_uncertain_every=2)
    assert st.try_merge_solo_groups() == eng.merge_solo_groups() > 0
    assert eng.last_stats()["host_resolved_steps"] > 0
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    eng.close()


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("chooser", [E.CHOOSE_FIRST, E.CHOOSE_SEEDED])
def test_match_reference_orientation_bit_exact(variant, chooser):
    """NodeGroupsPlugin::filter_tasks for every worker (scheduler_impl.rs:11-110) incl. GROUP_INDEX,
    GROUP_SIZE and the NEXT_P2P_ADDRESS worker, against one oracle call per node."""
    sw = make_swarm(6, 20000, 3000)
    kw = dict(chooser=chooser, chooser_seed=99)
    st = oracle_state_for(sw, reference_shaped=False, **kw)
    eng = E.Engine(sweep_variant=variant, **kw)
    host.load_swarm(eng, sw)
    st.try_form_new_groups()
    eng.form_groups()
    task, count = eng.match()
    nodes, cfgs, tasks, _ = orc.from_swarm(sw)
    cfg_of_node = np.full(sw.W, -1, dtype=np.int32)
    for _, _, cfg, mem, _ in st.groups():
        cfg_of_node[mem] = cfg
    first_o, count_o = orc.pair_sweep_per_worker(tasks, cfgs, cfg_of_node)
    assert np.array_equal(count, count_o)
    for w in range(sw.W):
        t, gi, gs, nxt = st.filter_tasks(w)
        a = eng.lookup(w)
        assert (-1 if a.task == NONE else a.task) == t == (-1 if task[w] == NONE else int(task[w])), w
        if t >= 0:
            assert (a.group_index, a.group_size, a.next_worker) == (gi, gs, nxt), w
    if chooser == E.CHOOSE_FIRST:
        assert np.array_equal(task, first_o)
    # the claim is sticky (SETNX): a second match returns the same table
    assert np.array_equal(eng.match()[0], task)
    assert oracle_groups(st) == engine_groups(eng)
    eng.close()


@pytest.mark.parametrize("variant", [0, 1])
def test_match_per_task_orientation(variant):
    """north_star orientation: per task, the first eligible compatible worker and the candidate count."""
    sw = make_swarm(8, 5000, 4000)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    eng = E.Engine(sweep_variant=variant)
    host.load_swarm(eng, sw)
    masks = orc.compat_masks(nodes, cfgs)
    elig = (sw.status == 2) & sw.has_p2p
    col = np.where(elig, masks & np.uint64(sw.enabled_mask()), np.uint64(0))
    tm = sw.task_masks()
    best, count = eng.match_per_task()
    for t in range(0, sw.T, 7):
        hit = np.nonzero(col & tm[t])[0]
        assert count[t] == len(hit) and best[t] == (hit[0] if len(hit) else NONE), t
    # after carving, assigned workers stop being candidates (mod.rs:492-497)
    eng.form_groups()
    gow = eng.get_groups()[0]
    col2 = np.where(gow < 0, col, np.uint64(0))
    best2, count2 = eng.match_per_task()
    for t in range(0, sw.T, 13):
        hit = np.nonzero(col2 & tm[t])[0]
        assert count2[t] == len(hit) and best2[t] == (hit[0] if len(hit) else NONE), t
    eng.close()


def test_match_per_task_with_prices():
    """The extension column: with non-zero prices the best bid of a task is min (price, worker index) over its
    candidates (include/pm_engine.h, pm_match_per_task); counts do not depend on prices."""
    sw = make_swarm(8, 3000, 2500)
    rng = np.random.default_rng(4)
    sw.price[:] = rng.integers(0, 40, sw.W).astype(np.uint32)      # many equal prices: the index breaks ties
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    masks = orc.compat_masks(nodes, cfgs)
    elig = (sw.status == 2) & sw.has_p2p
    col = np.where(elig, masks & np.uint64(sw.enabled_mask()), np.uint64(0))
    tm = sw.task_masks()
    order = np.lexsort((np.arange(sw.W), sw.price))                  # by (price, index)
    for variant in (0, 1):
        eng = E.Engine(sweep_variant=variant)
        host.load_swarm(eng, sw)
        best, count = eng.match_per_task()
        for t in range(0, sw.T, 5):
            hit = (col[order] & tm[t]) != 0
            assert count[t] == int(hit.sum()), t
            assert best[t] == (order[np.argmax(hit)] if hit.any() else NONE), t
        eng.close()


def test_lookups_from_other_threads_during_ticks():
    """filter_tasks is called concurrently from the HTTP workers while the management loop runs
    (plugins/mod.rs:66-78, tests.rs:642-645).  Eight threads hammer pm_lookup_task_for_worker while the main
    thread runs 120 ticks with churn in between; every row a reader ever saw must be, field for field, a row some
    tick published for that worker (no torn rows, no rows of a half-written table)."""
    import threading
    sw = make_swarm(14, 3000, 1200)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    flags = host.worker_flags(sw).astype(np.int64)
    eng.tick()
    row = lambda a: (a.task, a.group_slot, a.group_index, a.group_size, a.next_worker, a.group_id)
    published = [set() for _ in range(sw.W)]

    def snapshot():
        for w in range(sw.W):
            published[w].add(row(eng.lookup(w)))

    snapshot()
    stop = threading.Event()
    seen = [[] for _ in range(8)]

    def reader(k):
        rng = np.random.default_rng(100 + k)
        ws = rng.integers(0, sw.W, 4096)
        i = 0
        while not stop.is_set():
            w = int(ws[i & 4095])
            seen[k].append((w, row(eng.lookup(w))))
            i += 1

    threads = [threading.Thread(target=reader, args=(k,)) for k in range(8)]
    for t in threads:
        t.start()
    rng = np.random.default_rng(5)
    healthy = np.nonzero(sw.status == 2)[0]
    try:
        for tick in range(60):
            victims = rng.choice(healthy, size=6, replace=False)        # a death dissolves the whole group ...
            for w in victims:
                eng.on_worker_status(int(w), int(flags[w] & ~E.W_HEALTHY), True)
            eng.tick()
            snapshot()
            for w in victims:                                           # ... and the rejoin re-carves the leftovers
                eng.on_worker_status(int(w), int(flags[w]), False)
            eng.tick()
            snapshot()
    finally:   # (a tick that fails must not leave the readers running: the suite would never end)
        stop.set()
        for t in threads:
            t.join()
    n_reads = sum(len(s) for s in seen)
    assert n_reads > 10000
    # (a death between two ticks republishes the rows of the dissolved group's members as rows of workers in no group)
    no_group = (NONE, NONE, 0, 0, NONE, 0)
    for k in range(8):
        for w, r in seen[k]:
            assert r in published[w] or r == no_group, (w, r)
    assert max(len(p) for p in published) > 1          # the tables really changed under the readers
    eng.close()


def test_newest_task_plugin():
    eng = E.Engine()
    sw = make_swarm(3, 100000, 16)
    host.load_swarm(eng, sw)
    nodes, cfgs, tasks, _ = orc.from_swarm(sw)
    assert eng.newest_task() == orc.newest_task(tasks)
    # unsorted input + duplicates of the maximum: the LAST maximum wins (Iterator::max_by_key)
    ca = np.array([5, 9, 1, 9, 9, 3] * 1000, dtype=np.int64)
    eng.upload_tasks(np.full(len(ca), 2 ** 64 - 1, dtype=np.uint64), ca)
    assert eng.newest_task() == len(ca) - 2
    eng.upload_tasks(np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.int64))
    assert eng.newest_task() == NONE
    eng.close()


def test_tick_publishes_table_and_tasks_follow_uid():
    sw = make_swarm(12, 4000, 1500)
    st = oracle_state_for(sw, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    with pytest.raises(E.EngineError):
        eng.lookup(0)                                   # nothing published yet
    s = eng.tick()
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    want = [st.get_task_for_node(w) for w in range(sw.W)]
    got = [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
    assert got == want and s["pair_evals"] == sw.T * sw.W and s["n_groups"] == st.n_groups
    # a newer task is inserted at the front: claimed tasks keep their identity (uid), indices shift by one
    eng.upload_tasks(np.concatenate([[np.uint64(2 ** 64 - 1)], sw.task_masks()]),
                     np.concatenate([[sw.created_at[0] + 1], sw.created_at]),
                     np.concatenate([[np.uint64(12345)], sw.task_uid]))
    eng.tick()
    got2 = [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
    assert got2 == [(t + 1 if t >= 0 else -1) for t in want]
    # deleting a claimed task dissolves its groups (on_task_deleted, mod.rs:1259-1288)
    victim = max(set(t for t in want if t >= 0), key=want.count)
    keep = np.arange(sw.T) != victim
    n_before = len(engine_groups(eng))
    eng.upload_tasks(sw.task_masks()[keep], sw.created_at[keep], sw.task_uid[keep])
    assert len(engine_groups(eng)) < n_before
    eng.close()


def test_engine_argument_errors():
    eng = E.Engine()
    with pytest.raises(E.EngineError) as ei:
        eng.tick()
    assert ei.value.code == E.PM_ESTATE
    bad = np.zeros(1, dtype=E.config_row_dt)
    bad["min_group_size"], bad["max_group_size"] = 3, 2
    with pytest.raises(E.EngineError) as ei:
        eng.set_configs(bad, np.zeros(0, dtype=E.alt_row_dt))          # mod.rs:145-147 panics
    assert ei.value.code == E.PM_EINVAL
    bad["min_group_size"], bad["max_group_size"] = 0, 2
    with pytest.raises(E.EngineError):
        eng.set_configs(bad, np.zeros(0, dtype=E.alt_row_dt))
    eng.close()


def test_task_observer_flow():
    """tests.rs:679-731 and 1467-1628 through the engine: while no task lists a configuration nothing is enabled and
    nothing forms; tasks arrive (on_task_created enables their topologies) and the groups form; deleting the task a
    group works on dissolves it at once (on_task_deleted).  The enabled set is the shim's job
    (rust/gpu_match_plugin.rs push_enabled = protocol_amd.swarm.Swarm.enabled_mask): here it is set by hand."""
    sw = make_swarm(12, 60, 400)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    none = np.zeros(len(cfgs), dtype=np.uint8)
    st = orc.State(nodes, cfgs, enabled=none, tasks=tasks[:0], reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    eng.upload_tasks(masks[:0], created[:0], uid[:0])
    eng.set_enabled_mask(0)
    s = eng.tick()
    assert s["n_groups"] == 0 == st.try_form_new_groups()
    assert all(eng.lookup(w).task == NONE for w in range(0, sw.W, 7))
    # the tasks arrive (in front of an empty list), their topologies get enabled
    eng.tasks_insert_front(masks, created, uid)
    eng.set_enabled_mask(sw.enabled_mask())
    st.set_tasks(tasks)
    st.set_enabled(enabled)
    eng.tick()
    assert st.try_form_new_groups() > 20
    st.try_merge_solo_groups()
    want = [st.get_task_for_node(w) for w in range(sw.W)]
    got = [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
    assert got == want
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    # delete the task most groups work on
    claimed = [g[4] for g in st.groups() if g[4] >= 0]
    victim = max(set(claimed), key=claimed.count)
    keep = np.ones(len(tasks), dtype=bool)
    keep[victim] = False
    assert eng.tasks_delete(uid[victim:victim + 1]) == 1
    st.set_tasks(tasks[keep])
    st.remap_tasks(np.where(keep, np.cumsum(keep) - 1, -1))
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))          # dissolved at once, before any tick
    assert len(engine_groups(eng)) < len(claimed)
    eng.close()


def test_lookups_between_ticks_follow_task_deltas_and_deaths():
    """A heartbeat between two ticks (scheduler/mod.rs:26-36 between two runs of the management loop): the reference
    binds a group to its task by id (get_current_group_task, scheduler_impl.rs:33), so after new tasks arrived in
    front of the list every grouped worker still gets ITS task (under its new position), after a claimed task was
    deleted its groups are gone at once (mod.rs:1259-1288), after a worker died its whole group is
    (status_update_impl.rs:17-29) — all without a tick in between.  pm_lookup_task_for_worker must say the same."""
    sw = make_swarm(14, 300, 1200)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.tick()
    st.try_form_new_groups()
    st.try_merge_solo_groups()

    def check():
        want = [st.get_task_for_node(w) for w in range(sw.W)]
        got = [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
        assert got == want
        in_group = st.node_to_group >= 0
        assert [eng.lookup(w).group_slot != NONE for w in range(sw.W)] == list(in_group)

    check()
    assert sum(1 for w in range(sw.W) if eng.lookup(w).task != NONE) > 100
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    # ---- 25 new tasks in front of the list: positions shift, bindings stay
    n_new = 25
    new_masks = masks[:n_new].copy()
    new_created = created.max() + np.arange(n_new, 0, -1).astype(created.dtype)
    new_uid = (np.arange(n_new) + (1 << 41)).astype(uid.dtype)
    eng.tasks_insert_front(new_masks, new_created, new_uid)
    new_rows = tasks[:n_new].copy()
    new_rows["created_at"] = new_created
    old_n = len(tasks)
    tasks = np.concatenate([new_rows, tasks])
    st.set_tasks(tasks)
    st.remap_tasks(np.arange(old_n) + n_new)
    masks, created, uid = np.concatenate([new_masks, masks]), np.concatenate([new_created, created]), np.concatenate([new_uid, uid])
    check()
    # ---- the task most groups work on is deleted: those groups dissolve, the tasks behind it move up
    claimed = [g[4] for g in st.groups() if g[4] >= 0]
    victim = max(set(claimed), key=claimed.count)
    keep = np.ones(len(tasks), dtype=bool)
    keep[victim] = False
    assert eng.tasks_delete(uid[victim:victim + 1]) == 1
    st.set_tasks(tasks[keep])
    st.remap_tasks(np.where(keep, np.cumsum(keep) - 1, -1))
    tasks, masks, created, uid = tasks[keep], masks[keep], created[keep], uid[keep]
    check()
    # ---- a grouped worker dies: its group is gone for every member
    flags = host.worker_flags(sw).astype(np.int64)
    victim_w = int(np.nonzero(st.node_to_group >= 0)[0][5])
    st.set_node_status(victim_w, orc.ST_DEAD)
    eng.on_worker_status(victim_w, int(flags[victim_w]) & ~E.W_HEALTHY, True)
    check()
    # ---- and the next tick agrees with a fresh look at everything
    eng.tick()
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    check()
    eng.close()


def test_insert_front_with_republish_serves_idle_groups_at_once():
    """pm_tasks_insert_front_ex(rows, republish = 1): the reference offers a new task to a group that holds none at that
    group's next heartbeat (get_task_for_node: no current task -> pick among the applicable ones and claim,
    scheduler_impl.rs:33-74).  The plain insertion leaves such a group unserved until the next tick; with the
    republish flag the call re-matches the standing groups (pair sweep + claim + publish, no carve) and every worker is
    served what the oracle's heartbeat serves it — while groups that already hold a task keep it."""
    sw = make_swarm(19, 300, 1500)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks[:0], reference_shaped=False)
    eng = E.Engine()
    cfg_rows, alt_rows, req_models = host.pack_configs(sw.configs)
    eng.set_configs(cfg_rows, alt_rows)
    eng.set_model_table(host.build_model_table(req_models, sw.model_names), len(req_models), len(sw.model_names))
    eng.upload_workers(host.pack_workers(sw))
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    eng.upload_tasks(masks[:0], created[:0], uid[:0])                 # no task yet: every group is formed idle
    eng.set_enabled_mask(sw.enabled_mask())
    eng.tick()
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng)) and len(st.groups()) > 50
    served = lambda: [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
    assert set(served()) == {-1}
    cur = tasks[:0]
    t_next = int(created.max()) + 1

    def insert(rows_idx, republish):
        nonlocal cur, t_next
        n = len(rows_idx)
        c_new = (t_next + np.arange(n, 0, -1)).astype(created.dtype)
        t_next += n + 1
        eng.tasks_insert_front(masks[rows_idx], c_new, (uid[rows_idx] + np.uint64(t_next << 20)).astype(uid.dtype), republish=republish)
        new_rows = tasks[rows_idx].copy()
        new_rows["created_at"] = c_new
        old_n = len(cur)
        cur = np.concatenate([new_rows, cur])
        st.set_tasks(cur)
        st.remap_tasks(np.arange(old_n) + n)

    # ---- a few tasks of few topologies, plain insertion: nobody is served before the next tick ...
    few = np.nonzero(masks == masks[0])[0][:3]
    insert(few, republish=False)
    assert set(served()) == {-1}
    # ---- ... the same with the flag: every group the oracle's heartbeat serves is served, the rest stay idle
    insert(np.nonzero(masks == masks[1])[0][:3], republish=True)
    want = [st.get_task_for_node(w) for w in range(sw.W)]
    assert served() == want
    n_first = sum(t >= 0 for t in want)
    assert 0 < n_first
    # ---- more tasks: groups that hold a task keep it (under its new position), idle ones are offered the new list
    insert(np.arange(40, 80), republish=True)
    want = [st.get_task_for_node(w) for w in range(sw.W)]
    assert served() == want
    assert sum(t >= 0 for t in want) > n_first
    # ---- and the next tick agrees
    eng.tick()
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    assert served() == [st.get_task_for_node(w) for w in range(sw.W)]
    eng.close()


def test_pools_share_one_gpu():
    """pm_set_carve_workgroups: four engines (four pools, their own swarms) matched at the same time from four host
    threads, each with a quarter of the CUs for its carve — and once more with a share of ONE row-making workgroup, the
    smallest launch there is.  Whatever the share, and whatever the other pools are doing on the GPU meanwhile, every
    pool's groups are the oracle's."""
    import threading
    K = 4
    pools = []
    for k in range(K):
        sw = make_swarm(50 + k, 400, 3000 + 500 * k)
        eng = E.Engine(group_id_seed=7 + k)
        host.load_swarm(eng, sw)
        pools.append((eng, sw))
    want = []
    for k, (eng, sw) in enumerate(pools):
        st = oracle_state_for(sw, reference_shaped=False, group_id_seed=7 + k)
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        want.append(oracle_groups(st))
    for share in (60, 1, 0):
        for eng, _ in pools:
            eng.set_carve_workgroups(share)
        got = [None] * K
        go = threading.Barrier(K)

        def run(i):
            eng = pools[i][0]
            go.wait()
            for _ in range(3):
                eng.reset_groups()
                eng.tick()
            got[i] = engine_groups(eng)

        th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        # (group ids continue each engine's own stream across the repeats: compare configurations and members)
        strip = lambda gs: [(c, m) for (_gid, c, m, _task) in gs]
        for i in range(K):
            assert strip(got[i]) == strip(want[i]), (share, i)
            assert pools[i][0].debug_carve_counters()["stream"] == 1
    for eng, _ in pools:
        eng.close()


def test_tick_many_is_pm_tick_per_engine():
    """pm_tick_many: four pools matched in ONE call (carves started before the first is waited for, one host thread; and
    with the library's thread-per-engine variant) end where four separate pm_tick calls end: the oracle's groups, the same
    group ids, and row for row the same published table.  An engine that cannot tick fails the call and leaves the others'
    results standing."""
    K = 4
    pools = []
    for k in range(K):
        sw = make_swarm(70 + k, 300 + 40 * k, 2500 + 700 * k)
        eng = E.Engine(group_id_seed=11 + k)
        host.load_swarm(eng, sw)
        pools.append((eng, sw))
    engines = [eng for eng, _ in pools]
    row = lambda a: (a.task, a.group_slot, a.group_index, a.group_size, a.next_worker, a.group_id)
    table = lambda eng, sw: [row(eng.lookup(w)) for w in range(sw.W)]
    solo = []
    for k, (eng, sw) in enumerate(pools):
        st = oracle_state_for(sw, reference_shaped=False, group_id_seed=11 + k)
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        s = eng.tick()
        want = [st.get_task_for_node(w) for w in range(sw.W)]   # (a group claims its task at its first heartbeat)
        assert [eng.lookup(w).task if eng.lookup(w).task != NONE else -1 for w in range(sw.W)] == want
        assert sorted(engine_groups(eng)) == sorted(oracle_groups(st))
        solo.append((engine_groups(eng), table(eng, sw), s))
    for threads in (False, True):
        for share in (60, 0):
            for eng in engines:
                eng.reset_groups()
                eng.set_carve_workgroups(share)
            stats = E.tick_many(engines, threads=threads)
            assert len(stats) == K
            for i, (eng, sw) in enumerate(pools):
                assert engine_groups(eng) == solo[i][0], (threads, share, i)
                assert table(eng, sw) == solo[i][1], (threads, share, i)
                for key in ("n_groups", "n_formed", "n_merged", "pair_evals"):
                    assert stats[i][key] == solo[i][2][key], (key, threads, share, i)
                assert stats[i]["pair_evals"] == sw.T * sw.W and stats[i]["ms_total"] > 0.0
                assert eng.last_stats()["n_groups"] == solo[i][2]["n_groups"]
            # an incremental batch on the standing groups: nothing forms, nothing moves
            stats = E.tick_many(engines, threads=threads)
            for i, (eng, sw) in enumerate(pools):
                assert stats[i]["n_formed"] == 0 and table(eng, sw) == solo[i][1]
    # a subset, in another order
    for eng in engines:
        eng.reset_groups()
    stats = E.tick_many([engines[2], engines[0]])
    assert [s["n_groups"] for s in stats] == [solo[2][2]["n_groups"], solo[0][2]["n_groups"]]
    assert table(*pools[2]) == solo[2][1] and table(*pools[0]) == solo[0][1]
    assert all(engines[1].lookup(w).group_slot == NONE for w in range(pools[1][1].W))  # (not in the batch: still reset)
    # argument checks; an engine without tasks fails the call before any engine is touched
    bare = E.Engine()
    with pytest.raises(E.EngineError):
        E.tick_many([engines[0], engines[0]])
    with pytest.raises(E.EngineError):
        E.tick_many([engines[1], bare])
    assert all(engines[1].lookup(w).group_slot == NONE for w in range(pools[1][1].W))
    with pytest.raises(E.EngineError):
        E.tick_many([engines[1], bare], threads=True)   # (per-engine pm_tick: engine 1 ticks, the bare one reports)
    assert table(*pools[1]) == solo[1][1]
    bare.close()
    for eng in engines:
        eng.close()


def test_lookups_after_reset_groups_say_no_group():
    """pm_reset_groups (and pm_upload_workers without keep_groups) drops every group and restarts the id stream: a
    heartbeat before the next publish must not be served a row that names a slot or an id of the old list, and a task
    delta in between must not resolve such a row against the new one."""
    sw = make_swarm(23, 200, 900)
    eng = E.Engine(group_id_seed=5)
    host.load_swarm(eng, sw)
    eng.tick()
    grouped = [w for w in range(sw.W) if eng.lookup(w).group_slot != NONE]
    assert len(grouped) > 100
    eng.reset_groups()
    for w in range(sw.W):
        a = eng.lookup(w)
        assert a.group_slot == NONE and a.task == NONE and a.group_size == 0
    # new groups with the SAME ids (the id stream restarted), not yet published; a task delta patches the published rows
    eng.form_groups()
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    eng.tasks_insert_front(masks[:3], created.max() + np.arange(3, 0, -1).astype(created.dtype), (uid[:3] + np.uint64(1 << 40)).astype(uid.dtype))
    assert all(eng.lookup(w).group_slot == NONE for w in range(sw.W))
    eng.tick()
    assert sum(eng.lookup(w).group_slot != NONE for w in range(sw.W)) == len(grouped)
    eng.close()


def test_group_event_feed_semantics():
    """off by default; a drain with buffers that are too small reports the sizes and drains nothing; switching the
    feed off clears it; pm_reset_groups logs nothing"""
    import ctypes as C
    sw = baseline_config(0, seed=3)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.tick()
    assert eng.drain_group_events() == []                          # not enabled: nothing was logged
    eng.reset_groups()
    eng.enable_group_events()
    eng.form_groups()
    groups = engine_groups(eng)
    ne, nm = C.c_uint32(0), C.c_uint32(0)
    one = np.zeros(1, dtype=E.GROUP_EVENT)
    rc = E.lib().pm_drain_group_events(eng._h, one.ctypes.data, 1, None, 0, C.byref(ne), C.byref(nm))
    assert rc == E.PM_ERANGE and ne.value == len(groups) and nm.value == sum(len(g[2]) for g in groups)
    ev = eng.drain_group_events()                                  # still all there
    assert [(i, c, m) for k, i, c, m in ev] == [(i, c, m) for i, c, m, _t in groups]
    assert all(k == E.GROUP_CREATED for k, *_ in ev)
    eng.reset_groups()                                             # (bench helper: no counterpart, no events)
    assert eng.drain_group_events() == []
    eng.tick()
    eng.enable_group_events(False)
    eng.enable_group_events(True)
    assert eng.drain_group_events() == []                          # switching off cleared the log
    eng.close()


@pytest.mark.parametrize("deltas", [False, True])
def test_streaming_churn_parity(deltas):
    """(deltas: the task table is maintained with pm_tasks_insert_front / pm_tasks_delete instead of re-uploading
    the snapshot every tick — same results, and claimed tasks keep their binding without any look-up)
    BASELINE configs[4] in miniature: every tick new tasks arrive (newest first), ~1 % of the workers die
    (whole group dissolved) or join, one old task is deleted (its groups dissolve); the engine's incremental
    state must track the oracle's tick by tick — existing groups are sticky (Appendix A)."""
    rng = np.random.default_rng(42)
    sw = make_swarm(9, 1500, 3000)
    status0 = sw.status.copy()
    late = rng.random(sw.W) < 0.25
    sw.status = np.where(late, 0, sw.status).astype(np.uint8)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.enable_group_events()
    flags = host.worker_flags(sw).astype(np.int64)
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    joiners = list(np.nonzero(late & (status0 == 2))[0])
    next_uid = 1 << 40
    n_events = [0, 0]
    for tick in range(6):
        # ---- task churn: 40 new tasks (newest => front of get_all_tasks), one claimed task deleted from tick 2 on
        n_new = 40
        pick = rng.integers(0, len(tasks), n_new)
        new_rows = tasks[pick].copy()
        new_rows["created_at"] = int(created.max()) + 1 + np.arange(n_new)[::-1]
        keep = np.ones(len(tasks), dtype=bool)
        if tick >= 2:
            claimed = [g[4] for g in st.groups() if g[4] >= 0]
            if claimed:
                keep[claimed[tick % len(claimed)]] = False
        old_to_new = np.where(keep, n_new + np.cumsum(keep) - 1, -1)
        uid_prev = uid
        tasks = np.concatenate([new_rows, tasks[keep]])
        masks = np.concatenate([masks[pick], masks[keep]])
        created = np.concatenate([new_rows["created_at"], created[keep]])
        uid = np.concatenate([np.arange(next_uid, next_uid + n_new, dtype=np.uint64), uid[keep]])
        next_uid += n_new
        st.set_tasks(tasks)
        st.remap_tasks(old_to_new)
        if deltas:
            if not keep.all():
                assert eng.tasks_delete(uid_prev[~keep]) == int((~keep).sum())
            eng.tasks_insert_front(masks[:n_new], created[:n_new], uid[:n_new])
            assert eng.T == len(masks)
        else:
            eng.upload_tasks(masks, created, uid)
        # ---- worker churn: ~0.5 % die, ~0.5 % join
        alive = np.nonzero(st.nodes["status"] == 2)[0]
        for w in rng.choice(alive, size=15, replace=False):
            st.set_node_status(int(w), 4)
            flags[w] &= ~E.W_HEALTHY
            eng.on_worker_status(int(w), int(flags[w]), True)
        for _ in range(15):
            if joiners:
                w = int(joiners.pop())
                st.set_node_status(w, 2)
                flags[w] |= E.W_HEALTHY
                eng.on_worker_status(w, int(flags[w]), False)
        # ---- one management tick + every worker's scheduling
        s = eng.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        want = [st.get_task_for_node(w) for w in range(sw.W)]
        got = [(-1 if eng.lookup(w).task == NONE else eng.lookup(w).task) for w in range(sw.W)]
        assert got == want, f"tick {tick}"
        assert sorted(oracle_groups(st)) == sorted(engine_groups(eng)), f"tick {tick}"
        assert s["host_resolved_steps"] == 0
        # the webhook feed of the tick, in the reference's emission order: dissolutions by the deleted task and by
        # the deaths, then the creations of try_form_new_groups, then the merges (mod.rs:612-625, 974-1000, 1469-1481)
        ev = eng.drain_group_events()
        assert ev == st.drain_events(), f"tick {tick}"
        n_events[0] += sum(k == E.GROUP_CREATED for k, *_ in ev)
        n_events[1] += sum(k == E.GROUP_DESTROYED for k, *_ in ev)
    assert n_events[0] > 100 and n_events[1] > 20
    eng.close()


def test_task_deltas_equal_a_fresh_upload():
    """pm_tasks_insert_front / pm_tasks_delete against an engine that re-uploads the whole snapshot: the same
    published table (task positions in the CURRENT list), the same groups, the same per-task bids and newest task —
    through several rounds, including a growth of the table's index space and deletions at the front."""
    rng = np.random.default_rng(77)
    sw = make_swarm(15, 3000, 1500)
    a, b = E.Engine(), E.Engine()
    host.load_swarm(a, sw)
    host.load_swarm(b, sw)
    masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
    next_uid = 1 << 41
    with pytest.raises(E.EngineError):                       # not newer than the newest task: not a front insertion
        a.tasks_insert_front(masks[:1], created[:1], np.array([7], dtype=np.uint64))
    for rnd in range(6):
        a.tick()
        b.tick()
        ta = [a.lookup(w).task for w in range(sw.W)]
        assert ta == [b.lookup(w).task for w in range(sw.W)], rnd
        assert engine_groups(a) == engine_groups(b), rnd
        assert a.newest_task() == b.newest_task()
        ba, ca = a.match_per_task()
        bb, cb = b.match_per_task()
        assert np.array_equal(ba, bb) and np.array_equal(ca, cb), rnd
        # ---- delta: delete a few tasks (some claimed, the two in front), insert a batch of newer ones
        claimed = sorted(set(t for t in ta if t != NONE))
        kill = set(int(x) for x in rng.choice(len(masks), size=25, replace=False)) | {0, 1} | set(claimed[:3])
        keep = np.ones(len(masks), dtype=bool)
        keep[list(kill)] = False
        n_new = 70000 if rnd == 2 else 40                     # round 2 outgrows the index space of the table
        pick = rng.integers(0, len(masks), n_new)
        new_masks = masks[pick]
        new_created = int(created.max()) + 1 + np.arange(n_new)[::-1]
        new_uid = np.arange(next_uid, next_uid + n_new, dtype=np.uint64)
        next_uid += n_new
        assert a.tasks_delete(uid[~keep]) == int((~keep).sum())
        assert a.tasks_delete(uid[~keep]) == 0               # unknown ids are ignored
        a.tasks_insert_front(new_masks, new_created, new_uid)
        masks = np.concatenate([new_masks, masks[keep]])
        created = np.concatenate([new_created, created[keep]])
        uid = np.concatenate([new_uid, uid[keep]])
        b.upload_tasks(masks, created, uid)
        assert a.T == b.T == len(masks)
    a.close()
    b.close()


@pytest.mark.parametrize("env", [{"PM_STREAM_LA": "24"}, {"PM_STREAM_WGS": "1"}, {"PM_STREAM_ROW_SPINS": "1", "PM_STREAM_WGS": "2"},
                                 {"PM_STREAM_LA": "1024", "PM_STREAM_LA_DIV": "16"}])
def test_streaming_carve_under_pressure(env, monkeypatch):
    """The streaming carve (carve_variant 0) with its knobs turned the wrong way: a look-ahead of two dozen tickets, a
    single proposer workgroup, a validator that gives a row up after one poll (every such seed is a step for the exact
    sweep), a window far too wide (rows run out of live entries: exact steps, refreshes).  None of it may change a group."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for sw in (baseline_config(0, seed=3), make_swarm(31, 2000, 3000), baseline_config(1, seed=2)):
        st = oracle_state_for(sw, reference_shaped=(sw.W <= 2048), group_id_seed=7)
        eng = E.Engine(group_id_seed=7)
        host.load_swarm(eng, sw)
        assert st.try_form_new_groups() == eng.form_groups()
        assert oracle_groups(st) == engine_groups(eng), env
        c = eng.debug_carve_counters()
        assert c["stream"] == 1 and c["stream_aborts"] == 0, c
        eng.close()


@pytest.mark.parametrize("after", [1, 40, 300, 1500])
def test_streaming_carve_abort_falls_back_to_the_batch_pipeline(after):
    """CARVE_STATE_ABORTED — what a lost hand-shake inside the validator workgroup (proposers that cannot run beside it: a
    shared GPU) ends a streaming launch in — forced by the debug hook after `after` committed steps, early, in the
    middle of a large configuration and late.  What was committed stands, and the engine continues from the
    configuration the launch stopped in on the batch pipeline (pm_engine.cpp form_poll): a different kernel family,
    a rebuilt eligible list, per-batch candidate lists.  The groups are the oracle's all the same."""
    for sw in (baseline_config(1, seed=2), make_swarm(31, 2000, 3000), baseline_config(0, seed=3)):
        st = oracle_state_for(sw, reference_shaped=(sw.W <= 2048), group_id_seed=7)
        eng = E.Engine(group_id_seed=7)
        host.load_swarm(eng, sw)
        eng.debug_stream_abort_after(after)
        n_want = st.try_form_new_groups()
        assert n_want == eng.form_groups()
        assert oracle_groups(st) == engine_groups(eng), (after, sw.W)
        c = eng.debug_carve_counters()
        # (a carve with fewer steps through the chain than `after` never reaches the hook)
        assert c["stream_aborts"] in (0, 1), c
        if c["stream_aborts"]:
            assert c["stream"] == 0 and c["batches"] >= 1, c  # the rest of the carve ran as batches
        elif sw.W >= 10000:
            assert after >= 1500, c  # configs[1]'s first configuration commits 300 groups through the chain
        # the engine streams again afterwards: the hook off, the groups dissolved, the same carve once more
        eng.debug_stream_abort_after(0)
        eng.reset_groups()
        st2 = oracle_state_for(sw, reference_shaped=(sw.W <= 2048), group_id_seed=7)
        assert st2.try_form_new_groups() == eng.form_groups()
        c2 = eng.debug_carve_counters()
        assert c2["stream"] == 1 and c2["stream_aborts"] == 0, c2
        assert [g[1:] for g in oracle_groups(st2)] == [g[1:] for g in engine_groups(eng)]
        eng.close()
