"""Full-size BASELINE configurations through size-independent properties (the oracle cannot run 1e11 pairs),
plus mid-size bit-exact parity that exercises the big-list (> 8192 candidate) code paths."""
import numpy as np
import pytest

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import baseline_config, make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


@pytest.mark.parametrize("carve_variant", [0, 1, 3])
def test_form_groups_big_lists_bit_exact(carve_variant):
    """30k workers: several configurations have > 8192 candidates (big-list mode, proposal batches of 16384)."""
    sw = make_swarm(2, 2000, 30000, zipf=True)
    st = oracle_state_for(sw, reference_shaped=False)
    eng = E.Engine(carve_variant=carve_variant)
    host.load_swarm(eng, sw)
    assert st.try_form_new_groups() == eng.form_groups()
    assert oracle_groups(st) == engine_groups(eng)
    assert eng.last_stats()["host_resolved_steps"] == 0
    eng.close()


def _check_invariants(sw, eng, stats):
    gow, groups, members = eng.get_groups()
    masks = eng.compat_masks()
    # every worker in at most one group; group_of is consistent with the member lists (tests.rs:993-1213)
    seen = np.zeros(sw.W, dtype=np.int32)
    np.add.at(seen, members, 1)
    assert seen.max() <= 1 and int(seen.sum()) == len(members)
    for g in np.random.default_rng(0).choice(len(groups), size=min(2000, len(groups)), replace=False):
        b, n, c = int(groups[g]["member_begin"]), int(groups[g]["n_members"]), int(groups[g]["config"])
        mem = members[b:b + n]
        mn, mx = sw.configs[c][1], sw.configs[c][2]
        assert mn <= n <= mx
        assert (gow[mem] == g).all()
        assert ((masks[mem] >> np.uint64(c)) & np.uint64(1)).all()          # only compatible members
    # only eligible workers were grouped (mod.rs:492-497)
    elig = (sw.status == 2) & sw.has_p2p
    assert not (gow[~elig] >= 0).any()
    # greedy exhaustion: no configuration can still form a group from the leftovers (mod.rs:507-519)
    left = elig & (gow < 0)
    for c, (_n, mn, _mx, _r) in enumerate(sw.configs):
        if (sw.enabled_mask() >> c) & 1 and int(left.sum()) >= mn:
            assert int((left & (((masks >> np.uint64(c)) & np.uint64(1)) != 0)).sum()) < mn, c
    assert stats["n_groups"] == len(groups) and stats["pair_evals"] == sw.T * sw.W


def _check_tasks(sw, eng):
    """per-worker task table against a direct evaluation of the topology predicate on a sample"""
    gow, groups, _ = eng.get_groups()
    tm = sw.task_masks()
    rng = np.random.default_rng(1)
    for w in rng.choice(sw.W, size=300, replace=False):
        a = eng.lookup(int(w))
        if gow[w] < 0:
            assert a.task == NONE
            continue
        bit = np.uint64(1) << np.uint64(int(groups[gow[w]]["config"]))
        hit = np.nonzero(tm & bit)[0]
        assert a.task == (hit[0] if len(hit) else NONE)
        assert a.group_size == int(groups[gow[w]]["n_members"])


def _sha(*arrays):
    import hashlib
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _engine_digests(sw, eng):
    """the digests of tools/make_golden_scale.py, computed from what the engine published"""
    _, groups, members = eng.get_groups()
    gsha = _sha(groups["id"].astype(np.uint64), groups["config"].astype(np.uint32),
                groups["n_members"].astype(np.uint32), members.astype(np.uint32))
    tbl = np.zeros(sw.W, dtype=E.assignment_dt)
    for w in range(sw.W):
        a = eng.lookup(w)
        tbl[w] = (a.task, a.group_slot, a.group_index, a.group_size, a.next_worker, a.group_id)
    return gsha, tbl


def _check_against_golden(name, carve_variant):
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scale_digests.json")))[name]
    sw = baseline_config(gold["config"], seed=gold["seed"])
    assert (sw.W, sw.T) == (gold["W"], gold["T"])
    eng = E.Engine(carve_variant=carve_variant, group_id_seed=gold["seed"])
    host.load_swarm(eng, sw)
    stats = eng.tick()
    task, count = eng.match()                  # the claim is sticky: the same table, plus the applicable counts
    assert stats["n_formed"] == gold["n_formed"] and stats["n_merged"] == gold["n_merged"]
    got = engine_groups(eng)
    assert len(got) == gold["n_groups"]
    # the explicit head and tail first: they localise a mismatch before the digest just says "different"
    assert [[g[0], g[1], g[2]] for g in got[:1000]] == gold["first_groups"]
    assert [[g[0], g[1], g[2]] for g in got[-1000:]] == gold["last_groups"]
    gsha, tbl = _engine_digests(sw, eng)
    assert gsha == gold["groups_sha256"], "group list (ids, configurations, sizes, members) differs from the oracle"
    assert np.array_equal(tbl["task"], task)
    assert _sha(tbl["task"].astype(np.uint32)) == gold["task_sha256"], "per-worker task column differs"
    assert _sha(count.astype(np.uint32)) == gold["count_sha256"], "applicable-task counts differ"
    assert _sha(tbl["group_index"].astype(np.uint32), tbl["group_size"].astype(np.uint32),
                tbl["next_worker"].astype(np.uint32)) == gold["table_sha256"], "GROUP_INDEX / SIZE / NEXT differ"
    assert stats["host_resolved_steps"] == 0
    eng.close()


@pytest.mark.parametrize("carve_variant", [0, 3])
def test_config1_full_size_against_oracle_digest(carve_variant):
    """BASELINE configs[1], every group and every worker's row against the oracle's committed digests
    (tests/golden/scale_digests.json, tools/make_golden_scale.py)."""
    _check_against_golden("cfg1_seed1", carve_variant)


def test_config2_full_size_against_oracle_digest():
    """BASELINE configs[2] (1M tasks x 100k workers, Zipf): big-list mode, PM_TIE_BAND_BIG and tenth-of-a-list
    proposal batches are all live here; the oracle needs ~1 min for the carve and 1e11 string compares for the
    sweep, so its result is committed as digests plus the first / last 1000 groups."""
    _check_against_golden("cfg2_seed1", 0)


def test_config2_seed2_full_size_against_oracle_digest():
    """a second 1M x 100k swarm (seed 2)"""
    _check_against_golden("cfg2_seed2", 0)


@pytest.mark.parametrize("name", ["cfg1_seed1_seeded", "cfg2_seed1_seeded"])
def test_seeded_chooser_against_oracle_digest(name):
    """BASELINE configs[1] and configs[2] with the SEEDED chooser (the injected stand-in for rand::rng().choose,
    scheduler_impl.rs:66-70; chooser_rank_kernel + the r-th-hit pass): groups and every worker's task against the
    digests of the oracle's own get_task_for_node (tests/golden/scale_digests.json)."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scale_digests.json"))).get(name)
    if gold is None:
        pytest.skip(f"{name} not in tests/golden/scale_digests.json (tools/make_golden_scale.py {name})")
    sw = baseline_config(gold["config"], seed=gold["seed"])
    eng = E.Engine(group_id_seed=gold["seed"], chooser=E.CHOOSE_SEEDED, chooser_seed=gold["chooser_seed"])
    host.load_swarm(eng, sw)
    stats = eng.tick()
    assert stats["n_formed"] == gold["n_formed"] and stats["n_merged"] == gold["n_merged"]
    gsha, tbl = _engine_digests(sw, eng)
    assert gsha == gold["groups_sha256"]
    assert int((tbl["task"] != NONE).sum()) == gold["n_with_task"]
    assert _sha(tbl["task"].astype(np.uint32)) == gold["task_sha256"], "per-worker tasks differ (seeded chooser)"
    eng.close()


def test_config1_full_table_against_live_oracle():
    """configs[1]: every worker's (task, applicable count, GROUP_INDEX, GROUP_SIZE, NEXT) against the oracle run
    here, its pair sweep on all host cores (seed 2; seed 1 is pinned by the committed digest)."""
    import os
    sw = baseline_config(1, seed=2)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False, group_id_seed=2)
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    eng = E.Engine(group_id_seed=2)
    host.load_swarm(eng, sw)
    eng.tick()
    task, count = eng.match()
    assert [g[:3] for g in oracle_groups(st)] == [g[:3] for g in engine_groups(eng)]
    cfg_of_node = np.full(sw.W, -1, dtype=np.int32)
    for _, _, cfg, mem, _ in st.groups():
        cfg_of_node[mem] = cfg
    first_o, count_o = orc.pair_sweep_per_worker(tasks, cfgs, cfg_of_node, threads=os.cpu_count() or 1)
    assert np.array_equal(task, first_o) and np.array_equal(count, count_o)
    for w in range(sw.W):
        a = eng.lookup(w)
        if cfg_of_node[w] < 0:
            assert a.task == NONE and a.group_slot == NONE
            continue
        if w % 7 == 0:      # O(T) each: every 7th grouped worker through the oracle's own filter_tasks
            t, gi, gs, nxt = st.filter_tasks(w)
            assert (a.task, a.group_index, a.group_size, a.next_worker) == (NONE if t < 0 else t, gi, gs, nxt), w
    eng.close()


def test_config2_full_size_properties():
    """BASELINE configs[2]: 1M tasks x 100k workers, Zipf-skewed topologies."""
    sw = baseline_config(2, seed=1)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    stats = eng.tick()
    _check_invariants(sw, eng, stats)
    _check_tasks(sw, eng)
    # idempotence: a second tick on the settled swarm forms nothing new and republishes the same table
    before = [eng.lookup(w).task for w in range(0, sw.W, 97)]
    s2 = eng.tick()
    assert s2["n_formed"] == 0 and s2["n_groups"] == stats["n_groups"]
    assert before == [eng.lookup(w).task for w in range(0, sw.W, 97)]
    # the sequential reference kernel forms the same groups
    eng2 = E.Engine(carve_variant=1)
    host.load_swarm(eng2, sw)
    eng2.form_groups()
    assert [g[:3] for g in engine_groups(eng)] == [g[:3] for g in engine_groups(eng2)]
    eng.close()
    eng2.close()


@pytest.mark.parametrize("seed", [1, 2, 7])
@pytest.mark.parametrize("carve_variant", [0, 3])
def test_config1_full_size_groups_bit_exact(seed, carve_variant):
    """BASELINE configs[1] (10k workers, 24 mixed configurations): the groups — ids, configurations, members in
    carve order — equal the oracle's; the oracle's carve needs ~2 s at this size (its pair sweep is not run)."""
    sw = baseline_config(1, seed=seed)
    st = oracle_state_for(sw, reference_shaped=False)
    eng = E.Engine(carve_variant=carve_variant)
    host.load_swarm(eng, sw)
    assert st.try_form_new_groups() == eng.form_groups()
    assert st.try_merge_solo_groups() == eng.merge_solo_groups()
    assert oracle_groups(st) == engine_groups(eng)
    s = eng.last_stats()
    assert s["host_resolved_steps"] == 0
    eng.close()


def test_config1_full_size_properties():
    sw = baseline_config(1, seed=3)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    stats = eng.tick()
    _check_invariants(sw, eng, stats)
    _check_tasks(sw, eng)
    eng.close()


def test_solo_merge_at_baseline_size():
    """try_merge_solo_groups (mod.rs:631-971) with thousands of solo groups: at BASELINE sizes the committed digests
    have n_merged = 0, so the merge pass (CARVE_MODE_MERGE: proximity pass with filter_map semantics over the ordered
    solo list, first-come fallback, dissolve + create bookkeeping) is driven here on purpose — a (1, 1) configuration
    alone makes every 1-GPU node a solo group, then a (2, 8) configuration is enabled and the solos merge.  Groups
    and, event by event, the life-cycle feed (per merge: the dissolved solos in batch order, then the merged group —
    i.e. the creation order of the merged groups) against the oracle."""
    sw = make_swarm(33, 500, 20000)
    sw.configs = [("solo-1gpu", 1, 1, "gpu:count=1"), ("octet-1gpu", 2, 8, "gpu:count=1")]
    sw.topo[:] = -2
    sw.restricted[:] = False
    sw.n_topo[:] = 0
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    only_solo = np.array([1, 0], dtype=np.uint8)
    st = orc.State(nodes, cfgs, enabled=only_solo, tasks=tasks, reference_shaped=False, group_id_seed=4)
    eng = E.Engine(group_id_seed=4)
    host.load_swarm(eng, sw, enabled=0b01)
    eng.enable_group_events()
    n_solo = eng.form_groups()
    assert st.try_form_new_groups() == n_solo > 5000
    assert oracle_groups(st) == engine_groups(eng)
    assert eng.drain_group_events() == st.drain_events()
    st.set_enabled(np.array([1, 1], dtype=np.uint8))
    eng.set_enabled_mask(0b11)
    n_merged = eng.merge_solo_groups()
    assert st.try_merge_solo_groups() == n_merged > 500
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    ev = eng.drain_group_events()
    assert ev == st.drain_events()                                  # merged groups in creation order, solos in batch order
    assert sum(k == E.GROUP_CREATED for k, *_ in ev) == n_merged
    eng.close()


def test_solo_merge_of_the_bench_workload():
    """bench.py's `merge` leg (100k workers, 5,000 solo groups merged by a newly enabled (2, 8) configuration): groups and
    the life-cycle feed in creation order — the digest the bench line carries — against the oracle."""
    from protocol_amd.swarm import events_digest, solo_merge_swarm
    sw = solo_merge_swarm(1)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=np.array([1, 0], dtype=np.uint8), tasks=tasks, reference_shaped=False, group_id_seed=1)
    eng = E.Engine(group_id_seed=1)
    host.load_swarm(eng, sw, enabled=0b01)
    eng.enable_group_events()
    assert eng.form_groups() == st.try_form_new_groups() > 4500      # (the 5,000 less those without usable specs)
    eng.drain_group_events()
    st.drain_events()
    st.set_enabled(np.array([1, 1], dtype=np.uint8))
    eng.set_enabled_mask(0b11)
    n_merged = eng.merge_solo_groups()
    assert st.try_merge_solo_groups() == n_merged > 550
    ev = eng.drain_group_events()
    assert events_digest(ev) == events_digest(st.drain_events())
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    assert eng.debug_merge_streamed() == 1   # (the selections went through the streaming carve: one configuration)
    eng.close()


@pytest.mark.parametrize("how", ["plain", "abort", "uncertain", "one_workgroup"])
def test_solo_merge_through_the_streaming_carve(how, monkeypatch):
    """The merge pass of the bench workload's swarm through carve_stream_kernel in MERGE mode (pm_engine.cpp run_merge): as it
    runs; with the launch made to give up after 40 selections (CARVE_STATE_ABORTED: the single-workgroup kernel takes the
    rest of the list); with every third step sent to the host (glibc distances); with one row-making workgroup."""
    from protocol_amd.swarm import events_digest, solo_merge_swarm
    if how == "one_workgroup":
        monkeypatch.setenv("PM_STREAM_WGS", "1")
    sw = solo_merge_swarm(2)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=np.array([1, 0], dtype=np.uint8), tasks=tasks, reference_shaped=False, group_id_seed=1)
    eng = E.Engine(group_id_seed=1, debug_uncertain_every=3 if how == "uncertain" else 0)
    host.load_swarm(eng, sw, enabled=0b01)
    eng.enable_group_events()
    assert eng.form_groups() == st.try_form_new_groups() > 4500
    eng.drain_group_events()
    st.drain_events()
    st.set_enabled(np.array([1, 1], dtype=np.uint8))
    eng.set_enabled_mask(0b11)
    if how == "abort":
        eng.debug_stream_abort_after(40)
    n_merged = eng.merge_solo_groups()
    assert st.try_merge_solo_groups() == n_merged > 550
    assert events_digest(eng.drain_group_events()) == events_digest(st.drain_events())
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    assert eng.debug_merge_streamed() == 1
    if how == "uncertain":
        assert eng.last_stats()["host_resolved_steps"] > 0
    eng.close()


@pytest.mark.parametrize("seed", [26, 27])
def test_lists_kept_in_hbm(seed):
    """carve_step_mem: the carve of a candidate list too long for the LDS bitmaps (> 262,144 candidates of ONE
    configuration: keys, bitmaps and selection all in HBM / L2, one workgroup-wide argmin round per member, its own
    slot-bit and band geometry).  A swarm that size costs the oracle hours, so a test hook sends every list longer
    than 150 slots down that path: form + merge of a 6 k swarm, bit for bit."""
    sw = make_swarm(seed, 600, 6000)
    st = oracle_state_for(sw, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.debug_mem_lists_above(150)
    n_formed = eng.form_groups()
    assert st.try_form_new_groups() == n_formed > 100
    assert oracle_groups(st) == engine_groups(eng)
    stats = eng.last_stats()
    assert stats["carve_fast_steps"] < 0.5 * stats["carve_steps"]      # (the long lists had no proposals)
    assert st.try_merge_solo_groups() == eng.merge_solo_groups()
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    eng.close()
