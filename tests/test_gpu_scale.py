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
