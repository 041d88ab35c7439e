"""The proposer's spatial index (cell_*_kernel, cell_walk in pm_propose.inc / pm_prep.inc): a seed that walks the grid cells around it
must produce the row the whole-list sweep produces — entries, certificate flags and all — so every carve comes out the
same whichever way its proposals were made.  Modes (pm_debug_prune_mode): 0 never walk, 1 walk when it pays (the
default), 2 walk whenever there is an index (and build one for any swarm), 3 = 2 with every seed sent through the
walk's whole-list fallback."""
import json
import os

import numpy as np
import pytest

from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import baseline_config, make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for
import test_gpu_geography as geo

pytestmark = pytest.mark.gpu


def _carve(sw, mode, **kw):
    eng = E.Engine(**kw)
    eng.debug_prune_mode(mode)
    host.load_swarm(eng, sw)
    n = eng.form_groups()
    out = (n, engine_groups(eng), eng.debug_carve_counters(), eng.last_stats())
    eng.close()
    return out


@pytest.mark.parametrize("seed", [1, 2])
def test_config1_same_groups_whichever_way_the_rows_were_made(seed):
    """BASELINE configs[1]: the whole-list sweep, the default policy, the forced walk and the forced fallback."""
    sw = baseline_config(1, seed=seed)
    n0, g0, c0, _ = _carve(sw, 0)
    assert c0["cell_g"] == 0 and c0["pruned_batches"] == 0
    for mode in (1, 2, 3):
        n, g, c, stats = _carve(sw, mode)
        assert (n, g) == (n0, g0), mode
        if mode == 1:  # (the default policy builds no index below PM_CELL_AUTO_N eligible positions: it does not pay there)
            assert c["cell_g"] == 0 and c["pruned_batches"] == 0, c
            continue
        assert c["cell_g"] == 32 and 0 < c["n_indexed"] <= sw.W, c
        assert c["pruned_batches"] > 0, (mode, c)
        if mode == 2:
            assert c["pruned_batches"] >= c["batches"] // 2, c
        if mode == 3:
            assert c["prune_fallbacks"] > 0, c
        assert stats["host_resolved_steps"] == 0


@pytest.mark.parametrize("mode", [2, 3])
def test_big_lists_against_the_oracle(mode):
    """30k workers: lists above 8192 candidates (18 slot bits, the wider certificate band) through the index."""
    sw = make_swarm(2, 2000, 30000, zipf=True)
    st = oracle_state_for(sw, reference_shaped=False)
    n, g, c, stats = _carve(sw, mode)
    assert st.try_form_new_groups() == n
    assert oracle_groups(st) == g
    assert c["cell_g"] == 32 and c["pruned_batches"] > 0, c
    assert stats["host_resolved_steps"] == 0


def test_config2_forced_walk_against_the_oracle_digest():
    """1M x 100k, every batch through the 64^3 index — also the late ones, whose lists are a small part of it."""
    import hashlib
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scale_digests.json")))["cfg2_seed1"]
    sw = baseline_config(gold["config"], seed=gold["seed"])
    eng = E.Engine(group_id_seed=gold["seed"])
    eng.debug_prune_mode(2)
    host.load_swarm(eng, sw)
    stats = eng.tick()
    c = eng.debug_carve_counters()
    assert stats["n_formed"] == gold["n_formed"]
    _, groups, members = eng.get_groups()
    h = hashlib.sha256()
    for a in (groups["id"].astype(np.uint64), groups["config"].astype(np.uint32), groups["n_members"].astype(np.uint32),
              members.astype(np.uint32)):
        h.update(np.ascontiguousarray(a).tobytes())
    assert h.hexdigest() == gold["groups_sha256"]
    assert c["cell_g"] == 64 and c["pruned_batches"] > 0, c
    assert stats["host_resolved_steps"] == 0
    eng.close()


def test_default_policy_walks_the_long_lists_of_config2():
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scale_digests.json")))["cfg2_seed1"]
    sw = baseline_config(gold["config"], seed=gold["seed"])
    eng = E.Engine(group_id_seed=gold["seed"])
    host.load_swarm(eng, sw)
    eng.tick()
    c = eng.debug_carve_counters()
    # (the streaming carve counts configurations — some walk the index, the rest sweep their candidate bitmap — the
    # batch pipeline batches)
    if c["stream"]:
        assert c["cell_g"] == 64 and c["pruned_batches"] > 0 and c["stream_listed"] > 0, c
    else:
        assert c["cell_g"] == 64 and 0 < c["pruned_batches"] < c["batches"], c
    eng.close()


GEO_CASES = [n for n in dir(geo) if n.startswith("test_")]


@pytest.mark.parametrize("mode", [2, 3])
@pytest.mark.parametrize("case", GEO_CASES)
def test_geography_suite_through_the_index(case, mode, monkeypatch):
    """Every degenerate geography of test_gpu_geography.py (one site, no locations, exact ties that go to the host,
    near-mirror sites, a lone node behind a crowd, antipodes) once more, proposals made from the index."""
    orig = E.Engine.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        self.debug_prune_mode(mode)

    monkeypatch.setattr(E.Engine, "__init__", init)
    getattr(geo, case)()


def test_small_swarms_many_seeds():
    """700 workers, 12 seeds: forced index (32^3 over a few hundred positions), walk and fallback, against the oracle."""
    for seed in range(40, 52):
        sw = make_swarm(seed, 60, 700)
        st = oracle_state_for(sw, reference_shaped=True)
        n_o = st.try_form_new_groups()
        g_o = oracle_groups(st)
        for mode in (2, 3):
            n, g, c, _ = _carve(sw, mode)
            assert (n, g) == (n_o, g_o), (seed, mode)
            assert c["cell_g"] == 32, c


@pytest.mark.parametrize("half", ["north-east", "south-west"])
def test_edges_of_the_grid(half):
    """Workers at the poles, on the equator at the axes (unit coordinates of exactly +-1 and 0: the first and the last
    cell of the grid, and the clamp of cell_coord) and spread over half a globe (no antipodal pairs: those are outside
    the reference's domain), through the forced walk, the forced fallback and the whole-list sweep, against the oracle."""
    sw = geo._swarm(seed=77, W=1500)
    rng = np.random.default_rng(5)
    W = sw.W
    sw.has_loc[:] = rng.random(W) < 0.95
    if half == "north-east":
        lat = np.round(rng.random(W) * 89.0, 4)
        lon = np.round(-80.0 + rng.random(W) * 160.0, 4)
        spots = [(90.0, 0.0), (0.0, 0.0), (0.0, 80.0), (45.0, -80.0)]
    else:
        lat = np.round(-rng.random(W) * 89.0, 4)
        lon = np.round(100.0 + rng.random(W) * 160.0, 4)
        lon = np.where(lon > 180.0, lon - 360.0, lon)
        spots = [(-90.0, 0.0), (0.0, 180.0), (0.0, -90.0), (0.0, -180.0)]
    which = rng.integers(0, 3 * len(spots), W)          # a third of the swarm sits exactly on the spots
    for k, (la, lo) in enumerate(spots):
        lat = np.where(which == k, la, lat)
        lon = np.where(which == k, lo, lon)
    sw.lat[:] = lat
    sw.lon[:] = lon
    st = oracle_state_for(sw, reference_shaped=True)
    n_o = st.try_form_new_groups()
    g_o = oracle_groups(st)
    for mode in (0, 2, 3):
        n, g, c, _ = _carve(sw, mode)
        assert (n, g) == (n_o, g_o), mode
        assert mode == 0 or (c["cell_g"] == 32 and c["pruned_batches"] > 0), c
