"""A sweep of medium-sized swarms through the whole carve (form + merge) against the oracle, bit for bit: sizes on
both sides of the small-list / big-list boundary of the validator (8192 candidates), several seeds, and geographies
that stress different parts of the proposal path — the generator's default (90 % located, 40 % snapped to 32
cities), almost everybody in a handful of crowded cities (same-site chains, rows that leave the seed's site out,
near-miss tracker), half of the swarm without a location (location-less keys at the end of every row), and a fine
grid where many distinct sites are a few metres apart (the sine form of the key below ~10 km).

The reference's try_form_new_groups / try_merge_solo_groups (node_groups/mod.rs:478-628, 631-971) as restated by
oracle/pm_oracle.c; its best-effort variant (distances cached, predicate once per (configuration, node)) produces
the same groups as the reference-shaped one (tests/test_oracle_groups.py) and keeps this file at a few seconds."""
import numpy as np
import pytest

from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for

pytestmark = pytest.mark.gpu


def _crowded(sw, rng):
    """nine located workers in ten live in one of six cities (identical coordinates), the rest are scattered"""
    city = rng.integers(0, 6, sw.W)
    lat = np.array([48.8566, 40.7128, 35.6762, 51.5074, 37.7749, 52.52])
    lon = np.array([2.3522, -74.006, 139.6503, -0.1278, -122.4194, 13.405])
    snap = rng.random(sw.W) < 0.9
    sw.lat[:] = np.where(snap, lat[city], sw.lat)
    sw.lon[:] = np.where(snap, lon[city], sw.lon)


def _half_unlocated(sw, rng):
    sw.has_loc[:] = rng.random(sw.W) < 0.5


def _fine_grid(sw, rng):
    """everybody within ~15 km of one point, on a 0.001 degree grid: hundreds of distinct sites, all near"""
    sw.has_loc[:] = True
    sw.lat[:] = 47.0 + np.round(rng.uniform(0, 0.12, sw.W), 3)
    sw.lon[:] = 8.0 + np.round(rng.uniform(0, 0.18, sw.W), 3)


GEOGRAPHIES = {"default": None, "crowded": _crowded, "half_unlocated": _half_unlocated, "fine_grid": _fine_grid}


@pytest.mark.parametrize("W,seed", [(2500, 21), (6000, 22), (9000, 23), (14000, 24), (30000, 25)])
@pytest.mark.parametrize("geo", list(GEOGRAPHIES))
def test_carve_sweep_bit_exact(W, seed, geo):
    if geo != "default" and W in (6000, 14000):
        pytest.skip("the special geographies run at three sizes")
    sw = make_swarm(seed, 2000, W)
    if GEOGRAPHIES[geo] is not None:
        GEOGRAPHIES[geo](sw, np.random.default_rng(seed))
    st = oracle_state_for(sw, reference_shaped=False)
    eng = E.Engine()
    host.load_swarm(eng, sw)
    n_formed = eng.form_groups()
    assert st.try_form_new_groups() == n_formed > W // 40
    assert oracle_groups(st) == engine_groups(eng)
    steps = eng.last_stats()
    if geo != "fine_grid":          # (a regular grid is full of mirror-image sites: near ties at different sites, which
        assert steps["carve_fast_steps"] > 0.9 * steps["carve_steps"]      # the certificate rightly refuses — exact sweeps)
    assert steps["carve_fast_steps"] > 0.5 * steps["carve_steps"]          # the proposals carried the carve
    assert st.try_merge_solo_groups() == eng.merge_solo_groups()
    assert sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    eng.close()
