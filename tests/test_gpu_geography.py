"""Degenerate geographies through the carve: they drive the certificate / same-site / host-resolve branches of
the proposal rows and the speculative rounds.  Every case: groups (ids, configurations, members in carve order)
bit-exact against the oracle, for the default launch order, the sequential kernel and the single-wave validator."""
import numpy as np
import pytest

from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for

pytestmark = pytest.mark.gpu


def _check(sw, expect_host_resolved=None, **engine_kw):
    for variant in (0, 1, 3):
        st = oracle_state_for(sw, reference_shaped=True)
        eng = E.Engine(carve_variant=variant, **engine_kw)
        host.load_swarm(eng, sw)
        assert st.try_form_new_groups() == eng.form_groups(), variant
        assert oracle_groups(st) == engine_groups(eng), variant
        resolved = eng.last_stats()["host_resolved_steps"]
        if expect_host_resolved is not None:
            assert (resolved > 0) == expect_host_resolved, (variant, resolved)
        eng.close()


def _set_configs(sw, configs):
    """replace the configurations and fold the tasks' topology indices onto them"""
    sw.configs = configs
    n = len(configs)
    sw.topo = (sw.topo.astype(np.int64) % n).astype(np.int16)
    sw.topo[sw.n_topo[:, None] <= np.arange(sw.topo.shape[1])[None, :]] = -2
    sw.topo[~sw.restricted] = -2
    return sw


def _swarm(seed=31, W=700):
    return _set_configs(make_swarm(seed, 50, W), [("quad", 4, 4, None), ("pairs", 2, 2, None)])


def test_every_worker_at_one_site():
    sw = _swarm()
    sw.has_loc[:] = True
    sw.lat[:] = 48.8566
    sw.lon[:] = 2.3522
    _check(sw, expect_host_resolved=False)      # exact ties everywhere: input order decides, no near-ties


def test_no_worker_has_a_location():
    sw = _swarm()
    sw.has_loc[:] = False
    _check(sw, expect_host_resolved=False)


def test_two_sites_and_some_unlocated():
    sw = _swarm()
    rng = np.random.default_rng(3)
    sw.has_loc[:] = rng.random(sw.W) < 0.8
    west = rng.random(sw.W) < 0.5
    sw.lat[:] = np.where(west, 37.7749, 52.52)
    sw.lon[:] = np.where(west, -122.4194, 13.405)
    _check(sw, expect_host_resolved=False)


def test_wide_groups_drain_a_city():
    """groups of 16 out of clusters of ~40 co-located nodes: the same-site chain runs dry and is topped up from
    the proposal row"""
    sw = _set_configs(make_swarm(32, 50, 1200), [("wide", 9, 16, None), ("rest", 2, 3, None)])
    rng = np.random.default_rng(5)
    city = rng.integers(0, 30, sw.W)
    sw.has_loc[:] = True
    sw.lat[:] = -60.0 + 4.0 * city
    sw.lon[:] = -170.0 + 11.0 * city
    jitter = rng.random(sw.W) < 0.3          # a third of the nodes sit near, not at, their city
    sw.lat[:] += np.where(jitter, rng.normal(0, 0.05, sw.W), 0.0)
    sw.lon[:] += np.where(jitter, rng.normal(0, 0.05, sw.W), 0.0)
    _check(sw, expect_host_resolved=False)


def test_last_ulp_neighbours():
    """coordinates that differ in the last bits of the mantissa: distinct sites a few ulps apart — wherever two
    of them fall inside the certificate band the engine must not guess (the host settles such a step with
    glibc); either way the result is the oracle's"""
    sw = _swarm(33, 400)
    sw.has_loc[:] = True
    base_lat = np.float64(40.7128)
    base_lon = np.float64(-74.0060)
    k = np.arange(sw.W) % 7
    sw.lat[:] = base_lat + k * np.spacing(base_lat)
    sw.lon[:] = base_lon - k * np.spacing(base_lon)
    _check(sw)


def test_mirror_sites_tie_exactly_and_go_to_the_host():
    """Two sites mirrored about the seed's meridian are exactly equidistant — in the reference too, where the
    stable sort then falls back to input order.  They are different sites, so the certificate cannot vouch for
    the slot order: those steps are settled on the host, and the result is the oracle's."""
    sw = _swarm(36, 360)
    sw.has_loc[:] = True
    which = np.arange(sw.W) % 12           # one node in twelve at the centre, the rest alternate east / west
    sw.lat[:] = 12.5
    sw.lon[:] = np.where(which == 0, 30.0, np.where(which % 2 == 1, 30.25, 29.75))
    _check(sw, expect_host_resolved=True)


def test_near_mirror_sites_with_a_lone_survivor():
    """Sites NEARLY mirrored about the seed's meridian: lon 31.8 / 32.2 around 32.0 straddle a binade, so the two
    longitude differences are not bit-identical; the Haversine terms differ by ~4e-14 relative — far below the
    2^-39 the packed keys drop, so the keys tie and only the slot order separates them — while the reference's
    distances differ (glibc: east is farther by 1.8e-14 relative).  One site holds a single low-slot node, the
    other a crowd: the engine's slot order would take {lone, first of the crowd}; where the lone node is on the
    truly farther side the reference takes {first, second of the crowd}.  A SELECTED entry of another site inside
    the band must therefore fail the certificate (the band is symmetric around the last selected entry; the
    round-1 test looked at unselected entries only and certified such a step).  Both orientations are present."""
    from oracle import oracle_ffi as orc
    n_cluster, per = 8, 12
    sw = _set_configs(make_swarm(37, 50, n_cluster * per), [("trio", 3, 3, None), ("pairs", 2, 2, None)])
    sw.status[:] = 2
    sw.has_p2p[:] = True
    sw.has_loc[:] = True
    k = np.arange(sw.W) % per               # slot 0 of a cluster: centre, slot 1: the lone node, rest: the crowd
    c = np.arange(sw.W) // per
    sw.lat[:] = 10.0 + 3.0 * c
    lone_east = (c % 2) == 0
    east, west = 32.2, 31.8
    sw.lon[:] = np.where(k == 0, 32.0, np.where(k == 1, np.where(lone_east, east, west),
                                                np.where(lone_east, west, east)))
    d_e = orc.calculate_distance(10.0, 32.0, 10.0, east)
    d_w = orc.calculate_distance(10.0, 32.0, 10.0, west)
    assert d_e > d_w and d_e - d_w < 1e-12 * d_e          # a near tie, not an exact one; east is farther
    _check(sw, expect_host_resolved=True)


def test_lone_node_of_another_site_behind_a_crowd_beyond_the_row():
    """The tail certificate of a proposal row: a crowd of co-located nodes fills the row and continues far beyond
    it (all at the row's last entry's site — harmless), and ONE node of another site, nearly mirrored about the
    seed's meridian and therefore inside the band, comes last in input order: its key ties with the crowd's, the
    slot order puts it behind all of them, so it is not in the row — yet the reference, whose distances differ
    there, ranks it FIRST.  The proposer has to notice it among the candidates it rejected at the threshold (the
    tracker keeps the nearest rejected key at a site other than the first one's) and withhold `tail_ok`; the step
    then goes to the exact sweep and on to the host.  Several clusters, both orientations, lists longer than one
    64-slot stride so that the lone node is rejected, not pushed out."""
    from oracle import oracle_ffi as orc
    n_cluster, crowd = 6, 150
    per = crowd + 2
    sw = _set_configs(make_swarm(38, 50, n_cluster * per), [("eight", 8, 8, None), ("pairs", 2, 2, None)])
    sw.status[:] = 2
    sw.has_p2p[:] = True
    sw.has_loc[:] = True
    k = np.arange(sw.W) % per               # slot 0 of a cluster: the seed; 1 .. crowd: the crowd; last: the lone node
    c = np.arange(sw.W) // per
    sw.lat[:] = 10.0 + 3.0 * c
    lone_west = (c % 2) == 0                # west is the truly nearer side (see the previous test)
    east, west = 32.2, 31.8
    sw.lon[:] = np.where(k == 0, 32.0, np.where(k == per - 1, np.where(lone_west, west, east),
                                                np.where(lone_west, east, west)))
    d_e = orc.calculate_distance(10.0, 32.0, 10.0, east)
    d_w = orc.calculate_distance(10.0, 32.0, 10.0, west)
    assert d_e > d_w and d_e - d_w < 1e-12 * d_e
    _check(sw, expect_host_resolved=True)
    # what is being protected: in the clusters whose lone node is on the nearer side the reference's first group
    # holds it, although seven of the crowd come first in input order
    st = oracle_state_for(sw, reference_shaped=True)
    st.try_form_new_groups()
    first_groups = {min(m): m for (_s, _id, _c, m, _t) in st.groups()}
    for cl in range(n_cluster):
        seed, lone = cl * per, cl * per + per - 1
        assert (lone in first_groups[seed]) == bool(lone_west[seed])


def test_antipodal_points_are_outside_the_reference_domain():
    """Two clusters at exact antipodes.  For such pairs the reference's Haversine term rounds to a > 1 about half
    of the time, its distance is NaN, and `partial_cmp(..).unwrap_or(Equal)` (mod.rs:239-253) stops being an
    order: what `sort_by` returns then depends on the sort implementation (recent Rust may even panic), and the
    oracle's merge sort is just one such outcome.  The engine orders by the Haversine term itself, which stays
    well defined; all it can promise here is a valid, deterministic carve — the same for every kernel variant."""
    sw = _swarm(34, 300)
    sw.has_loc[:] = True
    east = (np.arange(sw.W) % 2) == 0
    rng = np.random.default_rng(9)
    sw.lat[:] = np.where(east, 10.0, -10.0) + rng.normal(0, 1e-7, sw.W)
    sw.lon[:] = np.where(east, 20.0, -160.0) + rng.normal(0, 1e-7, sw.W)
    st = oracle_state_for(sw, reference_shaped=True)
    n_oracle = st.try_form_new_groups()
    results = []
    for variant in (0, 1, 3):
        eng = E.Engine(carve_variant=variant)
        host.load_swarm(eng, sw)
        assert eng.form_groups() == n_oracle          # the greedy count does not depend on the order
        gow, groups, members = eng.get_groups()
        seen = np.zeros(sw.W, dtype=np.int32)
        np.add.at(seen, members, 1)
        assert seen.max() <= 1
        for g in groups:
            mem = members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])]
            # a group never straddles the antipodes: the three nearest of a seed are in its own cluster
            assert len(set(east[mem].tolist())) == 1
        results.append(engine_groups(eng))
        eng.close()
    assert results[0] == results[1] == results[2]


def test_proximity_disabled_is_first_come():
    sw = _swarm(35, 500)
    for variant in (0, 1, 3):
        st = oracle_state_for(sw, reference_shaped=True, proximity=False)
        eng = E.Engine(carve_variant=variant, proximity=False)
        host.load_swarm(eng, sw)
        assert st.try_form_new_groups() == eng.form_groups()
        assert oracle_groups(st) == engine_groups(eng)
        eng.close()
