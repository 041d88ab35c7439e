#!/usr/bin/env python3
"""Parity fuzz at scale: thousands of seeded swarms through try_form_new_groups (+ try_merge_solo_groups, + every worker's
task) on the engine and on the oracle — groups (ids, configurations, members in order) and tasks must be identical.

The certificate that lets the GPU order candidates by the Haversine TERM instead of glibc's distance rests on a
floating-point error argument (a 2^-36 band, the chord form's 7e-16 / sqrt(a)); the generators below
aim at where such an argument breaks — candidates that tie or nearly tie around a selection's boundary:

  plain        make_swarm as the tests use it (0.0001-degree grid, 32 cities), sizes drawn around the hand-over points of the
               streaming carve (a wave's worth of located candidates, 65..128, groups of one)
  mirror       sites mirrored about a seed's meridian / parallel: exact ties between DIFFERENT sites (host-resolved steps)
  near_mirror  ... nearly mirrored: offsets that straddle a binade, so the terms differ in the last bits
  ulp          coordinates a few ulps apart: distinct sites inside the band
  fine_grid    every node on a 1e-7-degree grid around a few centres: thousands of near-equal distances
  ring         nodes on circles around a centre (equal great-circle distance up to rounding) + the centre's crowd
  wide         configurations with max_group_size 64..200 (more than a row holds: exact steps) and 65..128-candidate lists
  solos        a (1, 1) configuration over a large part of the swarm, then merging enabled (the merge pass's rules)
  unlocated    most nodes without a location (first-come tails, seeds without location)

A third of the swarms run with every n-th step forced through the exact host path (debug_uncertain_every).

    python tools/parity_fuzz.py [--swarms N] [--seed S] [--max-workers W] [--kinds a,b,...]      (on a GPU box)
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from helpers import engine_groups, oracle_groups  # noqa: E402
from oracle import oracle_ffi as orc  # noqa: E402
from protocol_amd import engine as E, host  # noqa: E402
from protocol_amd.swarm import Stream, make_swarm  # noqa: E402

KINDS = ["plain", "mirror", "near_mirror", "ulp", "fine_grid", "ring", "wide", "solos", "unlocated"]


def set_configs(sw, configs):
    sw.configs = configs
    n = len(configs)
    sw.topo = (sw.topo.astype(np.int64) % n).astype(np.int16)
    sw.topo[sw.n_topo[:, None] <= np.arange(sw.topo.shape[1])[None, :]] = -2
    sw.topo[~sw.restricted] = -2
    return sw


def size_configs(rng, wide=False):
    """two to five configurations without requirements (every node a candidate: the geometry decides), sizes random"""
    out = []
    for k in range(int(rng.integers(2, 6))):
        mn = int(rng.choice([1, 2, 2, 3, 4, 5, 8]))
        mx = mn + int(rng.choice([0, 0, 1, 2, 4, 8]))
        if wide and k == 0:
            mn = int(rng.integers(2, 40))
            mx = int(rng.integers(max(mn, 64), 200))
        out.append((f"c{k}", mn, mx, None))
    return out


def generate(kind, seed, max_w):
    rng = np.random.default_rng(seed)
    r = Stream(seed, 77).u64(4)
    W = int(60 + r[0] % max(max_w - 60, 1))
    if kind in ("plain",) and seed % 4 == 0:
        W = int(60 + r[0] % 200)                      # small: a wave's worth of candidates or fewer per configuration
    T = int(100 + r[1] % 1500)
    sw = make_swarm(seed, T, W)
    if kind == "plain":
        return sw
    if kind == "unlocated":
        sw.has_loc[:] = rng.random(W) < rng.choice([0.0, 0.02, 0.1, 0.3])
        return sw
    if kind == "solos":
        sw = set_configs(sw, [("solo", 1, 1, "gpu:count=1"), ("merge-into", 2, int(rng.integers(2, 9)), "gpu:count=1"),
                              ("rest", 2, 4, None)])
        return sw
    sw = set_configs(sw, size_configs(rng, wide=(kind == "wide")))
    sw.has_loc[:] = rng.random(W) < 0.95
    if kind == "wide":
        return sw
    n_c = int(rng.integers(1, 6))
    clat = rng.uniform(-70, 70, n_c)
    clon = rng.uniform(-170, 170, n_c)
    which = rng.integers(0, n_c, W)
    if kind == "mirror":
        d = rng.choice([0.25, 0.5, 0.125, 1.0, 0.0625], n_c)           # exactly representable offsets: exact ties
        side = rng.integers(0, 5, W)                                       # centre, east, west, north, south
        sw.lat[:] = clat[which].round(2) + np.where(side == 3, d[which], np.where(side == 4, -d[which], 0.0))
        sw.lon[:] = clon[which].round(2) + np.where(side == 1, d[which], np.where(side == 2, -d[which], 0.0))
    elif kind == "near_mirror":
        d = rng.choice([0.2, 0.3, 0.1, 0.7, 1.1], n_c)                   # NOT representable: the two differences differ in the last bits
        side = rng.integers(0, 3, W)
        base = np.floor(clon[which])                                        # (an integer meridian: offsets straddle binades around it)
        sw.lat[:] = clat[which].round(1)
        sw.lon[:] = base + np.where(side == 1, d[which], np.where(side == 2, -d[which], 0.0))
    elif kind == "ulp":
        k = rng.integers(0, 9, W)
        sw.lat[:] = clat[which] + k * np.spacing(clat[which])
        sw.lon[:] = clon[which] - (k % 4) * np.spacing(clon[which])
    elif kind == "fine_grid":
        step = rng.choice([1e-7, 1e-6, 1e-5])
        sw.lat[:] = clat[which].round(3) + rng.integers(-40, 41, W) * step
        sw.lon[:] = clon[which].round(3) + rng.integers(-40, 41, W) * step
    elif kind == "ring":
        rad = rng.choice([0.01, 0.5, 3.0], n_c)
        ang = rng.integers(0, 24, W) * (2 * np.pi / 24)
        on_ring = rng.random(W) < 0.7
        sw.lat[:] = clat[which] * 0.5 + np.where(on_ring, rad[which] * np.sin(ang), 0.0)      # (|lat| <= 35 + radius: one hemisphere)
        sw.lon[:] = clon[which] + np.where(on_ring, rad[which] * np.cos(ang) / np.cos(np.radians(clat[which] * 0.5)), 0.0)
    sw.lat[:] = np.clip(sw.lat, -85.0, 85.0)
    sw.lon[:] = np.clip(sw.lon, -179.0, 179.0)
    return sw


def run_case(kind, seed, max_w):
    sw = generate(kind, seed, max_w)
    h = (seed * 2654435761) >> 11                          # (not seed % n: the kinds go round with the seed)
    every = (0, 0, 3, 0, 0, 7)[h % 6]                     # a third of the swarms with forced host steps
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    merge = kind == "solos"
    if merge:
        enabled_first = np.zeros(len(sw.configs), dtype=np.uint8)
        enabled_first[0] = 1
    st = orc.State(nodes, cfgs, enabled=(enabled_first if merge else enabled), tasks=tasks, reference_shaped=False, group_id_seed=seed)
    eng = E.Engine(group_id_seed=seed, debug_uncertain_every=every)
    host.load_swarm(eng, sw)
    t0 = time.perf_counter()
    if merge:                                             # solos first, then the merge pass with everything enabled
        eng.set_enabled_mask(1)
        eng.tick()
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        eng.set_enabled_mask((1 << len(sw.configs)) - 1)
        st.set_enabled(np.ones(len(sw.configs), dtype=np.uint8))
    stats = eng.tick()
    ms = (time.perf_counter() - t0) * 1e3
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    want_tasks = [st.get_task_for_node(w) for w in range(sw.W)]
    ok = sorted(oracle_groups(st)) == sorted(engine_groups(eng))
    if ok:
        for w in range(sw.W):
            t = eng.lookup(w).task
            if (-1 if t == E.PM_NONE else t) != want_tasks[w]:
                ok = False
                break
    c = eng.debug_carve_counters()
    out = dict(kind=kind, seed=seed, W=sw.W, T=sw.T, every=every, groups=int(stats["n_groups"]), merged=int(stats["n_merged"]),
               host_resolved=int(stats["host_resolved_steps"]), aborts=int(c["stream_aborts"]), ms=ms, ok=ok)
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--swarms", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=600000)
    ap.add_argument("--max-workers", type=int, default=2500)
    ap.add_argument("--kinds", default=",".join(KINDS))
    ap.add_argument("--verbose", action="store_true")
    args = ap.parse_args()
    kinds = args.kinds.split(",")
    t_all = time.perf_counter()
    per = {k: dict(n=0, bad=0, groups=0, host=0, forced=0, merged=0, workers=0) for k in kinds}
    failures = []
    for i in range(args.swarms):
        kind = kinds[i % len(kinds)]
        seed = args.seed + i
        try:
            r = run_case(kind, seed, args.max_workers if i % 50 else 4 * args.max_workers)   # (every 50th swarm four times the size)
        except Exception as ex:     # an engine error is a failure of the case, not of the run
            r = dict(kind=kind, seed=seed, W=0, T=0, every=0, groups=0, merged=0, host_resolved=0, aborts=0, ms=0.0, ok=False, error=repr(ex))
        p = per[kind]
        p["n"] += 1
        p["bad"] += 0 if r["ok"] else 1
        p["groups"] += r["groups"]
        p["host"] += r["host_resolved"]
        p["forced"] += 1 if r["every"] else 0
        p["merged"] += r["merged"]
        p["workers"] += r["W"]
        if not r["ok"]:
            failures.append(r)
        if args.verbose or not r["ok"]:
            print(f"{kind:12s} seed {seed} W {r['W']:6d} T {r['T']:5d} host-every {r['every']}: {r['groups']:5d} groups, merged {r['merged']:4d}, "
                  f"host-resolved {r['host_resolved']:4d}, {r['ms']:7.2f} ms  {'ok' if r['ok'] else 'MISMATCH ' + r.get('error', '')}", flush=True)
    print(f"parity fuzz: seeds {args.seed}..{args.seed + args.swarms - 1}, max workers {args.max_workers} (every 50th swarm x4)")
    print(f"{'kind':12s} {'swarms':>7s} {'forced':>7s} {'workers':>9s} {'groups':>9s} {'merged':>7s} {'host steps':>11s} {'mismatches':>11s}")
    for k in kinds:
        p = per[k]
        print(f"{k:12s} {p['n']:7d} {p['forced']:7d} {p['workers']:9d} {p['groups']:9d} {p['merged']:7d} {p['host']:11d} {p['bad']:11d}")
    bad = sum(p["bad"] for p in per.values())
    print(f"{args.swarms} swarms, {sum(p['groups'] for p in per.values())} groups, {bad} mismatches, {time.perf_counter() - t_all:.1f} s")
    for r in failures[:20]:
        print("  FAILED:", r)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
