#!/usr/bin/env python3
"""A sweep of seeded swarms of random sizes through try_form_new_groups on the engine and on the oracle — groups (ids,
configurations, members in order) must be identical.  Sizes are drawn so that configurations end (and start) everywhere
around the hand-over points of the streaming carve's validator: a wave's worth of located candidates and fewer (rows made
by helper waves), 65..128 (two a lane), groups of one node, and the chain above them.

    python tools/parity_fuzz.py [n_cases] [first_seed]     (on a GPU box; prints one line per case and a summary)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

from helpers import engine_groups, oracle_groups, oracle_state_for  # noqa: E402
from protocol_amd import engine as E, host  # noqa: E402
from protocol_amd.swarm import Stream, make_swarm  # noqa: E402


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    bad = 0
    t_all = time.perf_counter()
    for k in range(n_cases):
        seed = seed0 + k
        r = Stream(seed, 77).u64(4)
        W = int(60 + r[0] % (8000 if k % 3 else 1500))
        T = int(200 + r[1] % 3000)
        every = (0, 0, 0, 5)[int(r[2] % 4)]   # a quarter of the cases with every fifth step through the host
        sw = make_swarm(seed, T, W)
        st = oracle_state_for(sw, group_id_seed=seed)
        n_o = st.try_form_new_groups()
        eng = E.Engine(group_id_seed=seed, debug_uncertain_every=every)
        host.load_swarm(eng, sw)
        t0 = time.perf_counter()
        n_e = eng.form_groups()
        ms = (time.perf_counter() - t0) * 1e3
        ok = n_o == n_e and oracle_groups(st) == engine_groups(eng)
        c = eng.debug_carve_counters()
        print(f"seed {seed} W {W:6d} T {T:5d} host-every {every}: {n_e:5d} groups, {ms:7.2f} ms, aborts {c['stream_aborts']}, "
              f"host-resolved {eng.last_stats()['host_resolved_steps']:4d}  {'ok' if ok else 'MISMATCH'}", flush=True)
        bad += 0 if ok else 1
        eng.close()
    print(f"{n_cases} swarms, {bad} mismatches, {time.perf_counter() - t_all:.1f} s")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
