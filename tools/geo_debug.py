#!/usr/bin/env python3
"""Debug helper for tests/test_gpu_geography.py: first group that differs from the oracle, with context."""
import os
import sys
import ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_gpu_geography as tg
from helpers import oracle_state_for, oracle_groups, engine_groups
from protocol_amd import engine as E, host
from oracle import oracle_ffi as orc

case = sys.argv[1] if len(sys.argv) > 1 else "ulp"
if case == "ulp":
    sw = tg._swarm(33, 400)
    sw.has_loc[:] = True
    base_lat, base_lon = np.float64(40.7128), np.float64(-74.0060)
    k = np.arange(sw.W) % 7
    sw.lat[:] = base_lat + k * np.spacing(base_lat)
    sw.lon[:] = base_lon - k * np.spacing(base_lon)
else:
    sw = tg._swarm(34, 300)
    sw.has_loc[:] = True
    east = (np.arange(sw.W) % 2) == 0
    rng = np.random.default_rng(9)
    sw.lat[:] = np.where(east, 10.0, -10.0) + rng.normal(0, 1e-7, sw.W)
    sw.lon[:] = np.where(east, 20.0, -160.0) + rng.normal(0, 1e-7, sw.W)
    k = east.astype(int)
L = orc.lib()
L.orc_calculate_distance.restype = C.c_double
L.orc_calculate_distance.argtypes = [C.c_double] * 4
for variant in (1, 0):
    st = oracle_state_for(sw, reference_shaped=True)
    eng = E.Engine(carve_variant=variant)
    host.load_swarm(eng, sw)
    st.try_form_new_groups()
    eng.form_groups()
    a, b = oracle_groups(st), engine_groups(eng)
    s = eng.last_stats()
    print(f"variant {variant}: oracle {len(a)} groups, engine {len(b)}; host-resolved {s['host_resolved_steps']} steps {s['carve_steps']} fast {s['carve_fast_steps']}")
    taken = set()
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print(" first difference at group", i, "oracle", x[1], x[2], "engine", y[1], y[2])
            only_o = [m for m in x[2] if m not in y[2]]
            only_e = [m for m in y[2] if m not in x[2]]
            common = [m for m in x[2] if m in y[2]]
            print("  only oracle", only_o, "k", [int(k[m]) for m in only_o], " only engine", only_e, "k", [int(k[m]) for m in only_e],
                  " common k", [int(k[m]) for m in common])
            for seed in common:
                d = lambda m: L.orc_calculate_distance(sw.lat[seed], sw.lon[seed], sw.lat[m], sw.lon[m])
                print(f"   from {seed}: d(only-oracle)={[d(m).hex() for m in only_o]} d(only-engine)={[d(m).hex() for m in only_e]}")
            print("  free before this group (lowest indices of each side's extra member):",
                  [m in taken for m in only_o + only_e])
            break
        taken.update(x[2])
    eng.close()
