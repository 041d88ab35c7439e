#!/usr/bin/env python3
"""Time cold full-swarm matches of a variant build: PM_EXP_DEFINES=A=1,B python tools/variant_bench.py [config] [reps]
(prints ms per match, carve ms, batches; the groups are NOT checked here — run the parity tests on the variant kept)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import build as B
if os.environ.get("PM_EXP_LIB"):  # a variant built beforehand (tools/build_variants.py)
    B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
    B.needs_build = lambda: False
elif os.environ.get("PM_EXP_DEFINES"):
    alt = os.path.join(os.path.dirname(B.LIB_PATH), "libpm_engine_exp.so")
    B.build(force=True, defines=os.environ["PM_EXP_DEFINES"].split(","), out=alt)
    B.LIB_PATH = alt
    B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sw = baseline_config(ci, seed=1)
eng = E.Engine()
host.load_swarm(eng, sw)
eng.tick()
ms, carve = [], []
for _ in range(reps):
    eng.reset_groups()
    s = eng.tick()
    ms.append(s["ms_total"] if "ms_total" in s else s["ms_carve"])
    carve.append(s["ms_carve"])
ms.sort(); carve.sort()
import time
eng.match_per_task()
t0 = time.perf_counter()
for _ in range(3):
    eng.match_per_task()
print(f"  pm_match_per_task: {1e3 * (time.perf_counter() - t0) / 3:.3f} ms (incl. the D2H copy of both columns)")
import ctypes as C
out = (C.c_ulonglong * 42)()
E.lib().pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
E.lib().pm_debug_carve_prof(eng._h, out, 42)
print("  validation launches ended by: chain thin %d, seeds used up %d, exact-step thin %d, config exhausted %d | void: other config %d, too stale %d, not entered %d | batches %d void %d" % tuple(out[32:41]))
print(f"config {ci} defines {os.environ.get('PM_EXP_LIB') or os.environ.get('PM_EXP_DEFINES', '-')}: carve p50 {carve[len(carve) // 2]:.3f} ms, groups {s['n_groups']}, "
      f"steps {s['carve_steps']} ({s['carve_fast_steps']} fast), launches {s.get('carve_launches', '?')}")
