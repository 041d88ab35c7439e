#!/usr/bin/env python3
"""A neighbour row made by ONE wave alone on a CU (a -DPM_ROW_BENCH build: carve_row_bench_kernel), after a match of configs[1]:
    PM_EXP_LIB=protocol_amd/variants/libpm_engine_rowbench.so python tools/row_bench.py [config index ...]
cycles per row and per candidate, the split seed columns / sweep / finish, a checksum of the row (variant builds that take pieces out
of the row maker show in it)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B
B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

sw = baseline_config(1, seed=1)
eng = E.Engine()
host.load_swarm(eng, sw)
eng.tick()
L = E.lib()
L.pm_debug_row_bench.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong)]
L.pm_debug_row_bench.restype = C.c_int32
out = (C.c_ulonglong * 8)()
cis = [int(a) for a in sys.argv[1:]] or [0, 2, 7, 11]
print(f"library {os.path.basename(B.LIB_PATH)}")
modes = {0: "the row as it is made", 1: "keys, no row", 3: "no keys, no row: gathers and passes", 7: "no gathers either", 8: "the passes over the bitmap alone"}
for ci in cis:
  for mode, mname in modes.items():
    for kth in (3,) if mode else (3, 40):
        rc = L.pm_debug_row_bench(eng._h, ci, kth, 50, mode, out)
        if rc:
            print(f"configuration {ci}: rc {rc}")
            break
        t, reps, swept = out[0], max(out[1], 1), max(out[2], 1)
        print(f"[{mname:38s}] configuration {ci:2d} seed {out[7]:5d}: {swept:5d} candidates, {t / reps:9.0f} cycles a row = {t / reps / swept:5.1f} a candidate "
              f"({t / reps / 2400:.1f} us); seed columns {out[4] / reps:6.0f}, sweep {out[5] / reps:8.0f}, finish {out[6] / reps:6.0f}; check {out[3]:016x}")
eng.close()
