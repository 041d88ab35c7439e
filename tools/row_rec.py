#!/usr/bin/env python3
"""What the rows of a streaming carve cost their waves, from a measuring build (-DPM_ROW_REC: eight time stamps per ticket,
plain stores, nothing shared between the waves — the PM_CARVE_PROF build's counters are atomics on a handful of words and
slow the rows they count):  PM_EXP_LIB=protocol_amd/variants/libpm_engine_rowrec.so python tools/row_rec.py [T W]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B
B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

T, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
sw = make_swarm(1, T, W, zipf=(W >= 100000))
eng = E.Engine()
host.load_swarm(eng, sw)
for it in range(3):
    eng.reset_groups()
    s = eng.tick()
cap = 1 << 15
buf = np.zeros((cap, 8), dtype=np.uint64)
n = C.c_uint32(0)
L = E.lib()
L.pm_debug_row_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
L.pm_debug_row_records.restype = C.c_int32
assert L.pm_debug_row_records(eng._h, buf.ctypes.data, cap, C.byref(n)) == 0
r = buf[: n.value]
r = r[r[:, 0] != 0]
if not len(r):
    raise SystemExit("no records (not a PM_ROW_REC build)")
MODE_BITMAP, MODE_WALK = 1, 2  # (SROW_BITMAP, SROW_WALK: pm_device.h)
tick = 100.0  # the PM_ROW_REC build stamps with the real-time counter (100 MHz, one clock for all CUs)
print(f"T={T} W={W}: carve kernel {s['ms_carve_kernel']:.3f} ms, {len(r)} rows recorded")
d = lambda a, b: (r[:, b].astype(np.int64) - r[:, a].astype(np.int64)) / tick
mode = (r[:, 6] >> np.uint64(32)).astype(np.int64)
swept = (r[:, 6] & np.uint64(0xFFFFFFFF)).astype(np.int64)
total = d(0, 5)
for m, name in ((MODE_BITMAP, "bitmap sweep"), (MODE_WALK, "index walk")):
    k = mode == m
    if not k.any():
        continue
    t = total[k]
    print(f"  {name}: {k.sum()} rows, {swept[k].mean():.0f} candidates; us: mean {t.mean():.1f}, p10 {np.percentile(t, 10):.1f}, p50 {np.percentile(t, 50):.1f}, "
          f"p90 {np.percentile(t, 90):.1f}, p99 {np.percentile(t, 99):.1f}, max {t.max():.1f}")
    if m == MODE_BITMAP:
        seg = [("seen -> first pass packed", 0, 1), ("-> first keys", 1, 2), ("-> candidates through", 2, 3), ("-> finished", 3, 4), ("-> stored", 4, 5)]
        order = np.argsort(t)
        for lo, hi, lab in ((0, 0.25, "fastest quarter"), (0.25, 0.75, "middle half"), (0.75, 1.0, "slowest quarter")):
            idx = np.where(k)[0][order[int(lo * len(t)): int(hi * len(t))]]
            print(f"    {lab}: " + ", ".join(f"{nm} {((r[idx, b].astype(np.int64) - r[idx, a].astype(np.int64)) / tick).mean():.1f}" for nm, a, b in seg))
eng.close()
