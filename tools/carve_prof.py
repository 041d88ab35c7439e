#!/usr/bin/env python3
"""Phase breakdown of the carve kernel: builds a PM_CARVE_PROF variant of the library (s_memtime ticks
accumulated per phase by thread 0) and runs cold full-swarm matches on BASELINE configs[1]."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B

prof_lib = os.path.join(ROOT, "protocol_amd", "libpm_engine_prof.so")
B.build(force=True, defines=["PM_CARVE_PROF"] + (["PM_CARVE_PROF_FINE"] if os.environ.get("PM_PROF_FINE") else []), out=prof_lib)
B.LIB_PATH = prof_lib
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

T, W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
sw = make_swarm(1, T, W)
eng = E.Engine()
host.load_swarm(eng, sw)
for it in range(3):
    eng.reset_groups()
    s = eng.tick()
out = (C.c_ulonglong * 32)()
E.lib().pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
E.lib().pm_debug_carve_prof(eng._h, out, 32)
out[9] = out[26] + out[27] + out[28] + out[29]
tot = out[15] + out[9] + out[10] + out[13] + out[12] + out[14]
print(f"carve kernels {s['ms_carve_kernel']:.3f} ms, {s['carve_steps']} steps ({s['carve_fast_steps']} fast), "
      f"{1e3 * s['ms_carve_kernel'] / s['carve_steps']:.2f} us/step")
pct = lambda v: f"{100.0 * v / max(tot, 1):5.1f}%"
print(f"  launch anatomy (ticks): init/status={out[15]} ({pct(out[15])}) prepare(compaction)={out[9]} ({pct(out[9])}) "
      f"load-list={out[10]} ({pct(out[10])}) run(all steps)={out[13]} ({pct(out[13])}) flush={out[12]} ({pct(out[12])}) "
      f"group_of+exit={out[14]} ({pct(out[14])})")
print(f"  inside run: rounds={out[0]} ({pct(out[0])}; {out[1]} rounds, {out[2]} commits, {out[3]} retries, {out[4]} exact-sweep stops) "
      f"sequential path={out[11]} ({pct(out[11])}) staging={out[20]} ({pct(out[20])}; {out[21]} refills) exact steps={out[22]} ({pct(out[22])})")
print(f"  prepare: count passes={out[26]} ({out[30]} calls) placement={out[27]} proposal limit={out[28]} other={out[29]}")
print(f"  exact-sweep reasons: no proposal={out[16]} debug hook={out[17]} row exhausted={out[18]} certificate={out[19]}")
print(f"  proposer waves: {out[24]} proposals ({out[25]} with near misses in the tracker); ticks summed over waves: same-site links={out[5]} (slowest wave {out[31]}) sweep={out[6]} read-out={out[7]} flags={out[8]}; slowest wave={out[23]}")
if os.environ.get("PM_PROF_FINE"):
    print(f"  fine: wave0 spec={out[5]}; wave1 wait={out[6]} commit={out[7]} sync={out[8]} chk={out[16]} b2wait={out[17]}; wave7 chk={out[18]} spec={out[19]}")
print(f"  total ticks {tot}; ticks per ms = {tot / s['ms_carve_kernel']:.0f}")
