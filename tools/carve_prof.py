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
B.build(force=True, defines=["PM_CARVE_PROF"] + (["PM_CARVE_PROF_FINE"] if os.environ.get("PM_PROF_FINE") else []) +
        os.environ.get("PM_EXTRA_DEFINES", "").split(), out=prof_lib)
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
lanes = out[9]
out[9] = out[26] + out[27] + out[28] + out[29]
tot = out[15] + out[9] + out[10] + out[13] + out[12] + out[14]
print(f"carve kernels {s['ms_carve_kernel']:.3f} ms, {s['carve_steps']} steps ({s['carve_fast_steps']} fast), "
      f"{1e3 * s['ms_carve_kernel'] / s['carve_steps']:.2f} us/step")
pct = lambda v: f"{100.0 * v / max(tot, 1):5.1f}%"
print(f"  launch anatomy (ticks): init/status={out[15]} ({pct(out[15])}) prepare(compaction)={out[9]} ({pct(out[9])}) "
      f"load-list={out[10]} ({pct(out[10])}) run(all steps)={out[13]} ({pct(out[13])}) flush={out[12]} ({pct(out[12])}) "
      f"group_of+exit={out[14]} ({pct(out[14])})")
print(f"  inside run: chain of located steps={out[0]} ({pct(out[0])}; {out[1]} calls, {out[2]} commits = {out[0] / max(out[2], 1):.0f} ticks each, "
      f"{out[4]} hand-overs) wave 0 in total={out[11]} ({pct(out[11])}) exact steps={out[22]} ({pct(out[22])})")
print(f"  chain anatomy (ticks): waiting for rows={out[16]} steps={out[18]} stopping the other waves={out[17]}; {out[19]} batches of steps, {out[24]} of them waited, {out[23]} seeds dead at their turn")
print(f"  prepare: count passes={out[26]} ({out[30]} calls) placement={out[27]} proposal limit={out[28]} other={out[29]}")
print(f"  exact-sweep reasons: no proposal={out[20]} debug hook={out[21]} row exhausted={out[25]} certificate={out[31]}")
print(f"  total ticks {tot}; ticks per ms = {tot / s['ms_carve_kernel']:.0f}")
