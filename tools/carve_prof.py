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
B.build(force=True, defines=["PM_CARVE_PROF"], out=prof_lib)
B.LIB_PATH = prof_lib
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

T, W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000, int(sys.argv[2]) if len(sys.argv) > 2 else 10000
sw = make_swarm(1, T, W)
eng = E.Engine()
host.load_swarm(eng, sw)
names = ["seed", "keys", "level1", "barrier1", "level2", "certificate", "barrier2", "commit", "barrier3",
         "compaction", "flush", "fast-path"]
for it in range(3):
    eng.reset_groups()
    s = eng.tick()
out = (C.c_ulonglong * 24)()
E.lib().pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
E.lib().pm_debug_carve_prof(eng._h, out)
tot = out[15] + out[9] + out[10] + out[13] + out[12] + out[14]
print(f"carve kernels {s['ms_carve_kernel']:.3f} ms, {s['carve_steps']} steps ({s['carve_fast_steps']} fast), "
      f"{1e3 * s['ms_carve_kernel'] / s['carve_steps']:.2f} us/step")
for i, nm in enumerate(names):
    print(f"  {nm:12s} {out[i]:12d} ticks  {100.0 * out[i] / max(tot, 1):5.1f}%  {out[i] / max(s['carve_steps'], 1):8.1f} ticks/step")
print(f"  fast path (register-accumulated ticks): seed={out[0]} chain/same-site={out[1]} row+filter={out[2]} certificate={out[3]} commit={out[4]} loop-top={out[5]}")
print(f"  rounds: ticks={out[0]} rounds={out[1]} commits={out[2]} retries={out[3]} slow-stops={out[4]}; wave0 spec ticks={out[5]}; wave1 wait={out[6]} commit={out[7]} sync={out[8]}")
print(f"  rounds (wave1): chk={out[16]} b2wait={out[17]}; wave7: chk={out[18]} spec={out[19]}")
print(f"  counts(unused): prepares={out[8]} (sum n_list={out[7]}) refills={out[6]} launches={s['carve_launches']}")
print(f"  launch anatomy (ticks): init/status={out[15]} prepare(compaction)={out[9]} load-list={out[10]} run(all steps)={out[13]} flush={out[12]} group_of+exit={out[14]}")
print(f"  total ticks {tot}; ticks per ms = {tot / s['ms_carve_kernel']:.0f}")
