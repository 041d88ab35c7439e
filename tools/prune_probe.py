#!/usr/bin/env python3
"""What every proposal batch of a full-swarm carve looked like (list length, seeds, walked the spatial index or swept the
whole list), from a PM_BATCH_LOG build.  Run it under `rocprofv3 --kernel-trace` and join with tools/prune_join.py to
see what each batch's proposer launch cost: usage  prune_probe.py <config> <prune mode> <out.json>"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import build as B
alt = os.path.join(os.path.dirname(B.LIB_PATH), "libpm_engine_exp.so")
if not os.path.exists(alt) or os.environ.get("PM_EXP_REBUILD"):
    B.build(force=True, defines=["PM_BATCH_LOG"] + [d for d in os.environ.get("PM_EXP_DEFINES", "").split(",") if d], out=alt)
B.LIB_PATH = alt
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

ci, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
sw = baseline_config(ci, seed=1)
eng = E.Engine()
eng.debug_prune_mode(mode)
host.load_swarm(eng, sw)
eng.tick()
ms = []
for _ in range(3):
    eng.reset_groups()
    s = eng.tick()
    ms.append(s["ms_carve"])
buf = (C.c_uint32 * 1536)()
n = C.c_uint32(0)
E.lib().pm_debug_batch_log.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(C.c_uint32)]
E.check(E.lib().pm_debug_batch_log(eng._h, buf, 1536, C.byref(n)))
log = [[int(buf[3 * k]), int(buf[3 * k + 1]), int(buf[3 * k + 2])] for k in range(n.value // 3)]
prof = (C.c_ulonglong * 32)()
E.lib().pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
E.lib().pm_debug_carve_prof(eng._h, prof, 32)
walks = int(prof[2])
if walks:
    print(f"  walks {walks}: mean {prof[0] / walks:.0f} ticks, max {int(prof[1])} ticks, candidates evaluated per walk {prof[3] / walks:.0f}; "
          f"stopped in front of ring (0 = gave up): " + " ".join(f"{k}:{int(prof[16 + k])}" for k in range(16) if prof[16 + k]))
if prof[13]:
    nb = int(prof[13])
    print(f"  placement kernel over {nb} preparations (ticks): first block start -> every block through {prof[11] / nb:.0f}, "
          f"tail of the last block {prof[12] / nb:.0f}; one block start -> ticket: mean {prof[8] / max(int(prof[10]), 1):.0f}, max {int(prof[9])} "
          f"(loads in at {prof[14] / max(int(prof[10]), 1):.0f}, stores issued at {prof[15] / max(int(prof[10]), 1):.0f})")
json.dump({"config": ci, "mode": mode, "carve_ms": sorted(ms), "counters": eng.debug_carve_counters(), "batches": log},
          open(out, "w"))
print(f"config {ci} mode {mode}: carve {sorted(ms)[1]:.3f} ms, {len(log)} preparations, counters {eng.debug_carve_counters()}")
