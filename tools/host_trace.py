#!/usr/bin/env python3
"""Host-side marks of one cold match at BASELINE configs[1] (PM_TRACE_HOST=1: pm_engine.cpp host_mark, microseconds on
stderr), for reading next to the kernel timeline (tools/tick_timeline.py): where the HOST is while the GPU idles.
usage: PM_TRACE_HOST=1 python tools/host_trace.py [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PM_TRACE_HOST", "1")
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

sw = baseline_config(int(sys.argv[1]) if len(sys.argv) > 1 else 1, seed=1)
eng = E.Engine()
host.load_swarm(eng, sw)
for it in range(4):
    eng.reset_groups()
    sys.stderr.write(f"[pm host] ---- match {it}\n")
    s = eng.tick()
print(f"match {s['ms_total']:.3f} ms (compat {s['ms_compat']:.3f}, carve {s['ms_carve']:.3f}, sweep {s['ms_sweep']:.3f}, publish {s['ms_publish']:.3f})")
