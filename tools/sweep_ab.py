#!/usr/bin/env python3
"""The pair sweep of cold configs[2] matches (1M tasks x 100k workers) of a variant library beside the product's:
PM_EXP_LIB=protocol_amd/variants/libpm_engine_<name>.so python tools/sweep_ab.py [config] [reps]  (ms_sweep / its kernel, p50)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import build as B
if os.environ.get("PM_EXP_LIB"):
    B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
    B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
sw = baseline_config(ci, seed=1)
eng = E.Engine(time_proposer=True)   # (the sweep kernel's own events)
host.load_swarm(eng, sw)
eng.tick()
rows = []
for _ in range(reps):
    eng.reset_groups()
    s = eng.tick()
    rows.append((s["ms_sweep"], s.get("ms_sweep_kernel", 0.0), s["ms_publish"], s["ms_total"]))
rows.sort()
m = rows[len(rows) // 2]
print(f"config {ci} lib {os.environ.get('PM_EXP_LIB', 'product')}: sweep p50 {m[0]:.3f} ms (kernel {m[1]:.3f}), publish {m[2]:.3f}, match {m[3]:.3f}, groups {s['n_groups']}")
