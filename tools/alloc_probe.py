#!/usr/bin/env python3
"""Does a match's time depend on WHERE the engine's buffers landed?  Engines made one after the other in one process (a dummy
allocation of another size in front of each, so that the allocator hands out other addresses), the same swarm, the carve of each
timed over a few cold matches:  python tools/alloc_probe.py [config] [engines] [matches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_eng = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
sw = baseline_config(ci, seed=1)
keep = []
for k in range(n_eng):
    keep.append(torch.empty((1 + 37 * k) * 4096 + 123 * k, dtype=torch.uint8, device="cuda"))  # shifts what comes next
    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.tick()
    carve, kern = [], []
    for _ in range(reps):
        eng.reset_groups()
        s = eng.tick()
        carve.append(s["ms_carve"])
        kern.append(s["ms_carve_kernel"])
    print(f"engine {k}: carve ms " + " ".join(f"{c_:.3f}" for c_ in carve) + f"   kernel p50 {sorted(kern)[len(kern) // 2]:.3f}")
    eng.close()
