#!/usr/bin/env python3
"""Per-batch timing of the carve through the stepwise tick (one rank): seeds of the batch, proposer time, validation
(+ preparation of the next list) time.  usage: python tools/batch_probe.py [config index]
PM_EXP_DEFINES=A=1,B builds and uses a variant library with those -D flags (e.g. PROP_TILE=1024u,
PM_PROP_CAP_DIV_BIG=8u) — how the tile sizes and batch caps quoted in DESIGN.md were compared."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("PM_EXP_DEFINES"):  # experiment builds: a variant library with extra -D flags
    from protocol_amd import build as B
    alt = os.path.join(os.path.dirname(B.LIB_PATH), "libpm_engine_exp.so")
    B.build(force=True, defines=os.environ["PM_EXP_DEFINES"].split(","), out=alt)
    B.LIB_PATH = alt
    B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sw = baseline_config(ci, seed=1)
eng = E.Engine()
host.load_swarm(eng, sw)
eng.tick()
for rep in range(2):
    eng.reset_groups()
    rows = []
    eng.dist_tick_begin()
    while True:
        x, more = eng.dist_carve_next()          # (waits for the previous validation; launches the proposer)
        if not more:
            break
        t1 = time.perf_counter()
        torch.cuda.synchronize()                 # the proposer of this batch
        t2 = time.perf_counter()
        eng.dist_carve_validate()
        torch.cuda.synchronize()                 # validation + preparation of the next list
        t3 = time.perf_counter()
        rows.append((int(x.bytes_per_rank) // 512, t2 - t1, t3 - t2))
    eng.dist_match_begin()
    s = eng.dist_tick_end()
print(f"{len(rows)} batches, {s['n_groups']} groups; carve {s['ms_carve']:.2f} ms")
print("seeds per batch:", [r[0] for r in rows])
print("propose us:      ", [int(r[1] * 1e6) for r in rows], "sum", int(sum(r[1] for r in rows) * 1e6))
print("validate+prep us:", [int(r[2] * 1e6) for r in rows], "sum", int(sum(r[2] for r in rows) * 1e6))
