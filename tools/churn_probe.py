#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU, the stream of bench.py's `churn` sub-object (protocol_amd/churn.py): the cold match,
then `ticks` incremental ticks.  For kernel traces (tools/collect_profiles.py runs it under rocprofv3): prints the
per-tick split.  usage: python tools/churn_probe.py [ticks]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import build as B
if os.environ.get("PM_EXP_LIB"):  # a variant built beforehand (tools/build_variants.py)
    B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
    B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.churn import ChurnStream

ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cs = ChurnStream(1, 8)                                     # (the swarm bench.py and the golden digests use)
sw_all = cs.sw_all
packed = host.pack_workers(sw_all)
rows = lambda idx: {k: np.ascontiguousarray(v[idx]) for k, v in packed.items()}
eng = E.Engine()
cfg_rows, alt_rows, req_models = host.pack_configs(sw_all.configs)
eng.set_configs(cfg_rows, alt_rows)
eng.set_model_table(host.build_model_table(req_models, sw_all.model_names), len(req_models), len(sw_all.model_names))
eng.upload_workers(rows(np.arange(cs.W0)))
eng.upload_tasks(cs.masks, cs.created, cs.uid)
eng.set_enabled_mask(sw_all.enabled_mask())
s0 = eng.tick()
print(f"cold match: {s0['ms_total']:.2f} ms, {s0['n_groups']} groups")
flags = packed["flags"].astype(np.int64)
for t in range(min(ticks, 8)):
    leave, idx_new, new_tasks = cs.step()
    eng.on_worker_status_many(leave, flags[leave] & ~E.W_HEALTHY, np.ones(len(leave), dtype=np.uint32))
    eng.append_workers(rows(idx_new))
    eng.tasks_insert_front(*new_tasks[:3])
    t0 = time.perf_counter()
    s = eng.tick()
    print(f"tick {t}: {1e3 * (time.perf_counter() - t0):.2f} ms (carve {s['ms_carve']:.2f}, sweep {s['ms_sweep']:.2f}, "
          f"publish {s['ms_publish']:.2f}), {s['n_formed']} formed, {s['carve_launches']} carve launches")
eng.close()
