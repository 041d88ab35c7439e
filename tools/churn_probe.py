#!/usr/bin/env python3
"""BASELINE configs[4] on one GPU: 100k-worker swarm, per tick 1 % of the workers leave and 1 % join, 10k new
tasks arrive (newest first, the oldest drop out), then one incremental match (`pm_tick` on the standing groups).
Prints the per-tick latency split: status updates, task upload, match."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config

W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sw = baseline_config(2, seed=5)
rng = np.random.default_rng(1)
late = rng.random(sw.W) < 0.10                      # 10 % of the workers start offline and join over time
status0 = sw.status.copy()
sw.status = np.where(late, 0, sw.status).astype(np.uint8)
eng = E.Engine()
host.load_swarm(eng, sw)
flags = host.worker_flags(sw).astype(np.int64)
masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
s0 = eng.tick()
print(f"cold match: {s0['ms_total']:.2f} ms, {s0['n_groups']} groups")
joiners = list(np.nonzero(late & (status0 == 2))[0])
alive = set(np.nonzero(sw.status == 2)[0].tolist())
next_uid = 1 << 40
n_churn = sw.W // 100
n_new = 10000
for t in range(ticks):
    t0 = time.perf_counter()
    leave = rng.choice(np.fromiter(alive, dtype=np.int64), size=n_churn, replace=False)
    for w in leave:
        flags[w] &= ~E.W_HEALTHY
        eng.on_worker_status(int(w), int(flags[w]), True)
        alive.discard(int(w))
    for _ in range(min(n_churn, len(joiners))):
        w = int(joiners.pop())
        flags[w] |= E.W_HEALTHY
        eng.on_worker_status(w, int(flags[w]), False)
        alive.add(w)
    t1 = time.perf_counter()
    pick = rng.integers(0, len(masks), n_new)
    masks = np.concatenate([masks[pick], masks[:-n_new]])
    created = np.concatenate([int(created.max()) + 1 + np.arange(n_new)[::-1], created[:-n_new]])
    uid = np.concatenate([np.arange(next_uid, next_uid + n_new, dtype=np.uint64), uid[:-n_new]])
    next_uid += n_new
    eng.upload_tasks(masks, created, uid)
    t2 = time.perf_counter()
    s = eng.tick()
    t3 = time.perf_counter()
    print(f"tick {t}: status updates {1e3 * (t1 - t0):7.2f} ms ({2 * n_churn} calls)  task upload {1e3 * (t2 - t1):6.2f} ms  "
          f"match {1e3 * (t3 - t2):6.2f} ms (carve {s['ms_carve']:.2f}, sweep {s['ms_sweep']:.2f}, publish {s['ms_publish']:.2f}; "
          f"formed {s['n_formed']}, groups {s['n_groups']}, exact steps {s['carve_steps'] - s['carve_fast_steps']})")
eng.close()
