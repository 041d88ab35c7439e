#!/usr/bin/env python3
"""Where the spatial index starts to pay: cold-match carve p50 with and without it (PM_PRUNE_MODE=0: no index, every row
a bitmap sweep) for swarms between BASELINE configs[1] and configs[2]:  python tools/index_crossover.py [W ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys
sys.path.insert(0, %r)
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm
W = int(sys.argv[1]); reps = int(sys.argv[2])
sw = make_swarm(1, 10 * W, W, zipf=(W >= 100000))
eng = E.Engine(); host.load_swarm(eng, sw); eng.tick()
c = []
for _ in range(reps):
    eng.reset_groups(); s = eng.tick(); c.append(s["ms_carve"])
c.sort()
print("%%.3f %%d" %% (c[len(c) // 2], s["n_groups"]))
""" % ROOT

for W in [int(a) for a in sys.argv[1:]] or [10000, 15000, 20000, 30000, 50000]:
    row = []
    for mode in ("1", "0"):
        env = dict(os.environ, PM_PRUNE_MODE=mode)
        out = subprocess.run([sys.executable, "-c", CHILD, str(W), "12"], env=env, capture_output=True, text=True, timeout=300)
        row.append(out.stdout.strip().split("\n")[-1] if out.returncode == 0 else "failed: " + out.stderr[-200:])
    print(f"W={W}: carve p50 ms / groups  with index: {row[0]}   without: {row[1]}", flush=True)
