import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
t0 = time.time(); sw = baseline_config(2, seed=1, scale=scale); print("gen", time.time() - t0, sw.T, sw.W, flush=True)
eng = E.Engine()
t0 = time.time(); host.load_swarm(eng, sw); print("load", time.time() - t0, flush=True)
for it in range(2):
    eng.reset_groups()
    t0 = time.time(); s = eng.tick(); dt = time.time() - t0
    print(f"tick {dt*1e3:.1f} ms wall; carve {s['ms_carve']:.1f} sweep {s['ms_sweep']:.3f} compat {s['ms_compat']:.3f} publish {s['ms_publish']:.3f}; groups {s['n_groups']} steps {s['carve_steps']} fast {s['carve_fast_steps']} host-resolved {s['host_resolved_steps']}", flush=True)
