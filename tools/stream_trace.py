#!/usr/bin/env python3
"""Timeline of one streaming carve launch (PM_CARVE_PROF build, pm_debug_stream_trace): where the chain waits, and for what.

    python tools/stream_trace.py [T W] [--dump FILE]

Per configuration: when it was entered, when its run started, when the first row was parked, the chain's waits (start,
length, entry), how the rows arrived.  Times in microseconds from the first event (s_memtime ticks / ticks-per-us, the
latter calibrated against the carve's hipEvent time).
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B

prof_lib = os.environ.get("PM_PROF_LIB") or os.path.join(ROOT, "protocol_amd", "libpm_engine_prof.so")
if not os.environ.get("PM_PROF_NO_BUILD"):
    B.build(force=True, defines=["PM_CARVE_PROF"], out=prof_lib)
B.LIB_PATH = prof_lib
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

args = [a for a in sys.argv[1:] if not a.startswith("--")]
dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
if "--dump" in sys.argv:
    args = [a for a in args if a != dump]
if args and args[0] == "churn":  # the incremental tick of BASELINE configs[4] (protocol_amd/churn.py), third tick
    import numpy as np
    from protocol_amd.churn import ChurnStream
    cs = ChurnStream(1, 8)
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
    eng.tick()
    flags = packed["flags"].astype(np.int64)
    for t in range(3):
        leave, idx_new, new_tasks = cs.step()
        eng.on_worker_status_many(leave, flags[leave] & ~E.W_HEALTHY, np.ones(len(leave), dtype=np.uint32))
        eng.append_workers(rows(idx_new))
        eng.tasks_insert_front(*new_tasks[:3])
        s = eng.tick()
    T, W = "churn", cs.W0
else:
    T, W = (int(args[0]), int(args[1])) if len(args) > 1 else (100000, 10000)
    sw = make_swarm(1, T, W, zipf=(W >= 100000))
    eng = E.Engine()
    host.load_swarm(eng, sw)
    for it in range(3):
        eng.reset_groups()
        s = eng.tick()
cap = 1 << 17
buf = (C.c_ulonglong * (2 * cap))()
n = C.c_uint32(0)
L = E.lib()
L.pm_debug_stream_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32, C.POINTER(C.c_uint32)]
L.pm_debug_stream_trace.restype = C.c_int32
assert L.pm_debug_stream_trace(eng._h, buf, cap, C.byref(n)) == 0
ev = sorted((int(buf[2 * i]), int(buf[2 * i + 1]) & 0xFF, (int(buf[2 * i + 1]) >> 8) & 0xFFFFFF, int(buf[2 * i + 1]) >> 32) for i in range(n.value))
if not ev:
    raise SystemExit("no events (not a PM_CARVE_PROF build, or the carve did not stream)")
rows_ev = [e for e in ev if e[1] == 8]   # (written by other CUs: their clocks are not the validator's)
seg_ev = [e for e in ev if 20 <= e[1] <= 25]  # (a swept row's segments: see carve_stream_kernel's row makers)
ev = [e for e in ev if e[1] != 8 and not 20 <= e[1] <= 25]
t0 = ev[0][0]
span = ev[-1][0] - t0
tpu = span / max(1e3 * s["ms_carve_kernel"] - 45.0, 1.0)
us = lambda t: (t - t0) / tpu
print(f"T={T} W={W}: carve {s['ms_carve_kernel']:.3f} ms, {n.value} events over {span} ticks ({tpu:.0f} ticks per us)")
if dump:
    names = {1: "config", 2: "wait", 3: "go", 4: "tickets", 5: "parked", 6: "run", 7: "end", 8: "row", 9: "exact", 10: "fast", 11: "probe", 12: "tail", 13: "tiny", 14: "ranout", 15: "small", 16: "rowwait", 17: "slots_in", 18: "slots_out", 19: "slot_at", 26: "ack"}
    with open(dump, "w") as f:
        for t, ty, a, b in ev:
            f.write(f"{us(t):10.2f} {str(names.get(ty, ty)):8s} {a:8d} {b:10d}\n")
    with open(dump + ".segs", "w") as f:   # per swept row: ticket, us: seed columns, passes, gather waits, evaluation, rest of the batches, finish
        segs = {}
        for t, ty, a, b in seg_ev:
            segs.setdefault(a, [0.0] * 6)[ty - 20] = b / tpu
        for a in sorted(segs):
            f.write(f"{a:8d} " + " ".join(f"{v:8.2f}" for v in segs[a]) + "\n")
    with open(dump + ".rows", "w") as f:   # (other CUs' clocks: ticket, ticks the row took its wave)
        for t, ty, a, b in rows_ev:
            f.write(f"{a:8d} {b / tpu:10.2f}\n")
# per configuration
cfgs = [i for i, e in enumerate(ev) if e[1] == 1]
cfgs.append(len(ev))
tot_wait = 0.0
print("  cfg  cand   enter_us  len_us  runs commits  wait_us (n)  first_wait_us  rows_in_window  row_us(avg)  tickets")
for k in range(len(cfgs) - 1):
    seg = ev[cfgs[k]:cfgs[k + 1]]
    t_in = seg[0][0]
    ci, n_cand = seg[0][2], seg[0][3]
    runs = [e for e in seg if e[1] == 6]
    ends = [e for e in seg if e[1] == 7]
    commits = sum(e[3] for e in ends)
    waits = []
    w0 = None
    for e in seg:
        if e[1] == 2:
            w0 = e[0]
        elif e[1] == 3 and w0 is not None:
            waits.append((w0, e[0] - w0))
            w0 = None
    wsum = sum(d for _, d in waits) / tpu
    tot_wait += wsum
    first_wait = waits[0][1] / tpu if waits else 0.0
    tick = [e for e in seg if e[1] == 4]
    issued = (tick[-1][2] - tick[0][3]) if tick else 0
    lo, hi = (tick[0][3], tick[-1][2]) if tick else (0, 0)
    rows = [e for e in rows_ev if lo <= e[2] < hi]
    row_avg = sum(e[3] for e in rows) / tpu / max(len(rows), 1)
    print(f"  {ci:3d} {n_cand:5d} {us(t_in):10.1f} {(seg[-1][0] - t_in) / tpu:7.1f} {len(runs):5d} {commits:7d} {wsum:8.1f} ({len(waits):3d}) "
          f"{first_wait:10.1f} {len(rows):12d} {row_avg:12.1f} {issued:8d}")
print(f"  chain waits in total: {tot_wait:.1f} us")
# the end of every configuration: chain over -> located rest in registers (stream_small) -> first-come tail -> next configuration
print("  cfg   chain_over_us  to_small_us  small_us (groups)  first_come_us  to_next_config_us")
tot = [0.0, 0.0, 0.0, 0.0]
for k in range(len(cfgs) - 1):
    seg = ev[cfgs[k]:cfgs[k + 1]]
    nxt = ev[cfgs[k + 1]][0] if cfgs[k + 1] < len(ev) else seg[-1][0]
    ends = [e for e in seg if e[1] in (7, 9, 10)]
    t_over = ends[-1][0] if ends else seg[0][0]
    sm_in = [e for e in seg if e[1] == 15]
    sm_out = [e for e in seg if e[1] == 13]
    tl = [e for e in seg if e[1] == 12]
    t_a = sm_in[-1][0] if sm_in else t_over
    t_b = sm_out[-1][0] if sm_out else t_a
    t_c = tl[-1][0] if tl else t_b
    g0 = ends[-1][3] if (ends and ends[-1][1] != 7) else None
    d = [(t_a - t_over) / tpu, (t_b - t_a) / tpu, (t_c - t_b) / tpu, (nxt - t_c) / tpu]
    tot = [x + y for x, y in zip(tot, d)]
    print(f"  {seg[0][2]:3d} {us(t_over):14.1f} {d[0]:12.1f} {d[1]:9.1f} {d[2]:14.1f} {d[3]:18.1f}")
print(f"  totals (us): to_small {tot[0]:.1f}, small {tot[1]:.1f}, first_come {tot[2]:.1f}, to_next_config {tot[3]:.1f}")
# the longest waits
allw = []
w0 = None
for e in ev:
    if e[1] == 2:
        w0 = e
    elif e[1] == 3 and w0 is not None:
        allw.append(((e[0] - w0[0]) / tpu, us(w0[0]), w0[2], w0[3]))
        w0 = None
allw.sort(reverse=True)
print("  longest waits (us, at_us, entry, commits so far in the run):")
for w in allw[:16]:
    print(f"    {w[0]:7.1f} at {w[1]:9.1f}  entry {w[2]:5d}  commits {w[3]:5d}")
eng.close()
