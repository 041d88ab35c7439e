#!/usr/bin/env python3
"""A ticket's way through the streaming carve, on ONE clock (a -DPM_ROW_REC build stamps everything with the real-time counter):
issued by the ticketer -> seen by a row maker -> row stored -> its block's rows all there (parker) -> parked in the chain's ring ->
taken up by a batch of the chain.  Per configuration: the latencies of the stages, what stands between the chain and its next
entry, and a time-sliced view of the window (tickets issued / taken up / rows stored / entries parked / consumed / commits).

    PM_EXP_LIB=protocol_amd/variants/libpm_engine_rowrec.so python tools/pipeline_probe.py [T W] [--slice US] [--configs 0,2]
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B
B.LIB_PATH = os.path.abspath(os.environ["PM_EXP_LIB"])
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

argv = sys.argv[1:]
def opt(name, default):
    if name in argv:
        i = argv.index(name)
        v = argv[i + 1]
        del argv[i:i + 2]
        return v
    return default
slice_us = float(opt("--slice", "20"))
only = opt("--configs", "")
only = [int(x) for x in only.split(",")] if only else None
T, W = (int(argv[0]), int(argv[1])) if len(argv) > 1 else (100000, 10000)
sw = make_swarm(1, T, W, zipf=(W >= 100000))
eng = E.Engine()
host.load_swarm(eng, sw)
for it in range(3):
    eng.reset_groups()
    s = eng.tick()
L = E.lib()
TICK = 100.0  # real-time counter ticks per microsecond
# rows
cap = 1 << 14
rows = np.zeros((cap, 8), dtype=np.uint64)
n = C.c_uint32(0)
L.pm_debug_row_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
assert L.pm_debug_row_records(eng._h, rows.ctypes.data, cap, C.byref(n)) == 0
rows = rows[: n.value]
# events
ecap = 1 << 15
ebuf = np.zeros((ecap, 2), dtype=np.uint64)
L.pm_debug_stream_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
assert L.pm_debug_stream_trace(eng._h, ebuf.ctypes.data, ecap, C.byref(n)) == 0
ebuf = ebuf[: n.value]
ev = sorted((int(t), int(w) & 0xFF, (int(w) >> 8) & 0xFFFFFF, int(w) >> 32) for t, w in ebuf)
# batches
bbuf = np.zeros((ecap, 4), dtype=np.uint64)
L.pm_debug_chain_batches.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
assert L.pm_debug_chain_batches(eng._h, bbuf.ctypes.data, ecap, C.byref(n)) == 0
bat = [(int(t), int(w) & 0xFFFFFF, (int(w) >> 24) & 0xFF, (int(w) >> 32) & 0xFF, int(w) >> 40, int(t2), int(t3)) for t, w, t2, t3 in bbuf[: n.value]]
pbuf = np.zeros((4096, 8), dtype=np.uint64)
L.pm_debug_park_records.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
assert L.pm_debug_park_records(eng._h, pbuf.ctypes.data, 4096, C.byref(n)) == 0
park = {}   # first ticket -> record
for rec in pbuf[: n.value]:
    w0 = int(rec[0])
    if w0 >> 63 and int(rec[5]):
        park[w0 & 0xFFFFFFFF] = dict(live_up=(w0 >> 32) & 0xFFFF, n_real=(w0 >> 48) & 0xFF, up=int(rec[1]), there=int(rec[2]), turn=int(rec[3]),
                                     room=int(rec[4]), parked=int(rec[5]), q=int(rec[6]) & 0xFFFFFFFF, live_pk=(int(rec[6]) >> 32) & 0xFFFF, b=int(rec[6]) >> 48)
if not ev or not len(rows):
    raise SystemExit("no records (not a PM_ROW_REC build)")
t0 = ev[0][0]
us = lambda t: (t - t0) / TICK
print(f"T={T} W={W}: carve kernel {s['ms_carve_kernel']:.3f} ms; {len(ev)} events, {len(bat)} batches of the chain, {int((rows[:, 0] != 0).sum())} rows; "
      f"events span {us(ev[-1][0]):.1f} us")
# ---- runs: [start event index, t0_run]; blocks and entries are numbered per run
cfg_of_time = [(e[0], e[2], e[3]) for e in ev if e[1] == 1]
runs = []
for i, e in enumerate(ev):
    if e[1] == 6:
        runs.append({"t": e[0], "t0_run": e[2], "cand": e[3], "takeup": {}, "there": {}, "parked": {}, "which": {}, "end": None, "commits": 0})
    elif e[1] == 7 and runs:
        runs[-1]["end"] = e[0]
        runs[-1]["commits"] = e[3]
for t0k, rec in park.items():   # a block belongs to the run it was parked in
    for r in runs:
        if r["t"] <= rec["parked"] and (r["end"] is None or rec["up"] <= r["end"]) and t0k == r["t0_run"] + rec["b"] * 16:
            b = rec["b"]
            r["takeup"][b] = (rec["up"], rec["live_up"], rec["n_real"])
            r["there"][b] = rec["there"]
            r["parked"][b] = (rec["parked"], rec["q"], bin(rec["live_pk"]).count("1"))
            r["which"][b] = rec
# ticket issue times: event 4 (t_req after, before)
issue = {}
for e in ev:
    if e[1] == 4:
        for t in range(e[3], e[2]):
            issue[t] = e[0]
# batches -> runs (by time)
for r in runs:
    r["batches"] = [b for b in bat if r["t"] <= b[0] and (r["end"] is None or b[0] <= r["end"])]
def cfg_at(t):
    c = None
    for tt, ci, nc in cfg_of_time:
        if tt <= t:
            c = (ci, nc)
    return c
pct = lambda a, q: float(np.percentile(a, q)) if len(a) else 0.0
def dist(name, a):
    a = np.asarray(a, dtype=np.float64)
    if not len(a):
        return f"{name}: none"
    return f"{name}: mean {a.mean():.1f}, p50 {pct(a, 50):.1f}, p90 {pct(a, 90):.1f}, max {a.max():.1f}"
for r in runs:
    ci, nc = cfg_at(r["t"])
    if only is not None and ci not in only:
        continue
    if not r["takeup"]:
        continue
    end = r["end"] or ev[-1][0]
    print(f"\n=== configuration {ci} ({nc} candidates at its entry): run from {us(r['t']):.1f} to {us(end):.1f} us, {r['commits']} commits, "
          f"{len(r['takeup'])} blocks of 16 tickets taken up, {len(r['batches'])} batches of the chain")
    # per ticket with a row asked for
    q_issue_seen, seen_end, end_there, there_parked, parked_used, issue_takeup, total = [], [], [], [], [], [], []
    slowest_is_last = 0
    late_by = []
    ent_time = {}   # entry -> parked time
    for b, (t_up, live_m, n_real) in sorted(r["takeup"].items()):
        tks = [r["t0_run"] + b * 16 + k for k in range(16) if (live_m >> k) & 1]
        t_there = r["there"].get(b)
        pk = r["parked"].get(b)
        ends = []
        for t in tks:
            if t < len(rows) and rows[t, 0]:
                seen, endt = int(rows[t, 0]), int(rows[t, 5])
                if t in issue:
                    q_issue_seen.append((seen - issue[t]) / TICK)
                    issue_takeup.append((t_up - issue[t]) / TICK)
                seen_end.append((endt - seen) / TICK)
                ends.append(endt)
                if t_there:
                    end_there.append((t_there - endt) / TICK)
        if t_there and ends:
            late_by.append((max(ends) - t_up) / TICK)   # how long after take-up the block's last row was stored (< 0: all there already)
        if t_there and pk:
            there_parked.append((pk[0] - t_there) / TICK)
        if pk:
            for j in range(pk[2]):
                ent_time[pk[1] + j] = pk[0]
    used = {}
    for (tb, tail0, n_steps, live, commits, _t2, _t3) in r["batches"]:
        for q in range(tail0, tail0 + n_steps):
            used.setdefault(q, tb)
    parked_used = [(used[q] - ent_time[q]) / TICK for q in ent_time if q in used]
    print("  a ticket whose row was asked for (us):")
    print("    " + dist("issued -> seen by a row maker", q_issue_seen))
    print("    " + dist("seen -> row stored", seen_end))
    print("    " + dist("issued -> its block taken up by a parker", issue_takeup))
    print("    " + dist("row stored -> the block's rows all there", end_there))
    print("    " + dist("block taken up -> its last row stored (negative: it was there)", late_by))
    print("    " + dist("rows there -> entries parked (turn, room)", there_parked))
    print("    " + dist("parked -> taken up by a batch of the chain", parked_used))
    # a parker's cycle, block by block: taken up -> rows there -> parked -> (its next block, NPARK further on) taken up
    up_there = [(v["there"] - v["up"]) / TICK for v in r["which"].values()]
    there_turn = [(v["turn"] - v["there"]) / TICK for v in r["which"].values()]
    turn_room = [(v["room"] - v["turn"]) / TICK for v in r["which"].values()]
    room_pk = [(v["parked"] - v["room"]) / TICK for v in r["which"].values()]
    NP = 5
    pk_next = [(r["takeup"][b + NP][0] - r["parked"][b][0]) / TICK for b in r["parked"] if b + NP in r["takeup"]]
    cyc = [(r["takeup"][b + NP][0] - r["takeup"][b][0]) / TICK for b in r["takeup"] if b + NP in r["takeup"]]
    pks = sorted(v[0] for v in r["parked"].values())
    print("  a parker's cycle (us):")
    print("    " + dist("block taken up -> its rows there", up_there))
    print("    " + dist("rows there -> its turn", there_turn))
    print("    " + dist("turn -> room in the ring", turn_room))
    print("    " + dist("room -> parked", room_pk))
    turns = sorted(v["turn"] for v in r["which"].values())
    if len(turns) > 1:
        print("    " + dist("from one block's turn to the next block's", np.diff(turns) / TICK))
    print("    " + dist("parked -> the parker's next block taken up (its tickets issued?)", pk_next))
    print("    " + dist("the whole cycle", cyc))
    if len(pks) > 1:
        print("    " + dist("from one block's parking to the next block's", np.diff(pks) / TICK))
    nb = len(r["batches"])
    if nb:
        ents = sum(b[2] for b in r["batches"])
        lives = sum(b[3] for b in r["batches"])
        print(f"  the chain: {nb} batches, {ents / nb:.1f} entries a batch ({lives / nb:.1f} alive at the batch's head), "
              f"{(end - r['t']) / TICK / max(r['commits'], 1):.3f} us per commit over the run")
        gaps = np.diff([b[0] for b in r["batches"]]) / TICK
        # what a batch costs the chain when it is not waiting: the batches behind which the next one found a full ring (16 entries)
        B_ = r["batches"]
        fit = [(gaps[i], B_[i + 1][4] - B_[i][4], B_[i][3] - (B_[i + 1][4] - B_[i][4]), B_[i][2] - B_[i][3]) for i in range(len(B_) - 1) if B_[i + 1][2] == 16]
        if len(fit) >= 8:
            A = np.array([[1.0, f[1], f[2], f[3]] for f in fit])
            y = np.array([f[0] for f in fit])
            coef, *_ = np.linalg.lstsq(A, y, rcond=None)
            print(f"    {len(fit)} batches followed by a full one: {y.mean():.2f} us each for {A[:, 1].mean():.1f} commits, {A[:, 2].mean():.1f} seeds that died inside the batch, "
                  f"{A[:, 3].mean():.1f} dead at its head; least squares: {coef[0]:.2f} us + {coef[1]:.3f} a commit + {coef[2]:.3f} a seed that died inside + {coef[3]:.3f} a dead entry")
        full = [i for i in range(len(B_) - 1) if B_[i + 1][2] == 16 and B_[i][2] == 16]
        if full:
            seg = lambda f: np.mean([f(i) for i in full]) / TICK
            print(f"    full batches followed by full ones ({len(full)}): loop top -> entries looked at {seg(lambda i: B_[i][5] - B_[i][0]):.2f} us, -> steps done "
                  f"{seg(lambda i: B_[i][6] - B_[i][5]):.2f}, -> next loop top {seg(lambda i: B_[i + 1][0] - B_[i][6]):.2f}")
        small = [g for g, b in zip(gaps, r["batches"][:-1]) if b[2] <= 4]
        print(f"    time from a batch's head to the next one's: mean {gaps.mean():.2f} us; batches of <= 4 entries: {len(small)} of {nb}"
              + (f" (mean {np.mean(small):.2f} us)" if small else ""))
    # time slices
    print(f"  slices of {slice_us:.0f} us: tickets issued | taken up by parkers | rows stored | entries parked | entries consumed | commits   "
          f"(window = issued - taken up; rows being made = seen - stored)")
    tk_lo = r["t0_run"]
    tk_hi = max((r["t0_run"] + (b + 1) * 16 for b in r["takeup"]), default=tk_lo)
    iss = np.array(sorted(issue[t] for t in range(tk_lo, tk_hi + 4096) if t in issue and issue[t] <= end + 1))
    ups = np.array(sorted(v[0] for v in r["takeup"].values()))
    tr = rows[tk_lo: min(tk_hi + 4096, len(rows))]
    tr = tr[tr[:, 0] != 0]
    seen_a = np.sort(tr[:, 0].astype(np.int64))
    end_a = np.sort(tr[:, 5].astype(np.int64))
    pk_t = sorted((v[0], v[1] + v[2]) for v in r["parked"].values())
    bt = r["batches"]
    tt = r["t"]
    while tt < end:
        te = tt + slice_us * TICK
        c_iss = int(np.searchsorted(iss, te))
        c_up = int(np.searchsorted(ups, te)) * 16
        c_seen = int(np.searchsorted(seen_a, te))
        c_end = int(np.searchsorted(end_a, te))
        c_pk = max((q for t_, q in pk_t if t_ <= te), default=0)
        c_used = max((b[1] + b[2] for b in bt if b[0] <= te), default=0)
        c_com = max((b[4] for b in bt if b[0] <= te), default=bt[0][4] if bt else 0) - (bt[0][4] if bt else 0)
        print(f"    {us(te):8.1f}: {c_iss:6d} | {c_up:6d} | {c_end:6d} | {c_pk:6d} | {c_used:6d} | {c_com:6d}    window {c_iss - c_up:5d}, rows being made {c_seen - c_end:4d}, ring {c_pk - c_used:3d}")
        tt = te
eng.close()
