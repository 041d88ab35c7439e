#!/usr/bin/env python3
"""Anatomy of the streaming carve (carve_variant 0): builds a PM_CARVE_PROF variant of the library (s_memtime ticks
accumulated per phase) and runs cold full-swarm matches.

    python tools/stream_prof.py [T W]          default 100000 10000 (BASELINE configs[1]); 1000000 100000 = configs[2]

What the validator's workgroup did (chain wave / producer wave), what a row cost the proposer waves by the way it was
made, and what the configuration boundaries cost.  Environment: PM_STREAM_WGS, PM_STREAM_LA, PM_STREAM_LA_DIV.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from protocol_amd import build as B

prof_lib = os.environ.get("PM_PROF_LIB") or os.path.join(ROOT, "protocol_amd", "libpm_engine_prof.so")
if not os.environ.get("PM_PROF_NO_BUILD"):
    B.build(force=True, defines=["PM_CARVE_PROF"] + os.environ.get("PM_EXTRA_DEFINES", "").split(), out=prof_lib)
B.LIB_PATH = prof_lib
B.needs_build = lambda: False
from protocol_amd import engine as E, host
from protocol_amd.swarm import make_swarm

T, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 10000)
sw = make_swarm(1, T, W, zipf=(W >= 100000))
eng = E.Engine()
host.load_swarm(eng, sw)
for it in range(3):
    eng.reset_groups()
    s = eng.tick()
out = (C.c_ulonglong * 88)()
E.lib().pm_debug_carve_prof.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_uint32]
E.lib().pm_debug_carve_prof(eng._h, out, 88)
o = [int(v) for v in out]
info = eng.debug_carve_counters()
steps = max(s["carve_steps"], 1)
# s_memtime ticks per microsecond: the validator's configurations-run span is (nearly) the carve launch
tpu = max(o[58], 1) / max(1e3 * s["ms_carve_kernel"] - 40.0, 1.0)
us = lambda t: t / tpu
print(f"T={T} W={W}: carve {s['ms_carve_kernel']:.3f} ms, {s['carve_steps']} steps ({s['carve_fast_steps']} from rows), "
      f"{1e3 * s['ms_carve_kernel'] / steps:.3f} us/step, {s['carve_launches']} launches, match {s['ms_total']:.3f} ms")
print(f"  stream: tickets {info['stream_tickets']} ({info['stream_tickets'] / steps:.2f} per group), timeouts {info['stream_timeouts']}, "
      f"walk->sweep switches {info['stream_switches']}, configurations swept {info['stream_listed']}, started on tickets issued ahead "
      f"{info['stream_pre_used']} (tickets ahead lost {info['stream_pre_lost']}), proposer workgroups {info['stream_wgs']}, "
      f"exact steps {info['slow_steps']}, index grid {info['cell_g']} ({info['n_indexed']} positions), walks given up {info['prune_fallbacks']}")
print(f"  validator, per pass (us): configuration entry (bitmaps) {us(o[56]):.1f}, next configuration + ticket state {us(o[57]):.1f}, "
      f"configurations run {us(o[58]):.1f}")
print(f"  wave 0 (us): chain {us(o[0]):.1f} over {o[1]} calls, {o[2]} commits ({o[0] / max(o[2], 1):.0f} ticks per commit), {o[4]} hand-overs; "
      f"wave 0 in total {us(o[11]):.1f}; exact steps {us(o[22]):.1f}")
print(f"  chain anatomy (us): waiting for rows {us(o[16]):.1f} ({o[24]} waits), steps {us(o[18]):.1f}, stopping the other waves {us(o[17]):.1f}; "
      f"{o[19]} batches of steps, {o[23]} seeds dead at their turn")
if os.environ.get("PM_CHAIN_FINE") == "1":   # (a -DPM_CARVE_PROF,PM_CHAIN_FINE build: the chain's counters in the parkers' slots)
    print(f"  the steps' time (us): batch heads {us(o[5]):.1f} ({us(o[5]) / max(o[19], 1):.3f} a batch), plain steps {us(o[6]):.1f} "
          f"({us(o[6]) / max(o[2], 1):.3f} a commit), steps that needed attention {us(o[7]):.1f} ({o[8]} of them), batch tails {us(o[18]):.1f}")
print(f"  the parkers together (us): waiting for tickets {us(o[5]):.1f}, for room {us(o[6]):.1f}, for rows {us(o[7]):.1f} "
      f"({o[26]} polls, {o[27]} of {o[28]} rows late at the first look), digesting {us(o[8]):.1f}, idle between runs {us(o[9]):.1f}")
row = lambda t, n: f"{n} rows, {us(t) / max(n, 1):.2f} us each" if n else "none"
print(f"  proposer rows: bitmap sweep {row(o[10], o[12])}; index walk {row(o[13], o[14])}; position sweep {row(o[15], o[29])}; "
      f"slowest row {us(o[30]):.1f} us")
print(f"  a row on average: ticket seen -> candidates swept {us(o[59]) / max(o[12] + o[14] + o[29], 1):.2f} us, finished and written "
      f"{us(o[60]) / max(o[12] + o[14] + o[29], 1):.2f} us, {o[61] / max(o[12] + o[14] + o[29], 1):.0f} candidates evaluated")
print(f"  bitmap sweeps: {o[62 + 2]} passes {us(o[62]) / max(o[62 + 2], 1):.2f} us each, {o[62 + 3]} batches of 256 {us(o[62 + 1]) / max(o[62 + 3], 1):.2f} us each")
n_rows = max(o[12] + o[14] + o[29], 1)
print(f"  a row's compute (us, count per row): sorted insertions {us(o[72]) / n_rows:.2f} ({o[73] / n_rows:.0f}), near-miss tracker "
      f"{us(o[74]) / n_rows:.2f} ({o[75] / n_rows:.0f}), sine-form keys {us(o[76]) / n_rows:.2f} ({o[77] / n_rows:.1f} strides), evictions with a site "
      f"look-up {o[78] / n_rows:.1f}, strides offered {o[79] / n_rows:.1f}; bitmap-sweep batches: waiting for the gathers "
      f"{us(o[80]) / max(o[12], 1):.2f}, evaluating {us(o[81]) / max(o[12], 1):.2f} per swept row (the keys {us(o[82]) / max(o[12], 1):.2f}, "
      f"the row {us(o[83]) / max(o[12], 1):.2f})")
print(f"  the networks' calls: {us(o[87]) / n_rows:.2f} us per row in all; inside: counting + packing {us(o[84]) / n_rows:.2f}, "
      f"threshold + evictions {us(o[85]) / n_rows:.2f}, near misses {us(o[86]) / n_rows:.2f} (the networks themselves: the sorted insertions above)")
print(f"  stream_small: {o[66 + 5]} groups; per group (us): seed {us(o[66]) / max(o[71], 1):.2f}, keys {us(o[67]) / max(o[71], 1):.2f}, "
      f"selection {us(o[68]) / max(o[71], 1):.2f}, certificate {us(o[69]) / max(o[71], 1):.2f}, commit {us(o[70]) / max(o[71], 1):.2f}")
print(f"  exact-sweep reasons: no row {o[20]}, debug hook {o[21]}, row exhausted {o[25]}, certificate {o[31]}")
eng.close()
