#!/usr/bin/env python3
"""Runs tools/microbench/chain_step.hip (build it first, where hipcc is: see its head) and prints cycles per step."""
import ctypes as C
import os
import sys
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "libchain_step.so"))
L.chain_step_run.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
names = {0: "the run as it was (37 instructions)", 1: "without the collector's word", 3: "without the kill", 4: "the look alone", 5: "two steps a trip", 7: "... and the look waited for alone", 8: "7 without the collector's word", 9: "7 without the attention test", 10: "7, the test without its branch", 11: "7, kill by data, branch behind the look", 12: "7, the scalar test behind the vector chain", 13: "12, the kill mask in front of the branch", 14: "13 + a wait state behind the compare", 15: "14, the collector's words from a register pair", 16: "15, a row in one read (the product's run)"}
out = (C.c_ulonglong * 4)()
ms = C.c_double(0)
L.chain_step_run(0, 200, 16, 4, 0, 0, out, C.byref(ms))  # warm-up
print("cycles a step of the chain's plain run, ONE wave alone on a CU (tools/microbench/chain_step.hip); `check`: what the run killed and the words it left\n"
      "for the collector, summed — equal checks, equal results.  thin 1: the bitmap thinned to five eighths before every batch (the seeds alive),\n"
      "so that a stale live mask shows: variants 7 and 13 read VCC by its number right behind the compare that wrote it by its alias.")
for thin in (0, 1):
    for v in (0, 1, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        for batch_n, group_n in ((16, 4),) if v not in (0, 16) else ((16, 4), (8, 4), (16, 9)):
            rc = L.chain_step_run(v, 4000, batch_n, group_n, 0, thin, out, C.byref(ms))
            t, c, b = out[0], out[1], out[2]
            print(f"thin {thin}  {v:2d} {names[v]:48s} batch {batch_n:2d} group {group_n}: {c} steps, {t / max(c, 1):7.1f} cycles a step, check {out[3]:016x}")
for pollers in (1, 7):
    for v in (0, 16):
        rc = L.chain_step_run(v, 4000, 16, 4, pollers, 0, out, C.byref(ms))
        print(f"{pollers} waves of the workgroup polling an LDS word beside it  {v:2d} {names[v]:48s}: {out[0] / max(out[1], 1):7.1f} cycles a step")
