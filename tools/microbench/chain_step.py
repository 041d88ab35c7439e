#!/usr/bin/env python3
"""Runs tools/microbench/chain_step.hip (build it first, where hipcc is: see its head) and prints cycles per step."""
import ctypes as C
import os
import sys
here = os.path.dirname(os.path.abspath(__file__))
L = C.CDLL(os.path.join(here, "libchain_step.so"))
L.chain_step_run.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_ulonglong), C.POINTER(C.c_double)]
names = {0: "the product's loop", 1: "without the collector's word", 3: "without the kill", 4: "the look alone", 5: "two steps a trip", 7: "... and the look waited for alone", 8: "7 without the collector's word", 9: "7 without the attention test", 10: "7, the test without its branch", 11: "7, kill by data, branch behind the look", 12: "7, the scalar test behind the vector chain", 13: "12, the kill mask in front of the branch", 14: "13 + a wait state behind the compare", 15: "14, the collector's words from a register pair", 16: "15, a row in one read"}
out = (C.c_ulonglong * 4)()
ms = C.c_double(0)
L.chain_step_run(0, 200, 16, 4, 0, 0, out, C.byref(ms))  # warm-up
for thin in (0, 1):
  pollers = 0
  if True:
    for v in (0, 15, 16):
        for batch_n, group_n in ((16, 4), (16, 9)):
            rc = L.chain_step_run(v, 4000, batch_n, group_n, pollers, thin, out, C.byref(ms))
            t, c, b = out[0], out[1], out[2]
            print(f"thin {thin}  {names[v]:42s} batch {batch_n:2d} group {group_n}: rc {rc}, {c} steps, {t / max(c, 1):7.1f} ticks a step "
                  f"({t / max(b, 1):8.1f} a batch), kernel {ms.value:.2f} ms, check {out[3]:016x}")
