#!/usr/bin/env python3
"""Per-launch timeline of the last match in a rocprofv3 (rocpd sqlite) kernel trace: kernel, start offset,
duration and the idle gap in front of it.  usage: tick_timeline.py <results.db> [--all]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol") or (t.startswith("rocpd_") and "kernel_symbol" in t)][0]
    names = dict(cur.execute(f"select id, kernel_name from {ks}"))
    rows = list(cur.execute(f"select kernel_id,start,end from {kt} order by start"))
    seq = [(names[k], s, e) for k, s, e in rows]
    idx = max(i for i, (n, _, _) in enumerate(seq) if "compat_kernel" in n or "compat_sliced_kernel" in n)
    t0 = seq[idx][1]
    prev_end = None
    busy = 0
    brief = "--all" not in sys.argv
    for n, s, e in seq[idx:]:
        gap = (s - prev_end) / 1000 if prev_end else 0.0
        short = n.split("(")[0].replace("pm::", "").replace("_ZN2pm", "")[:34]
        if not brief or gap > 3.0 or "carve" not in n:
            print(f"{short:34s} start {(s - t0) / 1000:8.1f} us  dur {(e - s) / 1000:7.1f}  gap {gap:6.1f}")
        busy += e - s
        prev_end = e
    print(f"span {(prev_end - t0) / 1000:.1f} us, kernels+copies busy {busy / 1000:.1f} us")


if __name__ == "__main__":
    main()
