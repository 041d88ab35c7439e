#!/usr/bin/env python3
"""Dump the per-kernel summary (rocprofv3 --kernel-trace --stats, rocpd sqlite output) as CSV.

usage: tools/rocpd_summary.py <results.db> [out.csv]
"""
import csv
import sqlite3
import sys


def main():
    db = sys.argv[1]
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout
    c = sqlite3.connect(db)
    cur = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
    for name, calls, total, avg, pct in cur:
        w.writerow([name, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.4f}"])


if __name__ == "__main__":
    main()
