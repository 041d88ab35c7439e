#!/usr/bin/env python3
"""Sum a PMC counter per kernel from a rocprofv3 rocpd database (counter collection run).

usage: tools/pmc_traffic.py <results.db> <COUNTER>   -> prints {kernel: {calls, total}}
"""
import json
import sqlite3
import sys


def main():
    db, counter = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    q = f"select {name_col}, counter_name, sum(value), count(*) from counters_collection where counter_name = ? group by {name_col}"
    out = {}
    for kname, cname, total, n in c.execute(q, (counter,)):
        out[kname] = {"calls": n, "total": total}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
