#!/usr/bin/env python3
"""Join the batch log of tools/prune_probe.py with the proposer launches of the LAST match in the rocprofv3 trace of
that run: per batch, list length, seeds, walk or sweep, and what the launch cost.
usage  prune_join.py <probe.json> <results.db> [<probe2.json> <results2.db>]  (two runs: side by side, batch by batch)"""
import json
import sqlite3
import sys


def launches(dbp):
    db = sqlite3.connect(dbp)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kt = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol") or (t.startswith("rocpd_") and "kernel_symbol" in t)][0]
    names = dict(cur.execute(f"select id, kernel_name from {ks}"))
    seq = [(names[k], s, e) for k, s, e in cur.execute(f"select kernel_id,start,end from {kt} order by start")]
    idx = max(i for i, (n, _, _) in enumerate(seq) if "compat_kernel" in n or "compat_sliced_kernel" in n)
    last = seq[idx:]
    prop = [(e - s) / 1000 for n, s, e in last if "carve_propose" in n]
    build = sum((e - s) / 1000 for n, s, e in last if "cell_" in n)
    return prop, build


runs = []
for k in range(1, len(sys.argv) - 1, 2):
    probe = json.load(open(sys.argv[k]))
    prop, build = launches(sys.argv[k + 1])
    runs.append((probe, prop, build))
for probe, prop, build in runs:
    print(f"mode {probe['mode']}: carve {probe['carve_ms'][1]:.3f} ms, index build {build:.1f} us, proposer total {sum(prop):.1f} us "
          f"over {len(prop)} launches, {len(probe['batches'])} preparations")
nb = min(len(r[0]["batches"]) for r in runs)
print("batch  n_list  seeds " + "  ".join(f"m{r[0]['mode']}:grid  us" for r in runs))
for b in range(nb):
    nl, ns, _ = runs[0][0]["batches"][b]
    cols = []
    for probe, prop, _ in runs:
        cols.append(f"{probe['batches'][b][2]:7d} {prop[b] if b < len(prop) else float('nan'):7.1f}")
    print(f"{b:5d} {nl:7d} {ns:6d} " + "  ".join(cols))
