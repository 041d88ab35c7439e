#!/usr/bin/env python3
"""Debug helper: form groups on a large swarm with the pipelined (0), single-stream (3) and sequential (1)
carve variants and report where they differ."""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import engine as E, host
from protocol_amd.swarm import baseline_config, make_swarm

W = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sw = baseline_config(2, seed=1) if W == 100000 else make_swarm(2, 2000, W, zipf=True)
res = {}
for v in (3, 0, 0):
    eng = E.Engine(carve_variant=v)
    host.load_swarm(eng, sw)
    n = eng.form_groups()
    gow, groups, members = eng.get_groups()
    seen = np.zeros(sw.W, dtype=np.int32)
    np.add.at(seen, members, 1)
    st = eng.last_stats()
    print(f"variant {v}: {n} groups, {len(members)} members, max multiplicity {seen.max()}, dup workers {(seen > 1).sum()}, "
          f"steps {st['carve_steps']} fast {st['carve_fast_steps']} launches {st['carve_launches']}")
    key = [(int(g["config"]), tuple(members[int(g["member_begin"]):int(g["member_begin"]) + int(g["n_members"])].tolist())) for g in groups]
    if 3 in res and v == 0:
        ref = res[3]
        for i, (a, b) in enumerate(zip(ref, key)):
            if a != b:
                print("first difference at group", i, "ref", a[0], a[1][:6], "got", b[0], b[1][:6])
                # which configs are the groups around it
                print("ref configs around:", [r[0] for r in ref[max(0, i - 3):i + 3]], "got:", [r[0] for r in key[max(0, i - 3):i + 3]])
                break
        else:
            print("identical prefix; lengths", len(ref), len(key))
    res[v] = key
    eng.close()
