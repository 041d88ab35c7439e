"""Probe of the streaming carve (carve_variant 0) on one GPU: a ladder of swarms, each formed by the engine and by the
oracle (or checked against the batch pipeline where the oracle is too slow), with the carve's counters and time.

    python tools/stream_probe.py <case> [variant]      one case, in this process
    python tools/stream_probe.py all                   every case, each in its own bounded subprocess

Cases: cfg0 (1k x 256), small (2000 workers), mid (6000), cfg1 (100k x 10k), big (30k workers), cfg2 (1M x 100k).
"""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = ["cfg0", "small", "mid", "cfg1", "big", "cfg2"]


def swarm_of(case, seed=1):
    from protocol_amd.swarm import baseline_config, make_swarm
    if case == "cfg0":
        return baseline_config(0, seed=seed)
    if case == "small":
        return make_swarm(seed, 2000, 512)
    if case == "mid":
        return make_swarm(seed, 4000, 6000)
    if case == "cfg1":
        return baseline_config(1, seed=seed)
    if case == "big":
        return make_swarm(seed, 4000, 30000)
    if case == "cfg2":
        return baseline_config(2, seed=seed)
    raise SystemExit(f"unknown case {case}")


def groups_of(eng):
    from helpers import engine_groups
    return engine_groups(eng)


def run_case(case, variant):
    import numpy as np
    from protocol_amd import engine as E, host
    sw = swarm_of(case)
    eng = E.Engine(carve_variant=variant, group_id_seed=1)
    host.load_swarm(eng, sw)
    t0 = time.perf_counter()
    n = eng.form_groups()
    dt = (time.perf_counter() - t0) * 1e3
    st = eng.last_stats()
    info = eng.debug_carve_counters()
    got = groups_of(eng)
    # a second, warm run for the time
    eng.reset_groups()
    t0 = time.perf_counter()
    eng.form_groups()
    dt2 = (time.perf_counter() - t0) * 1e3
    st2 = eng.last_stats()
    out = {"case": case, "variant": variant, "W": int(sw.W), "groups": n, "ms_cold": round(dt, 3), "ms_warm": round(dt2, 3),
           "carve_ms": round(st2["ms_carve_kernel"], 4), "launches": st2["carve_launches"], "steps": st2["carve_steps"],
           "fast": st2["carve_fast_steps"], "host_resolved": st2["host_resolved_steps"], "info": info}
    # the checker: the oracle where it is quick, else the batch pipeline (itself pinned by the oracle digests)
    if sw.W <= 12000:
        from helpers import oracle_groups, oracle_state_for
        ost = oracle_state_for(sw, reference_shaped=(sw.W <= 2048), group_id_seed=1)
        ost.try_form_new_groups()
        want = oracle_groups(ost)
        out["checked_against"] = "oracle"
    else:
        e2 = E.Engine(carve_variant=3, group_id_seed=1)
        host.load_swarm(e2, sw)
        e2.form_groups()
        want = groups_of(e2)
        out["batch_carve_ms"] = round(e2.last_stats()["ms_carve_kernel"], 4)
        out["batch_launches"] = e2.last_stats()["carve_launches"]
        out["checked_against"] = "batch pipeline"
        e2.close()
    out["match"] = got == want
    if not out["match"]:
        k = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        out["first_diff"] = {"index": k, "got": got[k] if k < len(got) else None, "want": want[k] if k < len(want) else None,
                             "n_got": len(got), "n_want": len(want)}
    eng.close()
    print(json.dumps(out), flush=True)
    return out["match"]


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    if sys.argv[1] == "all":
        variant = sys.argv[2] if len(sys.argv) > 2 else "0"
        ok = True
        for case in CASES:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), case, variant], timeout=240, capture_output=True, text=True)
                print(r.stdout.strip() or f'{{"case": "{case}", "rc": {r.returncode}}}', flush=True)
                if r.returncode != 0:
                    ok = False
                    print(r.stderr[-2000:], flush=True)
            except subprocess.TimeoutExpired:
                ok = False
                print(f'{{"case": "{case}", "timeout": true}}', flush=True)
                break  # (a hung GPU: do not pile more work on it)
        sys.exit(0 if ok else 1)
    ok = run_case(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
