#!/usr/bin/env python3
"""K pools in ONE process on one GPU under a given number of HIP hardware queues — the `pools_on_one_gpu` leg of
bench.py on its own (K Python threads calling pm_tick; pm_tick_many from one thread and with a thread per engine),
printed as a table.  The runtime reads GPU_MAX_HW_QUEUES once, when it starts, so one process = one setting:

    for q in 4 8 16 32; do python tools/pools_probe.py --queues $q; done

profiles/r04_pools_hw_queues.json is this for 4 / 8 / 16 / 32 (taken through bench.py itself).  Next:
an engine that owns one stream instead of two, pm_tick_many in chunks of eight."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--queues", default="16", help="GPU_MAX_HW_QUEUES for this process ('default' = leave the runtime's 4)")
    ap.add_argument("--ks", default="2,4,8", help="pool counts for the Python-thread leg (pm_tick_many adds 16)")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    if args.queues != "default":
        os.environ["GPU_MAX_HW_QUEUES"] = args.queues    # before anything touches HIP
    if os.environ.get("PM_EXP_DEFINES"):  # a variant build of the library (extra -D flags), as tools/variant_bench.py
        from protocol_amd import build as B
        alt = os.path.join(os.path.dirname(B.LIB_PATH), "libpm_engine_exp.so")
        B.build(force=True, defines=[d for d in os.environ["PM_EXP_DEFINES"].split(",") if d], out=alt)
        B.LIB_PATH = alt
        B.needs_build = lambda: False
    import bench
    from protocol_amd import engine as E
    from protocol_amd import host
    out = bench.run_extra_pools(E, host, args.seed, ks=tuple(int(k) for k in args.ks.split(",")), steps=args.steps)
    if args.json:
        print(json.dumps(out, indent=1))
        return 0
    print(f"GPU_MAX_HW_QUEUES = {out.get('hip_hw_queues')}; one pool: {out['one_pool']['match_ms_p50']:.3f} ms per match")
    print("  K   python threads          pm_tick_many staged     pm_tick_many threads")
    tm = out.get("tick_many", {}).get("by_k", {})
    for K in sorted({*out["by_k"], *tm}, key=int):
        cell = lambda v: f"{v['x_one_pool']:5.2f}x p50 {v['match_ms_p50']:6.2f} ms" if v else " " * 22
        print(f"{int(K):3d}   {cell(out['by_k'].get(K))}   {cell(tm.get(K, {}).get('staged'))}   {cell(tm.get(K, {}).get('threads'))}")
    if "error" in out.get("tick_many", {}):
        print("pm_tick_many leg:", out["tick_many"]["error"])
    return 0


if __name__ == "__main__":
    sys.exit(main())
