#!/usr/bin/env python3
"""Build variant libraries of the engine (extra -D flags) where there is no GPU, so that a GPU call spends its minutes on
running them:  python tools/build_variants.py name=FLAG[,FLAG...] [name=...]  ->  protocol_amd/variants/libpm_engine_<name>.so
(git-ignored like every .so; they travel to the GPU box).  Used as PM_EXP_LIB=<path> by tests/conftest.py,
tools/variant_bench.py, tools/stream_prof.py."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from protocol_amd import build as B

OUT = os.path.join(os.path.dirname(B.LIB_PATH), "variants")


def one(spec: str) -> str:
    name, _, flags = spec.partition("=")
    out = os.path.join(OUT, f"libpm_engine_{name}.so")
    B.build(force=True, defines=[f for f in flags.split(",") if f], out=out)
    return out


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    with ThreadPoolExecutor(4) as ex:
        for p in ex.map(one, sys.argv[1:]):
            print(p)
