"""Pin the CPU oracle against the reference's own known-answer tests.

Vectors: tests/golden/node_rs_kats.json, transcribed from
crates/shared/src/models/node.rs:659-1241 (12 parser tests, 21 `meets` tests) and
crates/orchestrator/src/plugins/newest_task/mod.rs:29-55.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle_ffi as orc

KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "node_rs_kats.json")))

GPU_FIELDS = [("count", orc.G_COUNT), ("memory_mb", orc.G_MEM), ("memory_mb_min", orc.G_MEM_MIN),
              ("memory_mb_max", orc.G_MEM_MAX), ("total_memory_min", orc.G_TOT_MIN),
              ("total_memory_max", orc.G_TOT_MAX)]


def req_to_dict(row):
    gpus = []
    for i in range(int(row["n_gpu"])):
        g = row["gpu"][i]
        d = {}
        for name, bit in GPU_FIELDS:
            if int(g["flags"]) & bit:
                d[name] = int(g[name])
        if int(g["flags"]) & orc.G_MODEL:
            d["model"] = g["model"].decode()
        gpus.append(d)
    f = int(row["flags"])
    return {
        "gpu": gpus,
        "ram_mb": int(row["ram_mb"]) if f & orc.R_RAM else None,
        "storage_gb": int(row["storage_gb"]) if f & orc.R_STORAGE else None,
        "cpu": ({"cores": int(row["cpu_cores"])} if f & orc.R_CPU_CORES else {}) if f & orc.R_CPU else None,
    }


@pytest.mark.parametrize("kat", KATS["parser"], ids=lambda k: k["name"])
def test_parser_kat(kat):
    code, row, err = orc.parse_requirements(kat["req"])
    assert code == 0, err
    assert req_to_dict(row) == kat["expect"]


@pytest.mark.parametrize("kat", KATS["parser_errors"], ids=lambda k: k["name"])
def test_parser_error_kat(kat):
    code, row, err = orc.parse_requirements(kat["req"])
    assert code == 1, (code, err)


@pytest.mark.parametrize("kat", KATS["meets"], ids=lambda k: k["name"])
def test_meets_kat(kat):
    specs = orc.make_specs(*kat["specs"])
    code, req, err = orc.parse_requirements(kat["req"])
    assert code == 0, err
    assert orc.meets(specs, req) is kat["expect"]


def test_newest_task_kat():
    for kat in KATS["newest_task"]:
        tasks = np.concatenate([orc.make_task(c) for c in kat["created_at"]])
        assert orc.newest_task(tasks) == kat["expect_index"]
    assert orc.newest_task(np.zeros(0, dtype=orc.task_dt)) == -1
    # Iterator::max_by_key returns the LAST maximum (Appendix A)
    tasks = np.concatenate([orc.make_task(c) for c in (5, 9, 9, 3)])
    assert orc.newest_task(tasks) == 2


def test_parser_panics_and_number_grammar():
    # node.rs:251 — `.unwrap()` on a non-numeric min when max is already set panics in the reference
    assert orc.parse_requirements("gpu:memory_mb_max=5;gpu:memory_mb_min=x")[0] == 2
    assert orc.parse_requirements("gpu:total_memory_min=5;gpu:total_memory_max=x")[0] == 2
    # without the sibling bound it is a plain Err
    assert orc.parse_requirements("gpu:memory_mb_min=x")[0] == 1
    # Rust u32::from_str: leading '+' ok, '-' / inner space / overflow are errors
    assert orc.parse_requirements("ram_mb=+5")[0] == 0
    assert orc.parse_requirements("ram_mb=-5")[0] == 1
    assert orc.parse_requirements("ram_mb=4294967295")[0] == 0
    assert orc.parse_requirements("ram_mb=4294967296")[0] == 1
    assert orc.parse_requirements("ram_mb=1 2")[0] == 1
    # a later gpu:count only starts a new alternative if the current one already has a count (:205)
    code, row, _ = orc.parse_requirements("gpu:model=H100;gpu:count=2")
    assert code == 0 and req_to_dict(row)["gpu"] == [{"count": 2, "model": "H100"}]


def test_model_rule_details():
    # node.rs:463-484
    assert orc.model_matches("NVIDIA H100 80GB HBM3", "h100")
    assert orc.model_matches("h100", "NVIDIA H100 80GB HBM3")           # req contains spec
    assert orc.model_matches("RTX 4090", "rtx4090")                      # underscore-stripped
    assert orc.model_matches("rtx4090", "RTX_4090")
    assert orc.model_matches("anything", "")                            # contains("") is true
    assert orc.model_matches("anything", "zzz, ")                       # empty comma part matches
    assert not orc.model_matches("AMD Radeon RX 7900", "nvidia,rtx")


def test_meets_wrapping_total_memory():
    # node.rs:509: u32 multiply (wraps in release builds)
    specs = orc.make_specs(65536, "x", 65536)   # 2^32 wraps to 0
    code, req, _ = orc.parse_requirements("gpu:total_memory_min=1")
    assert not orc.meets(specs, req)
    code, req, _ = orc.parse_requirements("gpu:total_memory_max=0")
    assert orc.meets(specs, req)


def test_meets_count_none_semantics():
    # node.rs:447-461: spec count None passes only when the required count is 0
    specs = orc.make_specs(None, "A100", 40000)
    assert orc.meets(specs, orc.parse_requirements("gpu:count=0")[1])
    assert not orc.meets(specs, orc.parse_requirements("gpu:count=1")[1])


def test_haversine_coarse():
    h = KATS["haversine_coarse"]
    d = orc.calculate_distance(*h["montreal"], *h["dallas"])
    assert 2400.0 < d < 2500.0          # great-circle Montreal-Dallas is about 2430 km
    assert orc.calculate_distance(*h["montreal"], *h["montreal"]) == 0.0


def test_config_sort_rules():
    # mod.rs:150-164: min_group_size desc, then with-requirements first, stable
    cfgs = np.concatenate([
        orc.make_config("a", 1, 1, None),
        orc.make_config("b", 2, 4, None),
        orc.make_config("c", 2, 2, "gpu:count=8"),
        orc.make_config("d", 1, 8, "gpu:count=1"),
        orc.make_config("e", 2, 3, "ram_mb=1"),
    ])
    code, order = orc.sort_configs(cfgs)
    assert code == 0
    assert [cfgs[i]["name"].decode() for i in order] == ["c", "e", "b", "d", "a"]
    dup = np.concatenate([orc.make_config("a", 1, 1, None), orc.make_config("a", 1, 2, None)])
    assert orc.sort_configs(dup)[0] == 2
    bad = orc.make_config("x", 3, 2, None)
    assert orc.sort_configs(bad)[0] == 2


def test_fixture_is_the_references_test_module_structurally():
    """tests/golden/check_kats_against_reference.py parses the reference's own test module — every create_compute_specs call
    and ComputeSpecs literal, every requirement string through the `let` that names it, every assert!(..meets..) with its
    polarity, every from_str(..).is_err() — and compares the triples with the fixture (container-only: needs /root/reference)."""
    import importlib.util
    import os
    if not os.path.exists("/root/reference/crates/shared/src/models/node.rs"):
        pytest.skip("the reference is not present on this box")
    path = os.path.join(os.path.dirname(__file__), "golden", "check_kats_against_reference.py")
    spec = importlib.util.spec_from_file_location("check_kats", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main() == 0
    vectors = mod.reference_vectors(open(mod.REF).read())
    assert sum(1 for vs in vectors.values() for v in vs if v[0] == "meets") == 29
    assert sum(1 for vs in vectors.values() for v in vs if v[0] == "meets" and not v[3]) >= 9     # the negative assertions are seen as such
