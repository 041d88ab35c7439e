"""CPU-side checks of the product's host layer (no GPU): the C-ABI library loads and exports every
symbol include/pm_engine.h declares, its parser / model rule / config order agree with the reference's
known-answer vectors and with the oracle, and the engine refuses to run without an MI355X."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest
import torch

from oracle import oracle_ffi as orc
from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.swarm import GPU_MODELS, MIXED_CONFIGS, UNIFORM_CONFIGS, make_swarm, mix64

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KATS = json.load(open(os.path.join(ROOT, "tests", "golden", "node_rs_kats.json")))

ALT_FIELDS = [("count", E.G_COUNT), ("memory_mb", E.G_MEM), ("memory_mb_min", E.G_MEM_MIN),
              ("memory_mb_max", E.G_MEM_MAX), ("total_memory_min", E.G_TOT_MIN), ("total_memory_max", E.G_TOT_MAX)]


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pm_engine.h")).read()
    declared = set(re.findall(r"\b(pm_[a-z_]+)\s*\(", hdr))
    L = E.lib()
    for name in declared:
        assert hasattr(L, name), f"{name} declared in pm_engine.h but not exported"
    assert declared == set(E.EXPORTS)
    assert L.pm_abi_version() == 3


def test_library_exports_every_debug_hook():
    """include/pm_engine_debug.h: the test hooks and counters tests/, tools/ and bench.py call.  Not part of the drop-in
    boundary (the Rust shim binds none of them), but exported by the shipped library, with the declared arity."""
    hdr = open(os.path.join(ROOT, "include", "pm_engine_debug.h")).read()
    protos = _c_prototypes(hdr)
    assert set(protos) == {"pm_debug_carve_prof", "pm_debug_stream_trace", "pm_debug_mem_lists_above", "pm_debug_prune_mode",
                           "pm_debug_hbm_triad", "pm_debug_stream_abort_after", "pm_debug_merge_streamed", "pm_debug_delta_pushes",
                           "pm_debug_row_networks", "pm_debug_row_records", "pm_debug_chain_batches", "pm_debug_park_records"}
    L = E.lib()
    for name in protos:
        assert hasattr(L, name), f"{name} declared in pm_engine_debug.h but not exported"
    shim = open(os.path.join(ROOT, "rust", "gpu_match_plugin.rs")).read()
    assert "pm_debug_" not in shim
    public = open(os.path.join(ROOT, "include", "pm_engine.h")).read()
    assert not re.findall(r"\b(pm_debug_[a-z_]+)\s*\(", public)
    # every pm_debug_* the Python side calls is declared there
    used = set()
    for sub in ("protocol_amd", "tests", "tools"):
        for fn in os.listdir(os.path.join(ROOT, sub)):
            if fn.endswith(".py"):
                used |= set(re.findall(r"\.(pm_debug_[a-z_]+)\b", open(os.path.join(ROOT, sub, fn)).read()))
    # (batch_log: -DPM_BATCH_LOG builds only; row_bench: -DPM_ROW_BENCH builds only, tools/row_bench.py)
    assert used - {"pm_debug_batch_log", "pm_debug_row_bench"} <= set(protos), used - set(protos)


def _c_prototypes(text):
    """{name: number of parameters} of the `int32_t|void|... pm_*(...)` prototypes in a C header"""
    out = {}
    for m in re.finditer(r"\b(pm_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", re.sub(r"/\*.*?\*/", "", text, flags=re.S)):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def test_rust_shim_binds_the_header_as_declared():
    """rust/gpu_match_plugin.rs cannot be compiled in this image (no cargo): at least every function of its
    extern "C" block must exist in include/pm_engine.h with the same number of parameters, the repr(C) structs it
    passes by pointer must have the fields of the C structs in the same order, and nothing it calls may be missing
    from the block."""
    hdr = open(os.path.join(ROOT, "include", "pm_engine.h")).read()
    rs = open(os.path.join(ROOT, "rust", "gpu_match_plugin.rs")).read()
    protos = _c_prototypes(hdr)
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    bound = {}
    for m in re.finditer(r"fn (pm_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*[\w:*<> ]+)?;", block, flags=re.S):
        args = m.group(2).strip()
        bound[m.group(1)] = 0 if not args else len([a for a in args.split(",") if a.strip()])
    assert len(bound) >= 30
    for name, n in bound.items():
        assert name in protos, f"{name} is bound by the Rust shim but not declared in pm_engine.h"
        assert protos[name] == n, f"{name}: {n} parameters in the Rust shim, {protos[name]} in pm_engine.h"
    called = set(re.findall(r"\b(pm_[a-z_0-9]+)\s*\(", rs[rs.index("\n}\n", rs.index('extern "C" {')):]))
    called -= {"pm_engine_config", "pm_worker_soa", "pm_task_soa", "pm_stats", "pm_group_event", "pm_assignment", "pm_group"}
    assert called <= set(bound), f"called but not bound: {sorted(called - set(bound))}"
    # struct layouts: field names in order
    def c_fields(name):
        body = re.findall(r"typedef struct(?: \w+)?\s*\{([^{}]*)\}\s*" + name + r"\s*;", hdr, flags=re.S)[0]
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(",")
            first = names[0].split()[-1]
            fields.append(first.lstrip("*").split("[")[0])
            fields += [n.strip().lstrip("*").split("[")[0] for n in names[1:]]
        return fields

    def rs_fields(name):
        body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        return re.findall(r"pub (\w+)\s*:", body)

    for name in ("pm_engine_config", "pm_worker_soa", "pm_task_soa", "pm_config_row", "pm_gpu_alt_row", "pm_assignment",
                 "pm_group_event", "pm_stats", "pm_group", "pm_dist_xfer", "pm_group_vars"):
        assert rs_fields(name) == c_fields(name), name
    # the read surface the API routes call (node_groups/mod.rs:324-434, :1002-1065), with the reference's names
    for method in ("get_all_groups", "get_group_by_id", "get_all_node_group_mappings", "get_node_group", "get_node_groups_batch",
                   "get_idx_in_group", "get_available_configurations", "get_all_configuration_templates", "dissolve_group"):
        assert re.search(r"pub(?:\(crate\))? (?:async )?fn " + method + r"\(&self", rs), method
        assert re.search(r"\b" + method + r"\(", open(os.path.join(ROOT, "protocol_amd", "plugin", "gpu_match_plugin.hpp")).read()), method


_C_BASE = {"uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "float": "f32",
           "double": "f64", "char": "c_char", "void": "c_void", "pm_engine": "c_void", "size_t": "usize",
           "unsigned long long": "u64"}


def _c_decl_to_rust(decl: str) -> str:
    """One C declarator ('const uint32_t* flags', 'pm_engine* const* engines', 'uint32_t') as the Rust FFI type that has
    its layout: '*const u32', '*const *mut c_void', 'u32' (a pointer's target is const if the qualifier stands in
    front of that '*'; the opaque pm_engine is c_void on the Rust side)."""
    toks = re.findall(r"[A-Za-z_]\w*|\*", decl)
    is_type = lambda t: t in ("const", "unsigned", "long", "int", "struct") or t in _C_BASE or t.startswith("pm_") or t.endswith("_t")
    if toks and toks[-1] != "*" and not is_type(toks[-1]):
        toks = toks[:-1]  # the parameter / field name
    i, target_const, base = 0, False, []
    while i < len(toks) and toks[i] != "*":
        if toks[i] == "const":
            target_const = True
        else:
            base.append(toks[i])
        i += 1
    ty = _C_BASE.get(" ".join(base), " ".join(base))
    while i < len(toks):
        assert toks[i] == "*", decl
        ty = ("*const " if target_const else "*mut ") + ty
        target_const = False
        i += 1
        if i < len(toks) and toks[i] == "const":
            target_const = True
            i += 1
    return ty


def test_rust_shim_types_are_the_headers_types():
    """... and with the same TYPES: every parameter and return type of the extern block and every field of the repr(C)
    structs, translated from the C declaration (uint32_t -> u32, const T* -> *const T, pm_engine* -> *mut c_void,
    size_t -> usize), is what the shim wrote.  A width or a const / mut slip is an ABI bug no test over the C ABI sees."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pm_engine.h")).read(), flags=re.S)
    rs = open(os.path.join(ROOT, "rust", "gpu_match_plugin.rs")).read()
    norm = lambda t: re.sub(r"\s+", " ", t.strip())
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(pm_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", hdr):
        ret, args = m.group(1).strip().split("\n")[-1].strip(), m.group(3).strip()
        params = [] if args in ("", "void") else [_c_decl_to_rust(a) for a in args.split(",")]
        protos[m.group(2)] = (None if ret == "void" else _c_decl_to_rust(ret + " x"), params)
    block = rs[rs.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    n = 0
    for m in re.finditer(r"fn (pm_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*([\w:*<> ]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(3)
        params = [norm(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        assert ((norm(ret) if ret else None), params) == protos[name], (name, ret, params, protos[name])
        n += 1
    assert n >= 30
    # the structs: (field, type) in order
    def c_struct(name):
        body = re.findall(r"typedef struct(?: \w+)?\s*\{([^{}]*)\}\s*" + name + r"\s*;", hdr, flags=re.S)[0]
        out = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(",")          # 'float a, b' / 'const uint32_t *x, *y'
            head = parts[0]
            base = re.match(r"\s*((?:const\s+)?[A-Za-z_]\w*(?:\s+long\s+long)?)", head).group(1)
            out.append(_c_decl_to_rust(head))
            out += [_c_decl_to_rust(base + " " + more) for more in parts[1:]]
        return out

    def rs_struct(name):
        body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", rs, flags=re.S).group(1)
        return [norm(t) for t in re.findall(r"pub \w+\s*:\s*([^,\n]+?)\s*(?:,|\n|$)", body)]

    for name in ("pm_engine_config", "pm_worker_soa", "pm_task_soa", "pm_config_row", "pm_gpu_alt_row", "pm_assignment",
                 "pm_group_event", "pm_stats", "pm_dist_xfer", "pm_group_vars"):
        assert rs_struct(name) == c_struct(name), (name, rs_struct(name), c_struct(name))


@pytest.mark.skipif(torch.cuda.is_available(), reason="this check is for boxes without a GPU")
def test_engine_fails_loudly_without_gpu():
    with pytest.raises(E.EngineError) as ei:
        E.Engine()
    assert ei.value.code == E.PM_ENODEV and "no CPU fallback" in str(ei.value)


def parsed_to_dict(cfg, alts, names):
    gpus = []
    for a, nm in zip(alts, names):
        d = {k: int(a[k]) for k, bit in ALT_FIELDS if int(a["flags"]) & bit}
        if nm is not None:
            d["model"] = nm
        gpus.append(d)
    f = int(cfg["flags"])
    return {"gpu": gpus,
            "ram_mb": int(cfg["ram_mb"]) if f & E.R_RAM else None,
            "storage_gb": int(cfg["storage_gb"]) if f & E.R_STORAGE else None,
            "cpu": ({"cores": int(cfg["cpu_cores"])} if f & E.R_CPU_CORES else {}) if f & E.R_CPU else None}


@pytest.mark.parametrize("kat", KATS["parser"], ids=lambda k: k["name"])
def test_product_parser_kat(kat):
    assert parsed_to_dict(*host.parse_requirements(kat["req"])) == kat["expect"]


@pytest.mark.parametrize("kat", KATS["parser_errors"], ids=lambda k: k["name"])
def test_product_parser_error_kat(kat):
    with pytest.raises(E.EngineError) as ei:
        host.parse_requirements(kat["req"])
    assert ei.value.code == E.PM_EPARSE


def test_product_parser_panic_and_grammar_match_oracle():
    cases = ["gpu:memory_mb_max=5;gpu:memory_mb_min=x", "gpu:total_memory_min=5;gpu:total_memory_max=x",
             "gpu:memory_mb_min=x", "ram_mb=+5", "ram_mb=-5", "ram_mb=4294967295", "ram_mb=4294967296",
             "ram_mb=1 2", "gpu:model=H100;gpu:count=2", ";;;", "a", "=", "gpu:count=1;gpu:count=2;gpu:model=x",
             "gpu:model=a;gpu:model=b", "cpu:cores=4;cpu:cores=8", " \t gpu:count\t=\t7 \n;"]
    code_map = {0: E.PM_OK, 1: E.PM_EPARSE, 2: E.PM_EPANIC}
    for s in cases + [c[3] for c in MIXED_CONFIGS + UNIFORM_CONFIGS if c[3]]:
        ocode, orow, _ = orc.parse_requirements(s)
        try:
            got = parsed_to_dict(*host.parse_requirements(s))
            pcode = E.PM_OK
        except E.EngineError as ex:
            pcode, got = ex.code, None
        assert pcode == code_map[ocode], s
        if ocode == 0:
            from test_oracle_kats import req_to_dict
            assert got == req_to_dict(orow), s


def test_model_table_matches_oracle_rule():
    req_models = sorted({a["model"] for k in KATS["parser"] for a in k["expect"]["gpu"] if "model" in a} |
                        {"a100,h100,h200", "nvidia,rtx", "rtx4090,rtx_3090", "a6000,l40s", "v100", "mi300x", "", "zzz, ",
                         "RTX_4090", "nvidia_a100_80gb_pcie"})
    spec_models = GPU_MODELS + sorted({k["specs"][1] for k in KATS["meets"] if k["specs"][1]}) + ["", "_", "h 100"]
    bits = host.build_model_table(req_models, spec_models)
    words = (len(spec_models) + 31) // 32
    for r, rm in enumerate(req_models):
        for c, sm in enumerate(spec_models):
            got = (int(bits[r * words + (c >> 5)]) >> (c & 31)) & 1
            assert bool(got) == orc.model_matches(sm, rm) == host.model_matches(sm, rm), (sm, rm)


def test_config_order_matches_oracle():
    rng = np.random.default_rng(3)
    for trial in range(50):
        n = int(rng.integers(1, 24))
        sel = rng.permutation(len(MIXED_CONFIGS))[:n]
        cfgs = [MIXED_CONFIGS[i] for i in sel]
        cfg_rows, _, _ = host.pack_configs(cfgs)
        ocfgs = np.concatenate([orc.make_config(*c) for c in cfgs])
        enabled = int(rng.integers(0, 1 << n))
        code, tmpl = orc.sort_configs(ocfgs)
        assert code == 0
        en = np.array([(enabled >> i) & 1 for i in range(n)], dtype=np.uint8)
        out = np.zeros(n, dtype=np.uint32)
        m = orc.lib().orc_available_configs(ocfgs.ctypes.data, tmpl.ctypes.data, n, en.ctypes.data, out.ctypes.data)
        assert host.config_order(cfg_rows, enabled) == out[:m].tolist()


def test_swarm_generator_is_deterministic_and_in_spec():
    a, b = make_swarm(7, 5000, 2000), make_swarm(7, 5000, 2000)
    for k in ("address", "status", "gpu_count", "lat", "lon", "created_at", "topo", "task_uid"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k
    assert not np.array_equal(a.address, make_swarm(8, 5000, 2000).address)
    assert len(np.unique(a.address)) == a.W
    assert 0.95 < (a.status == 2).mean() < 0.99 and 0.97 < a.has_p2p.mean() <= 1.0
    assert 0.85 < a.has_loc.mean() < 0.95
    assert (a.lat >= 25).all() and (a.lat <= 60).all() and (a.lon >= -125).all() and (a.lon <= 40).all()
    assert (a.gpu_count.astype(np.uint64) * a.gpu_mem_mb.astype(np.uint64) < 2 ** 32).all()
    assert (np.diff(a.created_at) <= 0).all()                       # get_all_tasks order
    assert (a.price == 0).all()
    # splitmix64 here == the oracle's C implementation (and therefore the engine's)
    assert np.array_equal(mix64(np.arange(5, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(42)),
                          orc.splitmix64_stream(42, 5))


def test_swarm_masks_agree_with_oracle_topology_predicate():
    sw = make_swarm(2, 3000, 64)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    masks = sw.task_masks()
    for c in range(len(cfgs)):
        name = bytes(cfgs[c]["name"])
        want = np.array([orc.lib().orc_task_applicable(tasks[t:t + 1].ctypes.data, name) for t in range(sw.T)])
        got = ((masks >> np.uint64(c)) & np.uint64(1)).astype(np.int64)
        assert np.array_equal(want, got), c


def test_committed_bench_line_keeps_the_driver_contract():
    """The last bench line committed under profiles/ carries every key the driver and the judge read."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench.json")))
    assert lines, "no bench line committed under profiles/"
    d = json.load(open(lines[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    # (`bound` is one of the contract's two roofs; "latency-chain" — what the round-2 review asked the carve sequence to be
    # labelled — is in `binds`; achieved / peak / frac are the contract's HBM accounting)
    assert r["bound"] in ("hbm", "mfma") and r.get("binds") == "latency-chain" and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # whole-job throughput = pairs per step / time per step
    assert abs(d["value"] - d["config"]["tasks"] * d["config"]["workers_total"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_parser_differential_fuzz_product_vs_oracle():
    """Two independent implementations of ComputeRequirements::from_str (C++ product, C oracle) on a few thousand
    strings assembled from the grammar's tokens and from junk: same outcome class, same parsed requirements."""
    import random
    from test_oracle_kats import req_to_dict
    rng = random.Random(20250801)
    keys = ["gpu:count", "gpu:model", "gpu:memory_mb", "gpu:memory_mb_min", "gpu:memory_mb_max",
            "gpu:total_memory_min", "gpu:total_memory_max", "cpu:cores", "ram_mb", "storage_gb", "gpu", "cpu", "gpu:", ":",
            "GPU:count", "gpu:Count", "ram", "storage"]
    vals = ["0", "1", "8", "80000", "4294967295", "4294967296", "-1", "+2", "1.5", "", " 4 ", "x", "H100", "a100,h100",
            "RTX_4090", "rtx 3090", "a=b", "1e3", "0x10", "００"]
    seps = [";", ";", ";", ";;", " ; ", ",", "\n", ""]
    eqs = ["=", "=", "=", " = ", "==", ":", ""]
    code_map = {0: E.PM_OK, 1: E.PM_EPARSE, 2: E.PM_EPANIC}
    n_ok = n_err = n_panic = 0
    for _ in range(4000):
        parts = [rng.choice(keys) + rng.choice(eqs) + rng.choice(vals) for _ in range(rng.randint(0, 6))]
        s = "".join(p + rng.choice(seps) for p in parts)
        if rng.random() < 0.1:
            s = s.replace("gpu", "gpu ", 1)
        ocode, orow, _ = orc.parse_requirements(s)
        if ocode == 3:          # more alternatives than the oracle's fixed row holds
            continue
        try:
            got = parsed_to_dict(*host.parse_requirements(s))
            pcode = E.PM_OK
        except E.EngineError as ex:
            pcode, got = ex.code, None
        assert pcode == code_map[ocode], repr(s)
        if ocode == 0:
            assert got == req_to_dict(orow), repr(s)
            n_ok += 1
        elif ocode == 1:
            n_err += 1
        else:
            n_panic += 1
    assert n_ok > 200 and n_err > 200      # the generator reaches both sides of the grammar


def test_member_list_of_a_group_record(tmp_path):
    """protocol_amd/csrc/pm_members.h (the member list of the engine's host-side group records: inline up to eight members,
    heap beyond) against std::vector under the operations absorb_groups / compact_groups / run_merge perform —
    tests/cpp/members_test.cpp, built here with g++ (address + UB sanitizers when they link)."""
    import shutil
    import subprocess
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "tests", "cpp", "members_test.cpp")
    exe = str(tmp_path / "members_test")
    base = [gxx, "-std=c++17", "-O1", "-g", "-Wall", "-Werror", "-I", os.path.join(ROOT, "protocol_amd", "csrc"), src, "-o", exe]
    if subprocess.run(base + ["-fsanitize=address,undefined"], capture_output=True).returncode != 0:
        subprocess.check_call(base)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "members_test ok" in out.stdout, out.stdout + out.stderr


def test_product_library_is_the_build_the_gpu_suite_ran_on():
    """profiles/r*_verified_binary.txt (the newest) holds the sha256 of the six sections of libpm_engine.so that carry
    code and data — the gfx950 code object's .text / .rodata / .note, the host's .text / .rodata / .data — as they were
    in the build the last full `pytest -m gpu` run was made on.  They are deterministic from build to build (unlike the
    library as a whole), so this says: what is in the tree compiles to what was verified.  A change that is meant to leave
    the product alone (moving code between files, experiments behind #ifdef, comments) passes as it is; a change to the
    product fails here until the GPU suite has been run on it and the file refreshed (python tools/section_hashes.py)."""
    import glob
    import importlib.util
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_verified_binary.txt")))
    assert files, "no verified-binary record under profiles/"
    want = dict(re.findall(r"^\s*((?:gfx950|host) \.\w+)\s+([0-9a-f]{16})\b", open(files[-1]).read(), flags=re.M))
    assert len(want) == 6, want
    spec = importlib.util.spec_from_file_location("section_hashes", os.path.join(ROOT, "tools", "section_hashes.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    if not os.path.exists(os.path.join(sh.LLVM, "llvm-objcopy")):
        pytest.skip("no llvm-objcopy / clang-offload-bundler here")
    from protocol_amd import build as B
    # (the hashes are those of one toolchain: the record names the compiler it was made with — another one compiles the
    # same sources to other bytes, and that is no finding)
    text = open(files[-1]).read()
    rec = re.search(r"^\s*toolchain\s+(.+)$", text, flags=re.M)
    here = sh.toolchain()
    if rec and rec.group(1).strip() != here:
        pytest.skip(f"record made with {rec.group(1).strip()!r}, this is {here!r}")
    got = sh.hashes(B.build())
    if got != want and os.environ.get("PM_CHECK_VERIFIED_BINARY") != "1":
        # a product change between two GPU runs: reported, not failed (a CPU-only checkout cannot refresh the record);
        # PM_CHECK_VERIFIED_BINARY=1 makes it a failure — what the end of a round is checked with
        pytest.skip("the product library differs from the build the GPU suite ran on (" + os.path.basename(files[-1]) +
                    "): run the GPU suite on it, then refresh that file (python tools/section_hashes.py)")
    assert got == want, ("the product library differs from the build the GPU suite ran on (" + os.path.basename(files[-1]) +
                         "): run the GPU suite on it, then refresh that file", got, want)


def test_tuning_flags_compile_together(tmp_path):
    """the tuning constants of the streaming carve that a variant build may set with -D (tools/build_variants.py) still
    compile together for gfx950 (there is no experiment behind an #ifdef left in the product sources: what round 4 left
    was measured in round 5 and removed — profiles/r05_losing_*)"""
    from protocol_amd import build as B
    flags = ["PM_STREAM_PROP_WAVES_N=8", "STREAM_PRE_EARLY=256u", "STREAM_PRE_NEAR=512u", "STREAM_LA_ALL=1024u"]
    srcs = "".join(open(os.path.join(B.CSRC, f)).read() for f in B.SOURCES + B.HEADERS)
    assert "PM_EXP_" not in srcs.replace("PM_EXP_DEFINES", "")
    out = B.build(force=True, defines=flags, out=str(tmp_path / "libpm_engine_exp.so"))
    assert os.path.getsize(out) > 500_000
