"""Transcription check for tests/golden/node_rs_kats.json and group_vars_kats.json — STRUCTURAL since round 6.

Run in the build container (where /root/reference exists).  The reference's own test module
(crates/shared/src/models/node.rs, `mod tests`) is parsed, test function by test function: every `create_compute_specs(...)`
call and every `ComputeSpecs { .. }` literal becomes the fixture's 6-tuple [gpu_count, gpu_model, gpu_mem, cpu_cores, ram,
storage], every `ComputeRequirements::from_str(..)` its requirement string (through the `let` that names it), and every
`assert!(specs.meets(&requirements))` / `assert!(!specs.meets(..))` a (specs, requirement, expected) triple with its
POLARITY; `assert!(ComputeRequirements::from_str(..).is_err())` the parser's error vectors.  The fixture must hold exactly
these triples, per test function, in the reference's order.  (Until round 5 this only looked for the strings.)

The GPU box has no /root/reference: a container-only integrity check, not a run-time dependency.
"""
import json
import os
import re
import sys

REF = "/root/reference/crates/shared/src/models/node.rs"
HERE = os.path.dirname(os.path.abspath(__file__))
_STR = r"\"((?:[^\"\\]|\\.)*)\""


def _value(text):
    """Some(4) / Some("A100") / Some("A100".to_string()) / None -> python value"""
    text = text.strip().rstrip(",").strip()
    if text == "None":
        return None
    inner = re.fullmatch(r"Some\((.*)\)", text, re.S).group(1).strip()
    m = re.match(_STR, inner)
    return m.group(1) if m else int(inner.replace("_", ""))


def _field(body, name):
    m = re.search(r"\b" + name + r":\s*(None|Some\((?:[^()]|\([^()]*\))*\))", body)
    return _value(m.group(1)) if m else None


def _struct_literal(text):
    """ComputeSpecs { gpu: Some(GpuSpecs { .. }), cpu: Some(CpuSpecs { .. }), ram_mb: .., storage_gb: .., ..Default }"""
    gpu = re.search(r"gpu:\s*Some\(GpuSpecs\s*\{(.*?)\}\)", text, re.S)
    cpu = re.search(r"cpu:\s*Some\(CpuSpecs\s*\{(.*?)\}\)", text, re.S)
    outer = text
    for m in (gpu, cpu):
        if m:
            outer = outer.replace(m.group(0), "")
    g = gpu.group(1) if gpu else ""
    return [_field(g, "count"), _field(g, "model"), _field(g, "memory_mb"), _field(cpu.group(1), "cores") if cpu else None,
            _field(outer, "ram_mb"), _field(outer, "storage_gb")]


def reference_vectors(src):
    """-> {test fn: [("meets", specs, req, expect) | ("parse", req, "err" | "ok")] in source order}"""
    tests = src[src.index("mod tests {"):]
    fns = [(m.group(1), m.start()) for m in re.finditer(r"#\[test\]\s*fn (\w+)\(\)\s*\{", tests)]
    out = {}
    for i, (name, at) in enumerate(fns):
        body = tests[at:fns[i + 1][1] if i + 1 < len(fns) else len(tests)]
        events = []
        for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*create_compute_specs\(([^;]*?)\);", body, re.S):
            events.append((m.start(), "specs_call", m))
        for m in re.finditer(r"let\s+(?:mut\s+)?(\w+)\s*=\s*ComputeSpecs\s*\{(.*?)\n        \};", body, re.S):
            events.append((m.start(), "specs_lit", m))
        for m in re.finditer(r"let\s+(\w+)\s*=\s*" + _STR + r";", body, re.S):
            events.append((m.start(), "str", m))
        for m in re.finditer(r"let\s+(\w+)\s*=\s*ComputeRequirements::from_str\(\s*(" + _STR + r"|\w+)\s*\)\s*\.unwrap\(\);", body, re.S):
            events.append((m.start(), "req", m))
        for m in re.finditer(r"assert!\(\s*(!?)\s*(\w+)\s*\.meets\(\s*&(\w+)\s*\)", body):
            events.append((m.start(), "assert", m))
        for m in re.finditer(r"assert!\(\s*ComputeRequirements::from_str\(\s*(" + _STR + r"|\w+)\s*\)\s*\.is_(err|ok)\(\)", body):
            events.append((m.start(), "parse", m))
        for m in re.finditer(r"let\s+(\w+)\s*=\s*ComputeRequirements::from_str\(\s*(" + _STR + r"|\w+)\s*\);", body, re.S):
            events.append((m.start(), "result", m))     # (a Result kept in a variable ...)
        for m in re.finditer(r"assert!\(\s*(\w+)\s*\.is_(err|ok)\(\)", body):
            events.append((m.start(), "result_assert", m))   # (... and asserted on)
        specs, strs, reqs, results, got = {}, {}, {}, {}, []
        for _at, kind, m in sorted(events, key=lambda e: e[0]):
            if kind == "specs_call":
                args = [a for a in re.split(r",\s*(?![^()]*\))", m.group(2).strip().rstrip(",")) if a.strip()]
                assert len(args) == 6, (name, args)
                specs[m.group(1)] = [_value(a) for a in args]
            elif kind == "specs_lit":
                specs[m.group(1)] = _struct_literal(m.group(2))
            elif kind == "str":
                strs[m.group(1)] = m.group(2)
            elif kind == "req":
                e = m.group(2)
                reqs[m.group(1)] = e[1:-1] if e.startswith('"') else strs[e]
            elif kind == "assert":
                got.append(("meets", specs[m.group(2)], reqs[m.group(3)], m.group(1) != "!"))
            elif kind == "result":
                e = m.group(2)
                results[m.group(1)] = e[1:-1] if e.startswith('"') else strs[e]
            elif kind == "result_assert":
                if m.group(1) in results:
                    got.append(("parse", results[m.group(1)], m.group(2)))
            else:
                e = m.group(1)
                got.append(("parse", e[1:-1] if e.startswith('"') else strs[e], m.groups()[-1]))
        if got:
            out[name] = got
    return out


def _fn_of(kat_name):
    return re.match(r"\w+", kat_name).group(0)


def main() -> int:
    if not os.path.exists(REF):
        print("reference not present; nothing to check")
        return 0
    src = open(REF).read()
    kats = json.load(open(os.path.join(HERE, "node_rs_kats.json")))
    ref = reference_vectors(src)
    bad = n = 0
    # ---- meets: per test function the same (specs, requirement, expected) triples in the same order, nothing left over
    by_fn = {}
    for k in kats["meets"]:
        by_fn.setdefault(_fn_of(k["name"]), []).append(("meets", k["specs"], k["req"], bool(k["expect"])))
    ref_meets = {f: [v for v in vs if v[0] == "meets"] for f, vs in ref.items()}
    ref_meets = {f: vs for f, vs in ref_meets.items() if vs}
    for f in sorted(set(by_fn) | set(ref_meets)):
        n += max(len(by_fn.get(f, [])), len(ref_meets.get(f, [])))
        if by_fn.get(f) != ref_meets.get(f):
            bad += 1
            print(f"[meets] {f}: fixture {by_fn.get(f)}\n          reference {ref_meets.get(f)}")
    # ---- parser errors: every `from_str(..).is_err()` of the reference is in the fixture and the other way round
    ref_err = sorted(v[1] for vs in ref.values() for v in vs if v[0] == "parse" and v[2] == "err")
    fix_err = sorted(k["req"] for k in kats["parser_errors"] if "file" not in k)
    n += len(fix_err)
    for r in fix_err:
        if r not in ref_err:
            bad += 1
            print(f"[parser_errors] not an is_err() vector of the reference: {r!r}")
    for r in ref_err:
        if r not in fix_err:
            bad += 1
            print(f"[parser_errors] the reference's is_err() vector is missing from the fixture: {r!r}")
    ref_ok = {v[1] for vs in ref.values() for v in vs if v[0] == "parse" and v[2] == "ok"}
    fixture_reqs = {k["req"] for k in kats["parser"]} | {k["req"] for k in kats["meets"]}
    for r in sorted(ref_ok - fixture_reqs):
        print(f"[parser] note: the reference's is_ok() vector {r!r} is not a fixture row of its own")
    # ---- parser (ok) vectors: the requirement string verbatim in the file the vector cites (the expected fields are
    # assert_eq! lines of that test; tests/test_oracle_kats.py and test_host_helpers.py compare them with both parsers)
    for k in kats["parser"]:
        n += 1
        text = open(os.path.join("/root/reference", k["file"])).read() if "file" in k else src
        if f'"{k["req"]}"' not in text:
            bad += 1
            print(f"[parser] {k['name']}: requirement string not found verbatim: {k['req']!r}")
    for k in kats["parser_errors"]:
        if "file" in k and f'"{k["req"]}"' not in open(os.path.join("/root/reference", k["file"])).read():
            bad += 1
            print(f"[parser_errors] {k['name']}: not found in {k['file']}")
    # ---- group-variable templates: every template string verbatim in the reference tests it cites
    gv = json.load(open(os.path.join(HERE, "group_vars_kats.json")))
    srcs = {"group_vars": open("/root/reference/crates/orchestrator/src/plugins/node_groups/tests.rs").read(),
            "upload_name": open("/root/reference/crates/orchestrator/src/api/routes/storage.rs").read()}
    for sec, src2 in srcs.items():
        for k in gv[sec]:
            n += 1
            if "$" in k["in"] and f'"{k["in"]}"' not in src2 and k["in"] not in src2:
                print(f"[{sec}] {k['name']}: template not found verbatim: {k['in']!r}")
                bad += 1
    n_meets = sum(len(v) for v in ref_meets.values())
    print(f"checked {n} vectors ({n_meets} meets triples with their polarity parsed out of {len(ref_meets)} reference tests, "
          f"{len(ref_err)} is_err vectors), {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
