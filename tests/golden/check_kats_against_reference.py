"""Transcription check for tests/golden/node_rs_kats.json.

Run in the build container (where /root/reference exists): every requirement string and every
model string of the fixture must appear verbatim in the reference test module it cites.  The GPU box
has no /root/reference, so this is a container-only integrity check, not a run-time dependency.
"""
import json
import os
import sys

REF = "/root/reference/crates/shared/src/models/node.rs"
HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    if not os.path.exists(REF):
        print("reference not present; nothing to check")
        return 0
    src = open(REF).read()
    kats = json.load(open(os.path.join(HERE, "node_rs_kats.json")))
    bad = 0
    for sec in ("parser", "parser_errors", "meets"):
        for k in kats[sec]:
            if "file" in k:  # a vector from another reference file (JSON there: the string sits inside a raw literal)
                other = open(os.path.join("/root/reference", k["file"])).read()
                if f'"{k["req"]}"' not in other:
                    print(f"[{sec}] {k['name']}: requirement string not found verbatim in {k['file']}: {k['req']!r}")
                    bad += 1
                continue
            if f'"{k["req"]}"' not in src:
                print(f"[{sec}] {k['name']}: requirement string not found verbatim: {k['req']!r}")
                bad += 1
            m = k.get("specs", [None, None])[1]
            if m is not None and f'"{m}"' not in src:
                print(f"[{sec}] {k['name']}: model string not found verbatim: {m!r}")
                bad += 1
    n = sum(len(kats[s]) for s in ("parser", "parser_errors", "meets"))
    # group-variable templates: every template string must appear verbatim in the reference tests it cites
    gv = json.load(open(os.path.join(HERE, "group_vars_kats.json")))
    srcs = {"group_vars": open("/root/reference/crates/orchestrator/src/plugins/node_groups/tests.rs").read(),
            "upload_name": open("/root/reference/crates/orchestrator/src/api/routes/storage.rs").read()}
    for sec, src2 in srcs.items():
        for k in gv[sec]:
            n += 1
            if "$" in k["in"] and f'"{k["in"]}"' not in src2 and k["in"] not in src2:
                print(f"[{sec}] {k['name']}: template not found verbatim: {k['in']!r}")
                bad += 1
    print(f"checked {n} vectors, {bad} mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
