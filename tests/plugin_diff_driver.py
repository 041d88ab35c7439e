"""Differential run of the two transcriptions of rust/gpu_match_plugin.rs — tests/shim_replay.py (Python, statement for
statement) and protocol_amd/plugin (C++, compiled) — over tests/cpp/mock_engine.cpp, which logs every C-ABI call with its
arguments: the same store schedule must produce the same calls in the same order, the same answer to every heartbeat and
the same webhook feed.  Run as a script (tests/test_plugin_cpp.py starts it in a process of its own, because it points
protocol_amd.engine at the mock library):

    python tests/plugin_diff_driver.py <library with the mock engine, pm_host.cpp, the plugin and its C face>
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(lib_path: str, seeds=(3, 4, 5)) -> int:
    from protocol_amd import build as B
    from protocol_amd import engine as E
    B.LIB_PATH = lib_path                 # (this process only: the engine library IS the mock here)
    B.needs_build = lambda: False
    E._lib = None
    from protocol_amd import host
    from protocol_amd.swarm import make_swarm
    import plugin_cxx
    import shim_replay
    L = E.lib()
    plugin_cxx.plugin_lib(lib_path)
    L.pm_mock_calls.restype = C.c_char_p
    L.pm_mock_reset_calls.restype = None

    ALL = 0xFFFFFFFFFFFFFFFF
    W_store = 160
    rng = np.random.default_rng(11)

    def run(kind: str, seed: int):
        """the whole schedule on one of the two; returns [(step, calls, heartbeats, events)]"""
        sw = make_swarm(4 + seed, 60, W_store)
        L.pm_mock_reset_calls()
        shim = shim_replay.ShimReplay(sw) if kind == "py" else plugin_cxx.PluginCxx(sw)
        r = np.random.default_rng(seed)
        trace = []
        masks, created, uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
        cur = [(int(m), int(c), int(u)) for m, c, u in zip(masks, created, uid)]

        def enabled():
            e = 0
            for m, _c, _u in cur:
                if m != ALL:
                    e |= m
            return e

        def step(name, heartbeat_nodes=()):
            calls = L.pm_mock_calls().decode().splitlines()
            L.pm_mock_reset_calls()
            beats = [shim.filter_tasks(int(n)) for n in heartbeat_nodes]
            L.pm_mock_reset_calls()      # (look-ups are not logged; keep the log per step clean anyway)
            ev = list(shim.events)
            shim.events.clear()
            trace.append((name, calls, beats, ev))

        step("new")
        shim.sync_tasks(masks, created, uid, enabled())
        step("sync_tasks")
        status = sw.status.copy()
        healthy = {n for n in range(W_store) if status[n] == 2}
        present = list(range(100))
        snap = np.array(present)
        r.shuffle(snap)
        def read_surface(name):
            """the read surface as a step of its own: the answers go where the heartbeats go"""
            addr = sw.address_strings()
            strip = lambda g: None if g is None else (g["id"], g["config"], g["created_at"], tuple(g["nodes"]))
            groups = shim.get_all_groups()
            ans = [("all", [strip(g) for g in groups]), ("map", sorted(shim.get_all_node_group_mappings().items()))]
            for n in present[:30]:
                r = shim.get_node_group(addr[n])
                ans.append(("node", n, None if r is None else (r[0], strip(r[1]))))
            batch = shim.get_node_groups_batch([addr[n] for n in present[:30]] + ["0xnobody"])
            ans.append(("batch", sorted((a, strip(g)) for a, g in batch.items())))
            for g in groups[:3]:
                ans.append(("by id", strip(shim.get_group_by_id(g["id"]))))
            ans.append(("by id", shim.get_group_by_id("0123"), shim.get_group_by_id("zz")))
            if groups:
                shim.dissolve_group(groups[len(groups) // 2]["id"])
                shim.dissolve_group(groups[len(groups) // 2]["id"])      # the second time: no such group, Ok(())
                shim.dissolve_group("not-an-id")
                ans.append(("after dissolve", [strip(g) for g in shim.get_all_groups()]))
            calls = L.pm_mock_calls().decode().splitlines()
            L.pm_mock_reset_calls()
            ev = list(shim.events)
            shim.events.clear()
            trace.append((name, calls, ans, ev))

        shim.set_clock(1111)
        shim.sync_nodes(snap, healthy)
        step("sync_nodes A")
        shim.tick()
        step("tick 1", present)
        read_surface("read surface 1")
        t_max = int(created.max())
        for k in range(3):
            src = int(r.integers(0, len(masks)))
            t_max += 1
            new = (int(masks[src]), t_max, (1 << 40) + k)
            cur.insert(0, new)
            shim.on_task_created(new[0], new[1], new[2], enabled())
            step(f"on_task_created {k}", present[:20])
        # the task most heartbeats were answered with goes
        answered = [b for b in next(t for t in trace if t[0] == "tick 1")[2] if b is not None]
        victim = max(set(answered), key=answered.count)
        cur = [t for t in cur if t[2] != victim]
        shim.on_task_deleted(victim, enabled())
        step("on_task_deleted", present)
        for n in (present[3], present[17], present[40]):
            healthy.discard(n)
            shim.handle_status_change(n, healthy=False, dead=True)
        shim.handle_status_change(present[5], healthy=False, dead=False)
        healthy.discard(present[5])
        step("status changes", present)
        # discovery rewrote a few rows; some nodes left; new ones came
        rewritten = [n for n in present[50:] if sw.has_specs[n] and sw.ram_some[n]][:6]
        for n in rewritten:
            sw.ram_mb[n] += 1
        shim.packed_all = host.pack_workers(sw)
        gone = present[60:66]
        present = [n for n in present if n not in gone] + list(range(100, 140))
        snap = np.array(present)
        r.shuffle(snap)
        shim.sync_nodes(snap, healthy)
        step("sync_nodes B", present)
        shim.set_clock(2222)
        shim.tick()
        step("tick 2", present)
        read_surface("read surface 2")
        shim.sync_nodes(snap, healthy)   # nothing changed
        step("sync_nodes B again")
        n_rows = len(shim.rows)
        shim.close()
        return trace, n_rows

    bad = 0
    totals = [0, 0, 0, 0, 0]
    # the C face reports the constructor's refusals (the reference's two panics) instead of letting them cross the ABI
    import copy
    sw_bad = make_swarm(1, 4, 8)
    for mutate, want in ((lambda c: c.__setitem__(1, (c[0][0],) + tuple(c[1][1:])), "Configuration names must be unique"),
                         (lambda c: c.__setitem__(0, (c[0][0], 5, 2, c[0][3])), "Plugin configuration is invalid")):
        sw_b = copy.deepcopy(sw_bad)
        mutate(sw_b.configs)
        try:
            plugin_cxx.PluginCxx(sw_b)
            print("no refusal for", want)
            bad += 1
        except RuntimeError as ex:
            if want not in str(ex):
                print("refusal says", ex, "instead of", want)
                bad += 1
    for seed in seeds:
        py, rows_py = run("py", seed)
        cxx, rows_cxx = run("cxx", seed)
        if rows_py != rows_cxx:
            print(f"seed {seed}: known rows differ:", rows_py, rows_cxx)
            bad += 1
        for (name, c1, b1, e1), (_n, c2, b2, e2) in zip(py, cxx):
            if c1 != c2:
                bad += 1
                print(f"seed {seed} [{name}] C-ABI calls differ")
                for i in range(max(len(c1), len(c2))):
                    a = c1[i] if i < len(c1) else "<none>"
                    b = c2[i] if i < len(c2) else "<none>"
                    if a != b:
                        print("   py :", a[:300])
                        print("   c++:", b[:300])
                        break
            if b1 != b2:
                bad += 1
                print(f"seed {seed} [{name}] heartbeats differ:", [(i, x, y) for i, (x, y) in enumerate(zip(b1, b2)) if x != y][:5])
            if e1 != e2:
                bad += 1
                print(f"seed {seed} [{name}] webhook feeds differ: {len(e1)} vs {len(e2)} events; first:",
                      next(((x, y) for x, y in zip(e1, e2) if x != y), None))
        totals[0] += len(py)
        totals[1] += sum(len(c) for _n, c, _b, _e in py)
        totals[2] += sum(len(b) for n, _c, b, _e in py if not n.startswith("read surface"))
        totals[3] += sum(x is not None for n, _c, b, _e in py for x in b if not n.startswith("read surface"))
        totals[4] += sum(len(e) for _n, _c, _b, e in py)
    print(f"seeds {len(seeds)}, steps {totals[0]}, C-ABI calls {totals[1]}, heartbeats {totals[2]} ({totals[3]} served), webhook events {totals[4]}")
    print("DIFF OK" if not bad else f"DIFF FAILED ({bad})")
    return 0 if not bad else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
