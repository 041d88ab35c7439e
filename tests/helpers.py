"""Builders shared by the parity tests (oracle-side AoS rows)."""
import numpy as np

from oracle import oracle_ffi as orc


def mk_node(addr: str, status=orc.ST_HEALTHY, specs=None, loc=None, p2p=True) -> np.ndarray:
    """Mirror of create_test_node / create_test_node_with_location
    (crates/orchestrator/src/plugins/node_groups/tests.rs:24-56): p2p_id is always Some there."""
    n = np.zeros(1, dtype=orc.node_dt)
    n["address"] = addr.encode()
    n["status"] = status
    n["has_p2p"] = int(p2p)
    if specs is not None:
        n["has_specs"] = 1
        n["specs"] = specs
    if loc is not None:
        n["has_location"] = 1
        n["latitude"], n["longitude"] = loc
    return n


def nodes_of(*rows) -> np.ndarray:
    return np.concatenate(rows)


def cfgs_of(*rows) -> np.ndarray:
    return np.concatenate(rows)


def tasks_of(*rows) -> np.ndarray:
    return np.concatenate(rows) if rows else np.zeros(0, dtype=orc.task_dt)


def group_sizes(state) -> list:
    return sorted(len(g[3]) for g in state.groups())


# ------------------------------------------------------------------ engine <-> oracle adapters

def engine_groups(eng):
    """[(id, config, members in BTreeSet order, task or -1)] in creation order, like State.groups()[1:]"""
    _, groups, members = eng.get_groups()
    out = []
    for g in groups:
        b, n = int(g["member_begin"]), int(g["n_members"])
        t = int(g["task"])
        out.append((int(g["id"]), int(g["config"]), members[b:b + n].tolist(), -1 if t == 0xFFFFFFFF else t))
    return out


def oracle_groups(state):
    return [(gid, cfg, mem, task) for (_slot, gid, cfg, mem, task) in state.groups()]


def oracle_state_for(sw, **kw):
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    return orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, **kw)
