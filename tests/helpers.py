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
