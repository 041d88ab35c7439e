"""Structural ports of crates/orchestrator/src/plugins/node_groups/tests.rs onto the CPU oracle.

The reference tests assert group counts / sizes / membership sets (never ids or a node order,
because SMEMBERS order and the RNG are free) — the same invariants are asserted here.  Nodes the
reference adds "later" are present from the start with a non-Healthy status and flipped to Healthy.
"""
import numpy as np

from oracle import oracle_ffi as orc
from helpers import cfgs_of, group_sizes, mk_node, nodes_of, tasks_of

A = ["0x%s234567890123456789012345678901234567890" % d for d in "123456789"]


def test_group_formation_and_dissolution():
    # tests.rs:105-196
    nodes = nodes_of(mk_node(A[0], orc.ST_DISCOVERED), mk_node(A[1], orc.ST_DISCOVERED))
    st = orc.State(nodes, orc.make_config("test-config", 2, 5, None),
                   tasks=orc.make_task(0, ["test-config"]))
    assert st.try_form_new_groups() == 0
    st.set_node_status(0, orc.ST_HEALTHY)
    assert st.try_form_new_groups() == 0          # one healthy node < min 2
    st.set_node_status(1, orc.ST_HEALTHY)
    assert st.try_form_new_groups() == 1
    assert st.node_to_group[0] >= 0 and st.node_to_group[0] == st.node_to_group[1]
    st.set_node_status(0, orc.ST_DEAD)            # handle_status_change => dissolve whole group
    st.try_form_new_groups()
    assert st.node_to_group[0] < 0 and st.node_to_group[1] < 0 and st.n_groups == 0


def test_group_formation_with_multiple_configs():
    # tests.rs:199-300: 2-group first, the third node lands in a 1-group
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]), mk_node(A[2], orc.ST_DISCOVERED))
    cfgs = cfgs_of(orc.make_config("test-config-s", 2, 2, None), orc.make_config("test-config-xs", 1, 1, None))
    st = orc.State(nodes, cfgs, tasks=orc.make_task(0, ["test-config-s", "test-config-xs"]))
    st.try_form_new_groups()
    st.set_node_status(2, orc.ST_HEALTHY)
    st.try_form_new_groups()
    assert st.n_groups == 2 and group_sizes(st) == [1, 2]
    assert (st.node_to_group >= 0).all()


def test_group_formation_with_requirements():
    # tests.rs:303-506: a node that does not meet the requirement never joins; one that does, does
    specs_ok = orc.make_specs(8, "NVIDIA H100", 80000)
    specs_bad = orc.make_specs(1, "RTX 3090", 24000)
    cfg = orc.make_config("a100-group", 1, 1, "gpu:count=8;gpu:model=H100")
    nodes = nodes_of(mk_node(A[0], specs=specs_bad), mk_node(A[1], specs=specs_ok), mk_node(A[2]))
    st = orc.State(nodes, cfg, tasks=orc.make_task(0, ["a100-group"]))
    st.try_form_new_groups()
    n2g = st.node_to_group
    assert n2g[0] < 0 and n2g[1] >= 0 and n2g[2] < 0   # (Some req, None specs) => incompatible


def test_group_formation_with_max_size():
    # tests.rs:734-885: 5 nodes, min 2 / max 2 => two pairs + one leftover
    nodes = nodes_of(*[mk_node(a) for a in A[:5]])
    st = orc.State(nodes, orc.make_config("c", 2, 2, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    assert group_sizes(st) == [2, 2]
    assert int((st.node_to_group < 0).sum()) == 1


def test_node_cannot_be_in_multiple_groups():
    # tests.rs:993-1213: 3 nodes / max 2 => one pair + one leftover; a 4th node next tick pairs with
    # the leftover and the first group is untouched (the reference is incremental)
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]), mk_node(A[2]), mk_node(A[3], orc.ST_DISCOVERED))
    st = orc.State(nodes, orc.make_config("c", 2, 2, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    first = st.groups()
    assert len(first) == 1 and first[0][3] == [0, 1]
    st.set_node_status(3, orc.ST_HEALTHY)
    st.try_form_new_groups()
    g = st.groups()
    assert len(g) == 2 and g[0] == first[0] and sorted(g[1][3]) == [2, 3]
    n2g = st.node_to_group
    assert len(set(n2g.tolist())) == 2 and (n2g >= 0).all()


def test_reformation_on_death():
    # tests.rs:1215-1334
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]), mk_node(A[2], orc.ST_DISCOVERED))
    st = orc.State(nodes, orc.make_config("c", 2, 2, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    assert st.n_groups == 1
    st.set_node_status(0, orc.ST_DEAD)
    assert st.n_groups == 0
    st.set_node_status(2, orc.ST_HEALTHY)
    st.try_form_new_groups()
    g = st.groups()
    assert len(g) == 1 and sorted(g[0][3]) == [1, 2]


def test_get_idx_in_group():
    # tests.rs:1381-1446: GROUP_INDEX = rank of the address string in the BTreeSet
    nodes = nodes_of(mk_node(A[2]), mk_node(A[0]), mk_node(A[1]))
    st = orc.State(nodes, orc.make_config("c", 3, 3, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    idx = [st.filter_tasks(i)[1] for i in range(3)]
    assert idx == [2, 0, 1]
    t, gi, gs, nxt = st.filter_tasks(0)
    assert (t, gs) == (0, 3) and nxt == 1       # (2+1)%3 = rank 0 = node index 1


def test_building_largest_possible_groups():
    # tests.rs:1630-1800: groups are filled to max_group_size whenever enough nodes exist
    nodes = nodes_of(*[mk_node(a) for a in A[:7]])
    st = orc.State(nodes, orc.make_config("c", 2, 4, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    assert group_sizes(st) == [3, 4]


def test_group_formation_priority():
    # tests.rs:1803-1904: one 3-group + one 1-group, NOT four 1-groups — even when the small config
    # is listed first (constructor sort, mod.rs:150-164)
    nodes = nodes_of(*[mk_node(a) for a in A[:4]])
    cfgs = cfgs_of(orc.make_config("small-group", 1, 1, None), orc.make_config("large-group", 3, 3, None))
    st = orc.State(nodes, cfgs, tasks=orc.make_task(0, ["large-group", "small-group"]))
    st.try_form_new_groups()
    assert group_sizes(st) == [1, 3] and (st.node_to_group >= 0).all()


def test_multiple_groups_same_configuration():
    # tests.rs:1907-2009: 6 nodes, min=max=2 => 3 groups of 2
    nodes = nodes_of(*[mk_node(a) for a in A[:6]])
    st = orc.State(nodes, orc.make_config("c", 2, 2, None), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    assert group_sizes(st) == [2, 2, 2]


def test_allowed_topologies_and_same_task_for_group():
    # tests.rs:509-676, 888-990: both nodes of a group get the SAME task; topology filter applies
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]), mk_node(A[2]))
    cfgs = cfgs_of(orc.make_config("pair", 2, 2, None), orc.make_config("solo", 1, 1, None))
    tasks = tasks_of(orc.make_task(30, ["solo"]), orc.make_task(20, ["pair"]), orc.make_task(10, None))
    st = orc.State(nodes, cfgs, tasks=tasks)
    st.try_form_new_groups()
    assert [st.get_task_for_node(i) for i in range(3)] == [1, 1, 0]
    # a group whose configuration no task allows and no unrestricted task exists gets nothing
    st2 = orc.State(nodes, cfgs, tasks=tasks_of(orc.make_task(30, ["solo"])))
    st2.try_form_new_groups()
    assert [st2.get_task_for_node(i) for i in range(3)] == [-1, -1, 0]
    # restricted to an empty list => allowed nowhere
    st3 = orc.State(nodes, cfgs, tasks=tasks_of(orc.make_task(1, [])))
    st3.try_form_new_groups()
    assert [st3.get_task_for_node(i) for i in range(3)] == [-1, -1, -1]
    # node not in any group => empty (scheduler_impl.rs:208-209); default plugin chain => newest
    st4 = orc.State(nodes_of(mk_node(A[0], orc.ST_UNHEALTHY)), cfgs, tasks=tasks)
    assert st4.get_task_for_node(0) == -1
    assert st4.get_task_for_node(0, use_node_groups=False) == 0


def test_merge_only_compatible_groups():
    # tests.rs:2471-2634: with equal min_group_size the config WITH requirements is tried first, so
    # the 8-GPU nodes land in config-2, not the unconstrained config-1
    gpu = orc.make_specs(8, "RTX4090", 24)
    nodes = nodes_of(mk_node("0x" + "1" * 40), mk_node("0x" + "2" * 40),
                     mk_node("0x" + "3" * 40, specs=gpu), mk_node("0x" + "4" * 40, specs=gpu))
    cfgs = cfgs_of(orc.make_config("config-1", 1, 2, None), orc.make_config("config-2", 1, 2, "gpu:count=8"))
    st = orc.State(nodes, cfgs, tasks=tasks_of(orc.make_task(0, ["config-1"]), orc.make_task(0, ["config-2"])))
    st.try_form_new_groups()
    by_node = {m: g[2] for g in st.groups() for m in g[3]}
    assert by_node[2] == 1 and by_node[3] == 1 and by_node[0] == 0 and by_node[1] == 0
    st.try_merge_solo_groups()
    assert (st.node_to_group >= 0).all()


def _solo_then_merge(**policy):
    cfgs = cfgs_of(orc.make_config("merge-config", 1, 3, None))
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1], orc.ST_DISCOVERED), mk_node(A[2], orc.ST_DISCOVERED))
    st = orc.State(nodes, cfgs, tasks=tasks_of(orc.make_task(2, ["merge-config"]), orc.make_task(1, ["merge-config"])),
                   **policy)
    for i in range(3):      # nodes trickle in one per tick => three solo groups
        st.set_node_status(i, orc.ST_HEALTHY)
        st.try_form_new_groups()
    assert group_sizes(st) == [1, 1, 1]
    return st


def test_merge_solo_groups():
    # tests.rs:2171-2337 / 2339-2469: three solo groups merge into one 3-group which gets a task
    st = _solo_then_merge()
    assert st.try_merge_solo_groups() == 1
    g = st.groups()
    assert len(g) == 1 and sorted(g[0][3]) == [0, 1, 2] and g[0][4] == 0
    assert [st.get_task_for_node(i) for i in range(3)] == [0, 0, 0]


def test_no_merge_when_policy_disabled():
    # tests.rs:2636-2709 / 2014-2169
    st = _solo_then_merge(switching=False)
    assert st.try_merge_solo_groups() == 0 and group_sizes(st) == [1, 1, 1]
    # prefer_larger_groups=false: refuse only if a group already holds a task (mod.rs:277-287)
    st = _solo_then_merge(prefer_larger=False)
    assert st.get_task_for_node(0) == 0       # group of node 0 claims a task
    assert st.try_merge_solo_groups() == 0
    st = _solo_then_merge(prefer_larger=False)
    assert st.try_merge_solo_groups() == 1    # nobody holds a task => merge allowed


def test_edge_case_no_available_tasks():
    # tests.rs:2711-2781: merge still happens, the merged group is idle
    cfgs = cfgs_of(orc.make_config("merge-config", 1, 3, None))
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1], orc.ST_DISCOVERED))
    st = orc.State(nodes, cfgs, tasks=tasks_of())
    st.try_form_new_groups()
    st.set_node_status(1, orc.ST_HEALTHY)
    st.try_form_new_groups()
    assert st.try_merge_solo_groups() == 1
    g = st.groups()
    assert len(g) == 1 and g[0][4] == -1


def test_proximity_merging_prevents_wrong_nodes_grouping():
    # tests.rs:2861-3064: 2 Montreal + 2 Dallas solo groups merge city-wise
    mtl, dal = (45.5186, -73.5545), (32.7942, -96.7475)
    addrs = ["0x" + c * 40 for c in "1234"]
    locs = [mtl, dal, mtl, dal]
    nodes = nodes_of(*[mk_node(a, orc.ST_DISCOVERED, loc=l) for a, l in zip(addrs, locs)])
    st = orc.State(nodes, cfgs_of(orc.make_config("c", 1, 2, None)), tasks=orc.make_task(0, ["c"]))
    for i in range(4):
        st.set_node_status(i, orc.ST_HEALTHY)
        st.try_form_new_groups()
    assert group_sizes(st) == [1, 1, 1, 1]
    assert st.try_merge_solo_groups() == 2
    assert sorted(sorted(g[3]) for g in st.groups()) == [[0, 2], [1, 3]]


def test_proximity_carve_uses_seed_and_input_order_ties():
    # mod.rs:526-551 + Appendix A: seed = first compatible node WITH a location; stable sort keeps
    # input order among equal distances; nodes without a location sort last (f64::MAX)
    mtl, dal = (45.5186, -73.5545), (32.7942, -96.7475)
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1], loc=dal), mk_node(A[2], loc=mtl), mk_node(A[3], loc=dal),
                     mk_node(A[4], loc=dal), mk_node(A[5]))
    st = orc.State(nodes, cfgs_of(orc.make_config("c", 3, 3, None)), tasks=orc.make_task(0, ["c"]))
    st.try_form_new_groups()
    g = sorted(sorted(x[3]) for x in st.groups())
    assert g == [[0, 2, 5], [1, 3, 4]]     # seed 1 (first with location) + its two distance-0 peers


def test_no_proximity_is_first_come():
    # mod.rs:553-561
    mtl, dal = (45.5186, -73.5545), (32.7942, -96.7475)
    nodes = nodes_of(mk_node(A[0], loc=mtl), mk_node(A[1], loc=dal), mk_node(A[2], loc=mtl), mk_node(A[3], loc=dal))
    st = orc.State(nodes, cfgs_of(orc.make_config("c", 2, 2, None)), tasks=orc.make_task(0, ["c"]), proximity=False)
    st.try_form_new_groups()
    assert sorted(sorted(x[3]) for x in st.groups()) == [[0, 1], [2, 3]]


def test_disabled_config_forms_nothing_and_variants_agree():
    # get_available_configurations (mod.rs:399-418): only enabled configs are carved
    rng = np.random.default_rng(0)
    rows = []
    for i in range(200):
        loc = (float(rng.uniform(25, 60)), float(rng.uniform(-125, -65))) if rng.random() < 0.9 else None
        rows.append(mk_node("0x%040d" % i, specs=orc.make_specs(int(rng.choice([1, 8])), "NVIDIA H100", 80000), loc=loc))
    nodes = nodes_of(*rows)
    cfgs = cfgs_of(orc.make_config("a", 4, 4, "gpu:count=8"), orc.make_config("b", 2, 3, None))
    st = orc.State(nodes, cfgs, enabled=[0, 0])
    assert st.try_form_new_groups() == 0
    ref = orc.State(nodes, cfgs, reference_shaped=True)
    fast = orc.State(nodes, cfgs, reference_shaped=False)
    ref.try_form_new_groups()
    fast.try_form_new_groups()
    assert ref.groups() == fast.groups()
    assert ref.counters()[0] > fast.counters()[0]      # the reference shape recomputes Haversines


def test_threaded_pair_sweep_equals_the_sequential_one():
    """bench.py's all-cores CPU baseline splits the nodes of the reference-orientation sweep over threads"""
    import numpy as np
    from oracle import oracle_ffi as orc
    from protocol_amd.swarm import make_swarm
    sw = make_swarm(11, 700, 900)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    st.try_form_new_groups()
    cfg_of_node = np.full(sw.W, -1, dtype=np.int32)
    for _, _, c, mem, _ in st.groups():
        cfg_of_node[mem] = c
    a = orc.pair_sweep_per_worker(tasks, cfgs, cfg_of_node)
    for threads in (2, 5, 64):
        b = orc.pair_sweep_per_worker(tasks, cfgs, cfg_of_node, threads=threads)
        assert (a[0] == b[0]).all() and (a[1] == b[1]).all()


def test_webhook_feed_follows_the_emission_sites():
    """send_group_created for every formed group in formation order (mod.rs:612-625); a death dissolves the group:
    send_group_destroyed with its members (mod.rs:1469-1481); a merge sends destroyed for each solo group of the
    batch, then created for the merged group (mod.rs:974-1000).  Members always in group.nodes (BTreeSet) order."""
    nodes = nodes_of(*[mk_node(a) for a in A[:5]])
    cfgs = cfgs_of(orc.make_config("pair", 2, 2, None), orc.make_config("solo", 1, 1, None))
    st = orc.State(nodes, cfgs, tasks=orc.make_task(0, ["pair", "solo"]))
    assert st.try_form_new_groups() == 3                     # two pairs and a solo group
    ev = st.drain_events()
    groups = st.groups()
    assert ev == [(1, gid, cfg, mem) for _s, gid, cfg, mem, _t in groups]
    assert [len(m) for _k, _i, _c, m in ev] == [2, 2, 1]
    assert st.drain_events() == []
    st.set_node_status(0, orc.ST_DEAD)                       # its pair dissolves, the partner is free again
    (k, gid, _cfg, mem), = st.drain_events()
    assert (k, gid, mem) == (2, groups[0][1], groups[0][3])
    assert st.try_form_new_groups() == 1                     # the partner becomes a solo group
    (k, _gid, cfg, mem), = st.drain_events()
    assert (k, cfg, len(mem)) == (1, 1, 1)
    assert st.try_merge_solo_groups() == 1                   # the two solo groups merge into a pair
    ev = st.drain_events()
    assert [k for k, *_ in ev] == [2, 2, 1]
    assert sorted(ev[0][3] + ev[1][3]) == ev[2][3] and ev[2][2] == 0


def _enabled_by_tasks(cfgs, tasks):
    """available_node_group_configs as the task observers leave it: on_task_created enables every topology a task
    lists (mod.rs:1224-1243), on_task_deleted disables one when no remaining task lists it (mod.rs:1293-1318) — i.e.
    the union of the current tasks' allowed_topologies (an unrestricted task enables nothing)"""
    names = [bytes(n).rstrip(b"\0").decode() for n in cfgs["name"]]
    on = np.zeros(len(cfgs), dtype=np.uint8)
    for t in tasks:
        if t["restricted"]:
            for k in range(int(t["n_topologies"])):
                name = bytes(t["topologies"][k]).rstrip(b"\0").decode()
                if name in names:
                    on[names.index(name)] = 1
    return on


def test_group_scheduling_without_tasks():
    # tests.rs:679-731: no task ever enabled the configuration, so nothing forms and nobody gets anything
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]))
    cfgs = cfgs_of(orc.make_config("test-config", 2, 5, None))
    tasks = tasks_of()
    st = orc.State(nodes, cfgs, enabled=_enabled_by_tasks(cfgs, tasks), tasks=tasks)
    assert st.get_task_for_node(0) == -1
    assert st.try_form_new_groups() == 0 and st.n_groups == 0
    assert st.get_task_for_node(0) == -1 and st.get_task_for_node(1) == -1


def test_task_observer():
    # tests.rs:1467-1628: configurations follow the tasks; deleting a task dissolves the group that works on it
    nodes = nodes_of(mk_node(A[0]), mk_node(A[1]), mk_node(A[2], orc.ST_DISCOVERED))
    cfgs = cfgs_of(orc.make_config("test-config", 1, 1, None), orc.make_config("test-config2", 1, 1, None))
    tasks = tasks_of()
    st = orc.State(nodes, cfgs, enabled=_enabled_by_tasks(cfgs, tasks), tasks=tasks)
    assert st.try_form_new_groups() == 0 and st.node_to_group[0] < 0          # :1513-1520 no task, no group
    task, task2 = orc.make_task(1, ["test-config"]), orc.make_task(2, ["test-config2"])
    tasks = tasks_of(task2, task)                                              # get_all_tasks: newest first
    st.set_tasks(tasks)
    st.set_enabled(_enabled_by_tasks(cfgs, tasks))
    assert list(st.enabled) == [1, 1]                                          # :1566-1569 both available, in order
    assert st.try_form_new_groups() == 2
    n2g = st.node_to_group
    assert n2g[0] >= 0 and n2g[1] >= 0 and n2g[0] != n2g[1]                    # :1571-1583
    st.set_node_status(2, orc.ST_HEALTHY)                                      # node 3 appears
    assert st.try_form_new_groups() == 1 and st.node_to_group[2] >= 0          # :1585-1600
    # the third group works on `task` (assign_task_to_group, :1612-1616): it is a test-config group (first
    # available configuration), whose only applicable task that is
    assert st.get_task_for_node(2) == 1
    working_on_task = [g for g in st.groups() if g[4] == 1]
    # delete `task`: every group working on it dissolves immediately (:1618-1636); test-config is no longer listed
    tasks = tasks_of(task2)
    st.set_tasks(tasks)
    st.remap_tasks([0, -1])
    st.set_enabled(_enabled_by_tasks(cfgs, tasks))
    assert st.node_to_group[2] < 0 and list(st.enabled) == [0, 1]
    assert all(st.node_to_group[m] < 0 for g in working_on_task for m in g[3])
