"""GpuMatchPlugin::tick_dist — the Python-free multi-GPU tick of the compiled host side (protocol_amd/plugin: the five calls
of INTEGRATION.md "Multi-GPU" around ONE all-gather) — on the real engine:

  * N ranks in one process (LocalAllGather: device copies ordered by events), every rank fed the same store events: every
    rank's heartbeats, groups and read surface against the ORACLE, over intervals with deaths and new nodes;
  * the RCCL binding on whatever the box has: a world-of-one communicator whose ncclAllGather really runs on the tick's
    stream (one GPU), and two processes over RCCL / xGMI (needs two GPUs: skipped here, run by an 8-GPU node).
"""
import os
import pickle
import threading

import numpy as np
import pytest
import torch

from oracle import oracle_ffi as orc
from protocol_amd.swarm import make_swarm
from helpers import engine_groups, oracle_groups
import plugin_cxx
from plugin_cxx import PluginCxx

pytestmark = pytest.mark.gpu


def _tick_all(ranks, comms):
    errs = []

    def main(r):
        try:
            ranks[r].tick_dist(comms[r])
        except Exception as ex:   # (pmx_tick_dist released the others: nobody hangs at the exchange)
            errs.append((r, repr(ex)))

    th = [threading.Thread(target=main, args=(r,)) for r in range(len(ranks))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("world", [2, 4])
def test_cxx_tick_dist_in_process_ranks_track_the_oracle(world):
    W, T = 3000, 400
    sw = make_swarm(51, T, W)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    status_all = nodes["status"].copy()
    nodes["status"][2400:] = 0                                       # the last 600 nodes join in the second interval
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    healthy = {n for n in range(W) if status_all[n] == orc.ST_HEALTHY}
    ranks = [PluginCxx(sw) for _ in range(world)]
    comms = plugin_cxx.local_world(world)
    for p in ranks:
        p.sync_tasks(sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy(), sw.enabled_mask())
    addr = sw.address_strings()
    rng = np.random.default_rng(3)
    known = 2400
    for interval in range(3):
        if interval == 1:
            known = W
            for i in range(2400, W):
                st.set_node_status(i, int(status_all[i]))
        if interval == 2:   # deaths: replicated calls on every rank
            for w in rng.choice(sorted(healthy), size=60, replace=False):
                healthy.discard(int(w))
                st.set_node_status(int(w), orc.ST_DEAD)
                for p in ranks:
                    p.handle_status_change(int(w), healthy=False, dead=True)
        for p in ranks:
            p.sync_nodes(np.arange(known), healthy)                  # every rank is fed every store event
        _tick_all(ranks, comms)
        st.try_form_new_groups()
        st.try_merge_solo_groups()
        want_tasks = [st.get_task_for_node(w) for w in range(known)]  # (the oracle claims on this call)
        want_groups = sorted(oracle_groups(st))
        want_all = sorted((("%x" % gid, [addr[i] for i in mem]) for (_s, gid, _c, mem, _t) in st.groups()), key=lambda g: g[0])
        ev = st.drain_events()
        for r, p in enumerate(ranks):
            assert sorted(engine_groups(p.eng)) == want_groups, f"interval {interval} rank {r}: groups differ from the oracle"
            for w in range(known):
                got = p.filter_tasks(w)                              # Scheduler::get_task_for_node on rank r's full table
                assert got == (None if want_tasks[w] < 0 else int(sw.task_uid[want_tasks[w]])), (interval, r, w)
            assert [(g["id"], g["nodes"]) for g in p.get_all_groups()] == want_all
            # webhooks: rank 0 delivers what the reference would send, the others stay silent
            assert p.events == (ev if r == 0 else []), f"interval {interval} rank {r}"
            p.events.clear()
    for c in comms:
        plugin_cxx.comm_destroy(c)
    for p in ranks:
        p.close()


def test_rccl_binding_runs_a_collective_on_this_box(tmp_path):
    """What a one-GPU box can verify of RcclAllGather: librccl loads next to the engine, a communicator initialises through
    the id-file rendezvous, ncclAllGather is accepted on the tick's stream and completes with the data in place."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = plugin_cxx.dist_lib()
    rc = L.pmx_rccl_self_test(0, 1 << 20, str(tmp_path / "nccl.id").encode())
    assert rc == 0, L.pmx_last_error_dist().decode()


def _rccl_rank(rank, world, id_file, out_dir):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sw = make_swarm(52, 3000, 5000)
    p = PluginCxx(sw, device=rank)
    p.sync_tasks(sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy(), sw.enabled_mask())
    healthy = {n for n in range(sw.W) if sw.status[n] == 2}
    p.sync_nodes(np.arange(sw.W), healthy)
    comm = plugin_cxx.rccl_comm(rank, world, rank, id_file)
    p.tick_dist(comm)
    p.tick_dist(comm)                                                # a second tick on the same communicator (nothing to form)
    with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
        pickle.dump((engine_groups(p.eng), [p.filter_tasks(w) for w in range(sw.W)]), f)
    plugin_cxx.comm_destroy(comm)
    p.close()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="two processes over RCCL need two GPUs (one process per GPU)")
def test_cxx_two_processes_over_rccl(tmp_path):
    """tick_dist over RcclAllGather, one process per GPU: both ranks end with the oracle's groups and every worker's task."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_rccl_rank, args=(r, world, str(tmp_path / "nccl.id"), str(tmp_path))) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(600)
        assert pr.exitcode == 0
    sw = make_swarm(52, 3000, 5000)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    st = orc.State(nodes, cfgs, enabled=enabled, tasks=tasks, reference_shaped=False)
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    want_tasks = [st.get_task_for_node(w) for w in range(sw.W)]
    want = [None if t < 0 else int(sw.task_uid[t]) for t in want_tasks]
    for r in range(world):
        with open(os.path.join(str(tmp_path), f"rank{r}.pkl"), "rb") as f:
            groups, served = pickle.load(f)
        assert sorted(groups) == sorted(oracle_groups(st)), f"rank {r}"
        assert served == want, f"rank {r}"
