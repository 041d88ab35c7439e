"""The multi-GPU tick on the real engine (libpm_engine.so through the C ABI), on ONE MI355X: the ranks are engines
that share the device, so everything but the xGMI transport is the production path — ownership, the replicated
carve (one streaming launch per rank, side by side on the one GPU here), the segment layout of the table exchange,
the owned-rows pair sweep and the table scatter.

  * in-process ranks: N engines, N threads, an all-gather made of device copies between their buffers;
  * two processes over torch.distributed (gloo, host-staged) sharing the GPU: the real driver end to end.

Bar: every rank's groups and full published table are bit-identical to the single-GPU engine's (which the other
-m gpu tests pin to the oracle), and to the oracle directly."""
import os
import pickle
import socket
import threading

import numpy as np
import pytest
import torch

from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.dist import EngineLocal, ShardedEngine
from protocol_amd.swarm import baseline_config, make_swarm
from helpers import engine_groups, oracle_groups, oracle_state_for

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF


def _table(eng, W):
    t = np.zeros(W, dtype=E.assignment_dt)
    for w in range(W):
        a = eng.lookup(w)
        t[w] = (a.task, a.group_slot, a.group_index, a.group_size, a.next_worker, a.group_id)
    return t


def _single(sw, **kw):
    eng = E.Engine(**kw)
    host.load_swarm(eng, sw)
    stats = eng.tick()
    out = engine_groups(eng), _table(eng, sw.W), stats
    eng.close()
    return out


class _InProcGroup:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world


class _InProcExchanger:
    """all-gather between engines of one process: every rank posts its send tensor, then copies all of them"""

    def __init__(self, group, rank):
        self.g, self.rank, self.world = group, rank, group.world

    def all_gather(self, recv, send):
        if self.world == 1:
            if recv.data_ptr() != send.data_ptr():
                recv.copy_(send)
            return
        self.g.slots[self.rank] = send
        torch.cuda.current_stream().synchronize()       # my contribution is complete
        self.g.barrier.wait()
        n = send.numel()
        for r in range(self.world):
            recv[r * n:(r + 1) * n].copy_(self.g.slots[r])
        torch.cuda.current_stream().synchronize()
        self.g.barrier.wait()                            # nobody rewrites its segment before everyone has read it


def _run_in_process(sw, world, ticks=1, between=None, finish=None, per_task=True, **kw):
    """finish(eng, stats, exchanges) -> what out[r] holds instead of the groups and the table themselves (big swarms)"""
    grp = _InProcGroup(world)
    out = [None] * world
    errs = []

    def rank_main(r):
        try:
            eng = E.Engine(**kw)
            host.load_swarm(eng, sw)
            se = ShardedEngine(EngineLocal(eng, torch.device("cuda", 0)), sw.address,
                               exchanger=_InProcExchanger(grp, r))
            bids = se.match_per_task() if per_task else None
            stats = None
            for k in range(ticks):
                if between and k:
                    between(eng, k)
                stats = se.tick()
            out[r] = finish(eng, stats, se.exchanges) if finish else (engine_groups(eng), _table(eng, sw.W), stats, bids, se.exchanges)
            eng.close()
        except Exception as ex:  # a dead rank must not leave the others at the barrier
            errs.append((r, repr(ex)))
            grp.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    return out


def test_stepwise_tick_on_one_rank_is_the_plain_tick():
    sw = make_swarm(21, 3000, 2500)
    g1, t1, s1 = _single(sw, group_id_seed=3)
    (g2, t2, s2, _pt, n_x), = _run_in_process(sw, 1, group_id_seed=3)
    assert g1 == g2 and np.array_equal(t1, t2)
    assert s2["n_groups"] == s1["n_groups"] and s2["carve_steps"] == s1["carve_steps"]


@pytest.mark.parametrize("world", [2, 3, 8])
def test_in_process_ranks_equal_the_single_gpu_engine_and_the_oracle(world):
    sw = baseline_config(1, seed=4) if world < 8 else make_swarm(22, 5000, 4000)
    g1, t1, s1 = _single(sw, group_id_seed=9)
    st = oracle_state_for(sw, reference_shaped=False, group_id_seed=9)
    st.try_form_new_groups()
    st.try_merge_solo_groups()
    want = [g[:3] for g in oracle_groups(st)]
    res = _run_in_process(sw, world, group_id_seed=9)
    # per-task bids before the carve: folded over the ranks == one engine over all workers
    eng = E.Engine(group_id_seed=9)
    host.load_swarm(eng, sw)
    best1, count1 = eng.match_per_task()
    eng.close()
    for r, (g, t, s, (best, count), n_x) in enumerate(res):
        assert [x[:3] for x in g] == want, f"rank {r}: groups differ from the oracle"
        assert g == g1 and np.array_equal(t, t1), f"rank {r}: differs from the single-GPU engine"
        assert s["host_resolved_steps"] == 0 and s["n_groups"] == s1["n_groups"]
        assert np.array_equal(best, best1) and np.array_equal(count, count1), f"rank {r}: folded bids"
        assert n_x == 1                                # ONE exchange per tick: the published rows (the carve is replicated)


@pytest.mark.parametrize("world", [4, 8])
def test_in_process_ranks_config2_against_the_digest(world):
    """BASELINE configs[3]: the 1M-task x 100k-worker swarm of configs[2] hash-sharded over N ranks (in-process here: N
    engines on the one GPU, the exchange as device copies) — every rank's groups and full published table against the
    ORACLE's committed digests (tests/golden/scale_digests.json cfg2_seed1: group list, per-worker task, GROUP_INDEX /
    SIZE / NEXT), with exactly one exchange in the tick."""
    import json
    from test_gpu_scale import _engine_digests, _sha
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "scale_digests.json")))["cfg2_seed1"]
    sw = baseline_config(gold["config"], seed=gold["seed"])
    assert (sw.W, sw.T) == (gold["W"], gold["T"]) == (100000, 1000000)

    def finish(eng, stats, exchanges):
        got = engine_groups(eng)
        gsha, tbl = _engine_digests(sw, eng)
        return dict(stats=stats, exchanges=exchanges, n_groups=len(got), first=[[g[0], g[1], g[2]] for g in got[:1000]],
                    last=[[g[0], g[1], g[2]] for g in got[-1000:]], groups_sha=gsha,
                    task_sha=_sha(tbl["task"].astype(np.uint32)),
                    table_sha=_sha(tbl["group_index"].astype(np.uint32), tbl["group_size"].astype(np.uint32), tbl["next_worker"].astype(np.uint32)))

    res = _run_in_process(sw, world, finish=finish, per_task=False, group_id_seed=gold["seed"])
    for r, got in enumerate(res):
        assert got["exchanges"] == 1, f"rank {r}"
        assert got["stats"]["n_formed"] == gold["n_formed"] and got["stats"]["n_merged"] == gold["n_merged"], f"rank {r}"
        assert got["n_groups"] == gold["n_groups"] and got["first"] == gold["first_groups"] and got["last"] == gold["last_groups"], f"rank {r}"
        assert got["groups_sha"] == gold["groups_sha256"], f"rank {r}: group list differs from the oracle"
        assert got["task_sha"] == gold["task_sha256"], f"rank {r}: per-worker task column differs from the oracle"
        assert got["table_sha"] == gold["table_sha256"], f"rank {r}: GROUP_INDEX / SIZE / NEXT differ from the oracle"
        assert got["stats"]["host_resolved_steps"] == 0


def test_in_process_ranks_big_lists_and_forced_host_resolves():
    """30k workers (big-list mode) and the debug hook that sends every 5th step to the exact host path: the
    replicated host resolves must keep the ranks in lock step"""
    sw = make_swarm(2, 2000, 30000, zipf=True)
    g1, t1, _ = _single(sw)
    for (g, t, s, _pt, _n) in _run_in_process(sw, 2):
        assert g == g1 and np.array_equal(t, t1)
    sw = make_swarm(23, 1500, 3000)
    g1, t1, s1 = _single(sw, debug_uncertain_every=5)
    assert s1["host_resolved_steps"] > 0
    for (g, t, s, _pt, _n) in _run_in_process(sw, 3, debug_uncertain_every=5):
        assert g == g1 and np.array_equal(t, t1) and s["host_resolved_steps"] == s1["host_resolved_steps"]


def test_in_process_ranks_follow_churn():
    """second and third tick after deaths / rejoins (replicated status calls): sticky groups, re-carved leftovers"""
    sw = make_swarm(24, 2000, 3000)
    flags = host.worker_flags(sw).astype(np.int64)
    healthy = np.nonzero(sw.status == 2)[0]
    victims = np.random.default_rng(3).choice(healthy, size=40, replace=False)

    def between(eng, k):
        for w in victims[(k - 1) * 20:k * 20]:
            eng.on_worker_status(int(w), int(flags[w] & ~E.W_HEALTHY), True)

    eng = E.Engine()
    host.load_swarm(eng, sw)
    eng.tick()
    for k in (1, 2):
        between(eng, k)
        eng.tick()
    g1, t1 = engine_groups(eng), _table(eng, sw.W)
    eng.close()
    for (g, t, s, _pt, _n) in _run_in_process(sw, 2, ticks=3, between=between):
        assert sorted(g) == sorted(g1) and np.array_equal(t["task"], t1["task"])
        assert np.array_equal(t["group_id"], t1["group_id"])


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _proc_main(rank, world, port, out_dir, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = rank if backend == "nccl" else 0           # nccl (= RCCL): one GPU per rank; gloo: the ranks share GPU 0
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        sw = make_swarm(25, 4000, 6000)
        eng = E.Engine(device=dev, group_id_seed=5)
        host.load_swarm(eng, sw)
        se = ShardedEngine(EngineLocal(eng, torch.device("cuda", dev)), sw.address)  # TorchExchanger (gloo: host-staged)
        stats = se.tick()
        best, count = se.match_per_task()
        with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
            pickle.dump((engine_groups(eng), _table(eng, sw.W), stats, best, count, se.exchanges), f)
        eng.close()
    finally:
        dist.destroy_process_group()


def _two_processes(tmp_path, backend):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_proc_main, args=(world, _free_port(), str(tmp_path), backend), nprocs=world, join=True)
    sw = make_swarm(25, 4000, 6000)
    eng = E.Engine(group_id_seed=5)
    host.load_swarm(eng, sw)
    eng.tick()
    g1, t1 = engine_groups(eng), _table(eng, sw.W)
    best1, count1 = eng.match_per_task()
    eng.close()
    for r in range(world):
        with open(os.path.join(str(tmp_path), f"rank{r}.pkl"), "rb") as f:
            g, t, s, best, count, n_x = pickle.load(f)
        assert g == g1 and np.array_equal(t, t1), f"rank {r}"
        assert np.array_equal(best, best1) and np.array_equal(count, count1)
        assert n_x == 1   # the published rows: the one exchange of a tick


def test_two_processes_over_torch_distributed_share_the_gpu(tmp_path):
    _two_processes(tmp_path, "gloo")


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="the RCCL leg needs two GPUs (one process per GPU)")
def test_two_processes_over_rccl(tmp_path):
    """the same protocol with backend "nccl" (= RCCL over xGMI): the all-gather of the published rows (and the
    per-task fold's) reads and writes HBM directly, on the stream the engine's kernels run on.  Every rank must end with
    the single-GPU engine's groups, table and per-task bids.  (Skipped on a one-GPU box: the driver's 8-GPU node runs it.)"""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    _two_processes(tmp_path, "nccl")
