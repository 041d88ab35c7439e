"""The N>1 path on CPU: world_size 2 and 3 over gloo, real torch.distributed collectives.

Under test is protocol_amd.dist.ShardedEngine — the stepwise multi-GPU tick with its two kinds of exchange (the
all-gather of each carve batch's neighbour rows, the all-gather of the owned workers' published rows) and the
device-side fold of the per-task best bids.  The local compute of each rank is tests/dist_model.ModelLocal, a
numpy model of the engine's stepwise protocol (the engine itself needs an MI355X; -m gpu tests run the same
driver over libpm_engine.so).  The bar: every rank ends with exactly the groups, tasks and table of the
UNSHARDED oracle on the same swarm (one pool, node_groups/mod.rs:492-610)."""
import os
import pickle
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_ffi as orc
from protocol_amd.dist import NONE, ShardedEngine, shard_of
from protocol_amd.swarm import make_swarm
from dist_model import ModelLocal
from helpers import oracle_groups, oracle_state_for

SEED, T, W = 5, 400, 900


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_rank(sw):
    model = ModelLocal(sw, group_id_seed=SEED, min_cap=64)   # small batches: several per configuration
    se = ShardedEngine(model, sw.address)
    best0, count0 = se.match_per_task()                         # before the carve: every eligible worker bids
    se.tick()
    best1, count1 = se.match_per_task()                         # after: only the leftovers bid (mod.rs:492-497)
    return dict(groups=model.groups_sorted_members(), table=model.table.copy(), best0=best0, count0=count0,
                best1=best1, count1=count1, exchanges=se.exchanges, batches=model.batches,
                own=len(model.own) if se.world > 1 else sw.W)


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = _run_rank(make_swarm(SEED, T, W))
        with open(os.path.join(out_dir, f"rank{rank}.pkl"), "wb") as f:
            pickle.dump(res, f)
    finally:
        dist.destroy_process_group()


def _oracle_reference(sw):
    st = oracle_state_for(sw, reference_shaped=True, group_id_seed=SEED)
    nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
    masks = orc.compat_masks(nodes, cfgs)
    elig = (sw.status == 2) & sw.has_p2p
    tm = sw.task_masks()

    def bids(col):
        best = np.full(sw.T, NONE, dtype=np.uint32)
        count = np.zeros(sw.T, dtype=np.uint32)
        for t in range(sw.T):
            hit = np.nonzero(col & tm[t])[0]
            count[t] = len(hit)
            if len(hit):
                best[t] = hit[0]
        return best, count

    col0 = np.where(elig, masks & np.uint64(sw.enabled_mask()), np.uint64(0))
    b0 = bids(col0)
    st.try_form_new_groups()
    assert st.try_merge_solo_groups() == 0          # the model has no merge pass: the swarm must not need one
    b1 = bids(np.where(st.node_to_group < 0, col0, np.uint64(0)))
    rows = [st.filter_tasks(w) for w in range(sw.W)]   # (task, GROUP_INDEX, GROUP_SIZE, NEXT) incl. the claim
    return st, b0, b1, rows


def _check_rank(res, sw, st, b0, b1, rows, world):
    assert res["groups"] == [(gid, cfg, mem, task) for (gid, cfg, mem, task) in oracle_groups(st)], \
        "groups differ from the unsharded oracle"
    for w in range(sw.W):
        t, gi, gs, nxt = rows[w]
        task, slot, idx, size, nx, _gid = (int(v) for v in res["table"][w])
        if t < 0 and st.node_to_group[w] < 0:
            assert task == NONE and slot == NONE, w
        else:
            assert (task, idx, size, nx) == (NONE if t < 0 else t, gi, gs, nxt), w
    assert np.array_equal(res["best0"], b0[0]) and np.array_equal(res["count0"], b0[1])
    assert np.array_equal(res["best1"], b1[0]) and np.array_equal(res["count1"], b1[1])
    if world > 1:
        assert res["batches"] > 3 and res["exchanges"] == 1   # the ONE exchange of a tick: the published rows (the carve is replicated)
        assert 0 < res["own"] < sw.W


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_tick_equals_the_unsharded_oracle(world, tmp_path):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sw = make_swarm(SEED, T, W)
    st, b0, b1, rows = _oracle_reference(sw)
    for r in range(world):
        with open(os.path.join(str(tmp_path), f"rank{r}.pkl"), "rb") as f:
            res = pickle.load(f)
        _check_rank(res, sw, st, b0, b1, rows, world)


def test_single_process_path():
    sw = make_swarm(SEED, T, W)
    st, b0, b1, rows = _oracle_reference(sw)
    _check_rank(_run_rank(sw), sw, st, b0, b1, rows, 1)


def test_shard_function_is_the_documented_hash():
    a = np.array([0, 1, 2, 12345678901234567], dtype=np.uint64)
    out = []
    for v in a:   # splitmix64 finaliser, written out
        z = (int(v) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        out.append((z ^ (z >> 31)) % 8)
    assert shard_of(a, 8).tolist() == out
    # shards partition the workers and keep them roughly balanced
    sw = make_swarm(1, 10, 20000)
    sh = shard_of(sw.address, 8)
    counts = np.bincount(sh, minlength=8)
    assert counts.sum() == sw.W and counts.min() > 0.9 * sw.W / 8
