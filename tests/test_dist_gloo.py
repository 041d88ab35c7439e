"""The N>1 path on CPU: world_size-2 gloo, real torch.distributed collectives.

The local compute of each rank is an oracle-backed stand-in (this is a test: the oracle may be used here as
the checker); what is under test is protocol_amd.dist — hash sharding, the all-gather of per-task best bids
with its deterministic fold, and the assembly of the global per-worker task table — against the unsharded
oracle result on the same swarm."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle_ffi as orc
from protocol_amd.dist import NONE, ShardedMatcher, shard_of
from protocol_amd.swarm import Swarm, make_swarm


class OracleLocal:
    """match_per_task() of one shard computed with the oracle's masks (stand-in for libpm_engine.so)."""

    def __init__(self, sw: Swarm, idx: np.ndarray):
        nodes, cfgs, tasks, _ = orc.from_swarm(sw)
        masks = orc.compat_masks(nodes[idx], cfgs)
        elig = (sw.status[idx] == 2) & sw.has_p2p[idx]
        self.col = np.where(elig, masks & np.uint64(sw.enabled_mask()), np.uint64(0))
        self.tm = sw.task_masks()

    def match_per_task(self):
        T = len(self.tm)
        best = np.full(T, NONE, dtype=np.uint32)
        count = np.zeros(T, dtype=np.uint32)
        for t in range(T):
            hit = np.nonzero(self.col & self.tm[t])[0]
            count[t] = len(hit)
            if len(hit):
                best[t] = hit[0]
        return best, count


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, world: int, port: int, seed: int, out_dir: str):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sw = make_swarm(seed, 400, 600)
        shard = shard_of(sw.address, world)
        idx = np.nonzero(shard == rank)[0]
        m = ShardedMatcher(OracleLocal(sw, idx), idx, sw.W, device="cpu")
        best, count = m.match_per_task()
        # a per-worker column that is easy to verify: task id := global index % 7 (PM_NONE for every 5th)
        col = np.where(idx % 5 == 0, NONE, idx % 7).astype(np.uint32)
        table = m.gather_task_table(col)
        np.savez(os.path.join(out_dir, f"rank{rank}.npz"), best=best, count=count, table=table)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_fold_matches_unsharded(world, tmp_path):
    seed = 5
    mp.spawn(_worker, args=(world, _free_port(), seed, str(tmp_path)), nprocs=world, join=True)
    sw = make_swarm(seed, 400, 600)
    want_best, want_count = OracleLocal(sw, np.arange(sw.W)).match_per_task()
    g = np.arange(sw.W)
    want_table = np.where(g % 5 == 0, NONE, g % 7).astype(np.uint32)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        assert np.array_equal(got["best"], want_best), f"rank {r}: folded best bids differ from the unsharded result"
        assert np.array_equal(got["count"], want_count)
        assert np.array_equal(got["table"], want_table)


def test_shard_function_is_the_documented_hash():
    a = np.array([0, 1, 2, 12345678901234567], dtype=np.uint64)
    s = a.copy()
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


def test_single_process_path():
    sw = make_swarm(3, 100, 200)
    idx = np.arange(sw.W)
    m = ShardedMatcher(OracleLocal(sw, idx), idx, sw.W)
    best, count = m.match_per_task()
    wb, wc = OracleLocal(sw, idx).match_per_task()
    assert np.array_equal(best, wb) and np.array_equal(count, wc)
    assert np.array_equal(m.gather_task_table(np.arange(sw.W) % 3), (np.arange(sw.W) % 3).astype(np.uint32))
