"""Hash-sharded swarm across ranks (one process per GPU) — SURVEY.md §8(e).

Workers are partitioned by  shard = splitmix64(address) % world ; configurations and tasks are replicated.
Each shard is an independent carve domain (group formation never crosses shards — the same result as one
reference orchestrator per shard), so the only data-path exchange is the one the north_star names: an
all-gather of the per-task best bids followed by an identical deterministic fold on every rank, plus an
all-gather of the published per-worker task columns when a rank must answer lookups for every worker.

Collectives go through torch.distributed: backend "nccl" is RCCL over xGMI on MI355X; the same code runs
on "gloo" for the CPU tests.  The local compute is whatever object implements `match_per_task()` /
`task_column()` — libpm_engine.so in production.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist

from .swarm import mix64

NONE = 0xFFFFFFFF
_NO_BID = np.int64(2 ** 62)


def shard_of(address: np.ndarray, world: int) -> np.ndarray:
    """shard index of every worker: splitmix64 finaliser of the address, mod world."""
    return (mix64(np.asarray(address, dtype=np.uint64)) % np.uint64(world)).astype(np.int64)


class ShardedMatcher:
    """Cross-shard fold of per-task best bids and assembly of the global assignment table.

    local        object with match_per_task() -> (best_local_worker u32[T], count u32[T])
    global_index int64[W_local]: global worker index (position in the unsharded get_nodes order) of every
                 local worker, ascending — so "first local hit" is the smallest global index of the shard
    """

    def __init__(self, local, global_index: np.ndarray, n_workers_global: int, *, device="cpu", group=None):
        self.local = local
        self.global_index = np.ascontiguousarray(global_index, dtype=np.int64)
        assert np.all(np.diff(self.global_index) > 0), "shards keep the global worker order"
        self.n_global = int(n_workers_global)
        self.device = torch.device(device)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    # ---- north_star orientation: per task, the best bid over ALL shards
    def match_per_task(self):
        best_local, count_local = self.local.match_per_task()
        best_local = np.asarray(best_local, dtype=np.uint32)
        has = best_local != NONE
        # bid key = global worker index (price is the all-zero extension column in every parity run, so
        # (price, index) order == index order); _NO_BID marks "no candidate in this shard"
        key = np.full(best_local.shape, _NO_BID, dtype=np.int64)
        key[has] = self.global_index[best_local[has].astype(np.int64)]
        packed = np.stack([key, np.asarray(count_local, dtype=np.int64)], axis=0)  # (2, T)
        if self.world == 1:
            folded_key, total = packed[0], packed[1]
        else:
            mine = torch.from_numpy(packed).to(self.device).reshape(-1)
            flat = torch.empty(self.world * mine.numel(), dtype=mine.dtype, device=self.device)
            dist.all_gather_into_tensor(flat, mine, group=self.group)  # flat layout: same call on nccl and gloo
            gathered = flat.view(self.world, 2, -1)
            # identical deterministic fold on every rank: min over the bid keys, sum over the counts
            folded_key = gathered[:, 0, :].amin(dim=0).cpu().numpy()
            total = gathered[:, 1, :].sum(dim=0).cpu().numpy()
        best = np.where(folded_key == _NO_BID, NONE, folded_key).astype(np.uint32)
        return best, total.astype(np.uint32)

    # ---- reference orientation: every rank ends up with the whole per-worker task table
    def gather_task_table(self, task_of_local_worker) -> np.ndarray:
        col = np.asarray(task_of_local_worker, dtype=np.int64)
        if self.world == 1:
            out = np.full(self.n_global, NONE, dtype=np.int64)
            out[self.global_index] = col
            return out.astype(np.uint32)
        n_local = torch.tensor([len(col)], dtype=torch.int64, device=self.device)
        sizes = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        dist.all_gather(sizes, n_local, group=self.group)
        n_max = int(max(int(s.item()) for s in sizes))
        pad = torch.full((2, n_max), -1, dtype=torch.int64, device=self.device)
        pad[0, :len(col)] = torch.from_numpy(self.global_index).to(self.device)
        pad[1, :len(col)] = torch.from_numpy(col).to(self.device)
        flat = torch.empty(self.world * 2 * n_max, dtype=torch.int64, device=self.device)
        dist.all_gather_into_tensor(flat, pad.reshape(-1), group=self.group)
        g = flat.view(self.world, 2, n_max).cpu().numpy()
        out = np.full(self.n_global, NONE, dtype=np.int64)
        for r in range(self.world):
            ok = g[r, 0] >= 0
            out[g[r, 0][ok]] = g[r, 1][ok]
        return out.astype(np.uint32)
