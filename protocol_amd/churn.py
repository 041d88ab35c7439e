"""BASELINE configs[4] as a reproducible stream (one definition for bench.py, tools/make_golden_churn.py and the
parity test): a standing swarm of W0 workers; every tick `n_churn` live workers die (their groups dissolve,
status_update_impl.rs:17-29), `n_churn` brand-new workers are appended behind the table (node_groups/mod.rs:487-497)
and `n_new` tasks arrive in front of the task list (task_store.rs:79: newest first)."""
from __future__ import annotations

import numpy as np

from .swarm import make_swarm


class ChurnStream:
    def __init__(self, seed: int, ticks: int, W0: int = 100_000, n_churn: int = 1000, n_new: int = 10_000,
                 T0: int = 10_000):
        self.W0, self.n_churn, self.n_new, self.ticks = W0, n_churn, n_new, ticks
        self.sw_all = make_swarm(seed + 4, T0, W0 + n_churn * ticks)
        sw = self.sw_all
        self.masks, self.created, self.uid = sw.task_masks(), sw.created_at.copy(), sw.task_uid.copy()
        self.rng = np.random.default_rng(seed)
        self.alive = set(np.nonzero(sw.status[:W0] == 2)[0].tolist())
        self.W = W0
        self.next_uid = 1 << 40
        self.t_max = int(self.created.max())

    def step(self):
        """-> (leave: worker indices that die, idx_new: rows appended, (masks, created_at, uid) of the new tasks)"""
        sw = self.sw_all
        leave = self.rng.choice(np.fromiter(self.alive, dtype=np.int64), size=self.n_churn, replace=False)
        self.alive.difference_update(int(w) for w in leave)
        idx_new = np.arange(self.W, self.W + self.n_churn)
        self.alive.update(int(w) for w in idx_new if sw.status[w] == 2)
        self.W += self.n_churn
        pick = self.rng.integers(0, len(self.masks), self.n_new)
        new_tasks = (self.masks[pick], (self.t_max + 1 + np.arange(self.n_new)[::-1]).astype(self.created.dtype),
                     np.arange(self.next_uid, self.next_uid + self.n_new, dtype=np.uint64), pick)
        self.t_max += self.n_new
        self.next_uid += self.n_new
        return leave, idx_new, new_tasks
