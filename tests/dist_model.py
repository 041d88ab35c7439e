"""A numpy model of ONE RANK's local compute in the multi-GPU tick — TEST INFRASTRUCTURE (it uses the oracle's
predicate and distance function; the product never imports this).

It implements the same stepwise protocol as libpm_engine.so behind protocol_amd.dist.EngineLocal, so that
protocol_amd.dist.ShardedEngine — the code under test — can run on CPU with real torch.distributed (gloo)
collectives:

  tick_begin       eligible list, first candidate list, the batch's seeds (live located slots below a limit)
  carve_wait       the whole carve, REPLICATED on every rank (the engine's protocol since round 5; nothing is exchanged):
                   per batch the neighbour rows of its seeds, then the sequential chain of try_form_new_groups
                   (mod.rs:505-610) — a step is served from the seed's row (row minus dead entries — candidates only
                   ever leave) when the row still holds enough live entries, else by the exact sort; re-prepares when
                   the seed lies beyond the batch or half of the list is dead
  match_begin      topology filter + claim for the OWNED workers, rows packed into this rank's segment
  tick_end         scatter of the all-gathered segments into the full per-worker table

The row / table encodings are the model's own (int64 entries); what is shared with the engine is the protocol:
who computes what, the segment layout, and that every rank must end with the reference's groups — which only
happens if the rows of the other ranks really arrived.
"""
from __future__ import annotations

import numpy as np
import torch

from oracle import oracle_ffi as orc
from protocol_amd import host

NONE = 0xFFFFFFFF
ROW = 64
F64_MAX = 1.7976931348623157e308


class ModelLocal:
    def __init__(self, sw, *, group_id_seed=1, proximity=True, max_seeds=16384, min_cap=512):
        self.sw = sw
        nodes, cfgs, tasks, enabled = orc.from_swarm(sw)
        self.masks = orc.compat_masks(nodes, cfgs)
        self.elig0 = (sw.status == 2) & sw.has_p2p
        cfg_rows, _alts, _models = host.pack_configs(sw.configs)
        self.order = host.config_order(cfg_rows, sw.enabled_mask())   # carve order (mod.rs:150-164, 399-418)
        self.tm = sw.task_masks()
        self.W, self.T = sw.W, sw.T
        self.group_of = np.full(sw.W, -1, dtype=np.int64)
        self.groups = []                      # [id, cfg, members (carve order), task or -1]
        self.ids = orc.splitmix64_stream(group_id_seed, sw.W + 8)
        self.proximity = proximity
        self.max_seeds, self.min_cap = max_seeds, min_cap
        self.rank, self.world = 0, 1
        self.c_rank, self.c_world = 0, 1      # (who makes which row of a batch: this rank, all of them)
        self.batches = 0

    # ---------------------------------------------------------------- ownership
    def configure(self, rank, world, shard):
        self.rank, self.world = rank, world
        self.shard = np.asarray(shard, dtype=np.int64)
        self.own = np.nonzero(self.shard == rank)[0]
        counts = np.bincount(self.shard, minlength=world)
        self.cap_t = max(int(counts.max()), 1)
        idx_in_shard = np.zeros(self.W, dtype=np.int64)
        for r in range(world):
            m = self.shard == r
            idx_in_shard[m] = np.arange(int(m.sum()))
        self.xrow = self.shard * self.cap_t + idx_in_shard

    # ---------------------------------------------------------------- carve
    def tick_begin(self):
        self.elig = [int(w) for w in np.nonzero(self.elig0 & (self.group_of < 0))[0]]
        self.alive_w = np.zeros(self.W, dtype=bool)
        self.alive_w[self.elig] = True
        self.total_available = len(self.elig)
        self.ci = 0
        self.done = False
        self._prepare()

    def _prepare(self):
        """candidate list of the next configuration whose loop would be entered (mod.rs:505-519)"""
        sw = self.sw
        while self.ci < len(self.order):
            c = int(self.order[self.ci])
            mn, mx = sw.configs[c][1], sw.configs[c][2]
            if self.total_available >= mn:
                lst = [w for w in self.elig if self.alive_w[w] and (int(self.masks[w]) >> c) & 1]
                if len(lst) >= mn and lst:
                    break
            self.ci += 1
        else:
            self.done = True
            return
        self.cfg, self.mn, self.mx = c, mn, mx
        self.lst = np.array(lst, dtype=np.int64)
        n = len(lst)
        self.alive = np.ones(n, dtype=bool)
        self.loc = sw.has_loc[self.lst].copy()
        self.n_cand = n
        self.prop_k = 0
        self.seeds, self.seed_no, self.limit, self.rows_pr = [], {}, 0, 0
        if self.proximity and mx - 1 < ROW - 1:
            self.prop_k = min(mx - 1 + 48, ROW - 1)
            cap = max(self.min_cap, n // 10 if n > 8192 else n // 5)
            cap = min(cap, self.max_seeds)
            located = np.nonzero(self.loc)[0]
            self.limit = n if len(located) <= cap else (int(located[cap - 1]) // 64 + 1) * 64
            self.limit = min(self.limit, n)
            self.seeds = [int(s) for s in located if s < self.limit]
            self.seed_no = {s: i for i, s in enumerate(self.seeds)}
            self.rows_pr = (len(self.seeds) + self.c_world - 1) // self.c_world

    def _sorted_others(self, seed):
        """every live candidate but the seed, by (distance to the seed, slot) — sort_nodes_by_proximity
        (mod.rs:234-255): stable sort, missing location = f64::MAX"""
        sw = self.sw
        idx = np.nonzero(self.alive)[0]
        idx = idx[idx != seed]
        w = self.lst[idx]
        d = np.full(len(idx), F64_MAX)
        has = self.loc[idx]
        ws = int(self.lst[seed])
        d[has] = orc.distance_column(float(sw.lat[ws]), float(sw.lon[ws]), sw.lat[w[has]], sw.lon[w[has]])
        return idx[np.argsort(d, kind="stable")]

    def carve_wait(self):
        while not self.done:   # every batch's rows made and used by this rank
            _more, send, recv = self._carve_next_batch()
            if send is not None:
                recv.copy_(send)
            self.carve_validate()

    def _carve_next_batch(self):
        if self.done:
            return False, None, None
        if not self.prop_k or not self.seeds:
            return True, None, None
        self.batches += 1
        send = torch.full((self.rows_pr * ROW,), -1, dtype=torch.int64)
        for i, s in enumerate(self.seeds):
            if i % self.c_world != self.c_rank:
                continue
            near = self._sorted_others(s)
            row = np.full(ROW, -1, dtype=np.int64)
            k = min(self.prop_k, len(near))
            row[:k] = near[:k]
            row[ROW - 1] = 1 if len(near) <= self.prop_k else 0     # the row holds every live candidate
            j = i // self.c_world
            send[j * ROW:(j + 1) * ROW] = torch.from_numpy(row)
        self.recv = torch.full((self.c_world * self.rows_pr * ROW,), -7, dtype=torch.int64)
        return True, send, self.recv

    def carve_validate(self):
        recv = self.recv.numpy() if self.prop_k and self.seeds else None
        while True:
            if not (self.total_available >= self.mn and self.n_cand >= self.mn and self.n_cand > 0):
                break                                                     # mod.rs:507, :517-519
            live = np.nonzero(self.alive)[0]
            f_any = int(live[0])
            live_loc = live[self.loc[live]]
            want = min(self.mx - 1, self.n_cand - 1)
            if self.proximity and len(live_loc):
                seed = int(live_loc[0])                                   # first WITH a location (mod.rs:526-530)
                if self.prop_k and seed >= self.limit:
                    self._prepare()                                       # beyond the batch: next batch, same config
                    return
                if self.prop_k:
                    i = self.seed_no[seed]
                    r0 = ((i % self.c_world) * self.rows_pr + i // self.c_world) * ROW
                    row = recv[r0:r0 + ROW]
                    assert row[ROW - 1] in (0, 1), "the seed's row was never made"
                    ent = row[:ROW - 1]
                    ent = ent[ent >= 0]
                    alive_ent = ent[self.alive[ent]]
                    if len(alive_ent) >= want or row[ROW - 1] == 1:
                        chosen = alive_ent[:want]
                    else:
                        chosen = self._sorted_others(seed)[:want]        # row exhausted: exact sort
                else:
                    chosen = self._sorted_others(seed)[:want]
            else:
                seed = f_any                                              # first-come (mod.rs:553-561, :238)
                chosen = live[live != seed][:want]
            members = [seed] + [int(x) for x in chosen]
            if len(members) < self.mn:
                break                                                     # mod.rs:564-566
            g = len(self.groups)
            mem_w = [int(self.lst[s]) for s in members]
            self.groups.append([int(self.ids[g]), self.cfg, mem_w, -1])
            for s, w in zip(members, mem_w):
                self.alive[s] = False
                self.alive_w[w] = False
                self.group_of[w] = g
            self.n_cand -= len(members)
            self.total_available -= len(members)
            if self.prop_k and self.n_cand * 2 < len(self.lst) and len(self.lst) > 256:
                self._prepare()                                           # half of the list is dead: re-prepare
                return
        self.ci += 1
        self._prepare()

    # ---------------------------------------------------------------- match (owned workers) + table
    def _row_of(self, w):
        g = int(self.group_of[w])
        if g < 0:
            return [NONE, NONE, 0, 0, NONE, 0]
        gid, cfg, mem, task = self.groups[g]
        if task < 0:                                                      # claim: first applicable task
            hit = np.nonzero((self.tm >> np.uint64(cfg)) & np.uint64(1))[0]
            task = int(hit[0]) if len(hit) else -1
        by_addr = sorted(mem, key=lambda x: int(self.sw.address[x]))      # BTreeSet<String> order
        idx = by_addr.index(w)
        nxt = by_addr[(idx + 1) % len(by_addr)]
        return [NONE if task < 0 else task, g, idx, len(mem), nxt, gid & 0x7FFFFFFFFFFFFFFF]

    def match_begin(self):
        if self.world == 1:
            self.table = np.array([self._row_of(int(w)) for w in range(self.W)], dtype=np.int64).reshape(self.W, 6)
            return None, None
        self.recv_t = torch.full((self.world * self.cap_t * 6,), -1, dtype=torch.int64)
        send = torch.full((self.cap_t * 6,), -1, dtype=torch.int64)
        for k, w in enumerate(self.own):
            send[k * 6:(k + 1) * 6] = torch.tensor(self._row_of(int(w)), dtype=torch.int64)
        return send, self.recv_t

    def tick_end(self):
        if self.world > 1:
            x = self.recv_t.numpy().reshape(self.world * self.cap_t, 6)
            self.table = x[self.xrow]
        for w in range(self.W):                                           # the claim is per group (SETNX)
            g = int(self.group_of[w])
            if g >= 0 and self.table[w, 0] != NONE:
                self.groups[g][3] = int(self.table[w, 0])
        return {"n_groups": len(self.groups)}

    def lookup(self, w):
        return tuple(int(v) for v in self.table[w])

    def groups_sorted_members(self):
        """[(id, config, members in address order, task)] like helpers.oracle_groups"""
        return [(gid, cfg, sorted(mem, key=lambda x: int(self.sw.address[x])), task)
                for gid, cfg, mem, task in self.groups]

    # ---------------------------------------------------------------- per-task orientation (owned workers bid)
    def match_per_task_device(self):
        col = np.where(self.elig0 & (self.group_of < 0), self.masks & np.uint64(self.sw.enabled_mask()), np.uint64(0))
        if self.world > 1:
            col = np.where(self.shard == self.rank, col, np.uint64(0))
        best = np.full(self.T, -1, dtype=np.int32)
        count = np.zeros(self.T, dtype=np.int32)
        for t in range(self.T):
            hit = np.nonzero(col & self.tm[t])[0]
            count[t] = len(hit)
            if len(hit):
                best[t] = hit[0]
        return torch.from_numpy(best), torch.from_numpy(count)
