"""Multi-GPU driver: one process per GPU, `torch.distributed` collectives (backend "nccl" = RCCL over xGMI on
MI355X; "gloo" for the CPU tests and for several ranks sharing one GPU) — SURVEY.md §8(e), DESIGN.md §7.

Every rank holds the WHOLE swarm (100k worker rows are 6 MB) and worker w is owned by
    shard = splitmix64(address) % world                                   (shard_of)
The reference carves from one pool (node_groups/mod.rs:492-503), so the carve domain is NOT split — a rank per
shard carving on its own would form different groups than one orchestrator.  What is split is the parallel work:

  * the carve itself is REPLICATED: a chain of dependent steps (group g + 1's seed depends on what group g took) that
    no number of GPUs shortens; every rank runs it whole, as one streaming launch (libpm_engine.so, DESIGN.md
    section 7), and — the result being the reference's, whatever the timing — ends with the identical groups and
    ids.  Nothing is exchanged for it.  (Until round 5 a batch's neighbour rows were dealt over the ranks and
    all-gathered: 55 host-waited exchanges per tick at 1M x 100k, slower on 8 GPUs than on one.  Gone with ABI 3.)
  * the pair sweep + chooser + claim run for the OWNED workers only; the published rows are ALL-GATHERED once per
    tick and scattered into every rank's full table (any rank can answer any worker's heartbeat);
  * per-task best bids (north_star orientation) are computed over the owned workers and folded across ranks
    (min over the global worker index, sum over the counts) — on the device, no host round trip.

`ShardedEngine.tick()` is the whole protocol; the local compute behind it is anything that implements the
stepwise tick (`EngineLocal` = libpm_engine.so through the C ABI; the CPU tests plug in a numpy model).
Nothing here touches oracle/.
"""
from __future__ import annotations

import contextlib

import numpy as np
import torch
import torch.distributed as dist

from .swarm import mix64

NONE = 0xFFFFFFFF
_NO_BID = 2 ** 62


def shard_of(address: np.ndarray, world: int) -> np.ndarray:
    """owner rank of every worker: splitmix64 finaliser of the address, mod world."""
    return (mix64(np.asarray(address, dtype=np.uint64)) % np.uint64(world)).astype(np.int64)


class _DevMem:
    """torch view of raw device memory (no copy) through __cuda_array_interface__"""

    def __init__(self, ptr: int, n: int, typestr: str = "<i8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 3}


class TorchExchanger:
    """all-gather through torch.distributed.  CUDA tensors on a non-NCCL backend (gloo: CPU tests, several ranks
    on one GPU) are staged through host memory; on nccl (= RCCL) the collective reads and writes HBM directly."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.backend = dist.get_backend(group) if dist.is_initialized() else None

    def all_gather(self, recv: torch.Tensor, send: torch.Tensor):
        if self.world == 1:
            if recv.data_ptr() != send.data_ptr():
                recv.copy_(send)
            return
        if send.is_cuda and self.backend != "nccl":
            send_h = send.cpu()
            recv_h = torch.empty(recv.numel(), dtype=recv.dtype)
            dist.all_gather_into_tensor(recv_h, send_h, group=self.group)
            recv.copy_(recv_h)
        else:
            dist.all_gather_into_tensor(recv, send, group=self.group)


class EngineLocal:
    """The stepwise tick of libpm_engine.so as torch tensors.  The engine is switched to a torch stream so that its
    kernels and the collectives (which torch orders against the current stream) need no host synchronisation."""

    def __init__(self, engine, device: torch.device):
        self.eng = engine
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(self.device)
        engine.set_stream(self.stream.cuda_stream)

    def stream_ctx(self):
        return torch.cuda.stream(self.stream)

    def _tensors(self, x, world):
        n = int(x.bytes_per_rank) // 8
        if n == 0:
            return None, None
        send = torch.as_tensor(_DevMem(int(x.send_ptr), n), device=self.device)
        recv = torch.as_tensor(_DevMem(int(x.recv_ptr), n * world), device=self.device)
        return send, recv

    def configure(self, rank, world, shard):
        self.world = world
        self.eng.dist_configure(rank, world, shard)

    def tick_begin(self):
        self.eng.dist_tick_begin()

    def carve_wait(self):
        self.eng.dist_carve_wait()

    def match_begin(self):
        return self._tensors(self.eng.dist_match_begin(), self.world)

    def tick_end(self):
        return self.eng.dist_tick_end()

    def match_per_task_device(self):
        b, c, n = self.eng.match_per_task_device()
        as_i32 = lambda p: torch.as_tensor(_DevMem(p, n, "<i4"), device=self.device)
        return as_i32(b), as_i32(c)


class ShardedEngine:
    """One rank of the multi-GPU matcher.

    local      the local compute: tick_begin / carve_wait / match_begin / tick_end /
               match_per_task_device (+ configure, stream_ctx) — EngineLocal in production
    address    uint64[W]: the worker addresses (the same table on every rank); ownership = shard_of(address)
    exchanger  object with all_gather(recv, send), world, rank — TorchExchanger by default
    """

    def __init__(self, local, address: np.ndarray, *, exchanger=None):
        self.local = local
        self.x = exchanger or TorchExchanger()
        self.world, self.rank = self.x.world, self.x.rank
        self.shard = shard_of(address, self.world).astype(np.uint8)
        local.configure(self.rank, self.world, self.shard)
        self.exchanges = 0
        # bench: time spent in the collectives (device events around every all-gather, read back after the tick)
        self.time_exchanges = False
        self.exchange_ms = 0.0
        self._ev = []

    def _gather(self, recv, send):
        if self.time_exchanges and send.is_cuda:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.x.all_gather(recv, send)
            b.record()
            self._ev.append((a, b))
        else:
            self.x.all_gather(recv, send)
        self.exchanges += 1

    def _ctx(self):
        f = getattr(self.local, "stream_ctx", None)
        return f() if f else contextlib.nullcontext()

    def tick(self) -> dict:
        """one full-swarm match; every rank ends with the identical groups and the full published table"""
        L = self.local
        with self._ctx():
            L.tick_begin()
            L.carve_wait()                      # the whole carve, replicated: nothing to exchange
            send, recv = L.match_begin()
            if send is not None:                # the published rows of the owned workers
                self._gather(recv, send)
            stats = L.tick_end()
            if self._ev:
                torch.cuda.current_stream().synchronize()
                self.exchange_ms += sum(a.elapsed_time(b) for a, b in self._ev)
                self._ev.clear()
            return stats

    def match_per_task(self):
        """north_star orientation: per task the best bid (smallest global worker index) and the number of bidders
        over ALL ranks.  The fold runs where the local results live (HBM for EngineLocal)."""
        with self._ctx():
            best, count = self.local.match_per_task_device()      # int32 views; PM_NONE reads as -1
            key = best.to(torch.int64)
            key = torch.where(key < 0, torch.full_like(key, _NO_BID), key)
            packed = torch.stack([key, count.to(torch.int64) & 0xFFFFFFFF]).reshape(-1)
            if self.world > 1:
                gathered = torch.empty(self.world * packed.numel(), dtype=torch.int64, device=packed.device)
                self.x.all_gather(gathered, packed)
                g = gathered.view(self.world, 2, -1)
                key, total = g[:, 0, :].amin(dim=0), g[:, 1, :].sum(dim=0)   # identical on every rank
            else:
                key, total = packed.view(2, -1)[0], packed.view(2, -1)[1]
            best_out = torch.where(key == _NO_BID, torch.full_like(key, NONE), key)
        return best_out.cpu().numpy().astype(np.uint32), total.cpu().numpy().astype(np.uint32)
