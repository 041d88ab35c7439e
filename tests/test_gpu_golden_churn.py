"""The workloads the bench line times beyond BASELINE configs[1] / [2]'s reference orientation, at their real sizes,
against committed oracle digests (tests/golden/churn_digests.json, tools/make_golden_churn.py):

  * BASELINE configs[4] — the churn stream of bench.py's `churn` sub-object (protocol_amd/churn.py): the cold match on
    100k workers, then all eight ticks (the two warm-up ticks and the six bench.py times) of 1000 deaths + 1000 appended workers + 10k tasks in front of the list; groups,
    every worker's task and the group life-cycle feed after every tick, on one engine and on two in-process ranks;
  * pm_match_per_task (north_star orientation) at BASELINE configs[1] and [2]: every task's best bid and bidder count.
"""
import json
import os
import sys
import threading

import numpy as np
import pytest
import torch

from protocol_amd import engine as E
from protocol_amd import host
from protocol_amd.churn import ChurnStream
from protocol_amd.dist import EngineLocal, ShardedEngine
from protocol_amd.swarm import baseline_config
from helpers import engine_groups

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from make_golden_churn import (CHURN_SEED, CHURN_TICKS_PINNED, CHURN_TICKS_PLANNED, events_digest,  # noqa: E402
                               groups_digest, sha)

pytestmark = pytest.mark.gpu
NONE = 0xFFFFFFFF
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "churn_digests.json")))


@pytest.mark.parametrize("name", ["cfg1_seed1", "cfg2_seed1"])
def test_per_task_orientation_at_full_size(name):
    gold = GOLD["per_task"][name]
    sw = baseline_config(gold["config"], seed=gold["seed"])
    assert (sw.W, sw.T) == (gold["W"], gold["T"])
    eng = E.Engine()
    host.load_swarm(eng, sw)
    best, count = eng.match_per_task()
    assert int((count == 0).sum()) == gold["n_without_candidate"]
    assert sha(count.astype(np.uint32)) == gold["count_sha256"], "bidder counts differ from the oracle's masks"
    assert sha(best.astype(np.uint32)) == gold["best_sha256"], "best bids differ from the oracle's masks"
    eng.close()


def _task_column(eng, W):
    return np.array([eng.lookup(w).task for w in range(W)], dtype=np.uint32)


def _check_tick(eng, W, gold, stats, tag):
    assert W == gold["W"], tag
    assert stats["n_formed"] == gold["n_formed"] and stats["n_merged"] == gold["n_merged"], (tag, stats)
    groups = [(g[0], g[1], g[2]) for g in engine_groups(eng)]
    assert len(groups) == gold["n_groups"], tag
    assert groups_digest(groups) == gold["groups_sha256"], f"{tag}: groups differ from the oracle"
    assert sha(_task_column(eng, W)) == gold["task_sha256"], f"{tag}: per-worker tasks differ from the oracle"
    ev = eng.drain_group_events()
    assert len(ev) == gold["n_events"] and events_digest(ev) == gold["events_sha256"], f"{tag}: life-cycle feed differs"
    assert stats["host_resolved_steps"] == 0


def _replay(tick_fn, eng, reconfigure=None):
    """the stream of bench.py's churn sub-object through `eng`; tick_fn() runs one management tick"""
    gold = GOLD["churn"]
    assert (gold["seed"], gold["ticks_planned"]) == (CHURN_SEED, CHURN_TICKS_PLANNED)
    cs = ChurnStream(CHURN_SEED, CHURN_TICKS_PLANNED)
    sw_all = cs.sw_all
    packed = host.pack_workers(sw_all)
    rows = lambda idx: {k: np.ascontiguousarray(v[idx]) for k, v in packed.items()}
    cfg_rows, alt_rows, req_models = host.pack_configs(sw_all.configs)
    eng.set_configs(cfg_rows, alt_rows)
    eng.set_model_table(host.build_model_table(req_models, sw_all.model_names), len(req_models), len(sw_all.model_names))
    eng.upload_workers(rows(np.arange(cs.W0)))
    eng.upload_tasks(cs.masks, cs.created, cs.uid)
    eng.set_enabled_mask(sw_all.enabled_mask())
    eng.enable_group_events()
    if reconfigure:
        reconfigure(sw_all.address[:cs.W0])
    flags = packed["flags"].astype(np.int64)
    _check_tick(eng, cs.W0, gold["cold"], tick_fn(), "cold")
    for k in range(CHURN_TICKS_PINNED):
        leave, idx_new, new_tasks = cs.step()
        eng.on_worker_status_many(leave, flags[leave] & ~E.W_HEALTHY, np.ones(len(leave), dtype=np.uint32))
        eng.append_workers(rows(idx_new))
        eng.tasks_insert_front(*new_tasks[:3])
        if reconfigure:
            reconfigure(sw_all.address[:cs.W])
        _check_tick(eng, cs.W, gold["ticks"][k], tick_fn(), f"tick {k}")


def test_churn_stream_against_oracle_digests():
    eng = E.Engine(group_id_seed=1)
    _replay(eng.tick, eng)
    eng.close()


def test_churn_stream_with_delta_sync_of_the_group_state():
    """the same stream as a deployment runs it: nobody asks for the group list between the ticks (pm_get_groups compacts it,
    and the tick behind a compaction uploads it whole), so a tick's 1,000 dissolved groups and 1,000 new rows reach the
    device as a delta — their workers' group_of, a fill for the new rows, the changed flags — and the dissolved groups
    stay in the list as tombstones until a quarter of it is dead.  Every tick: each worker's task and the life-cycle
    feed against the oracle's digests; after the last one the groups themselves."""
    gold = GOLD["churn"]
    eng = E.Engine(group_id_seed=1)
    cs = ChurnStream(CHURN_SEED, CHURN_TICKS_PLANNED)
    sw_all = cs.sw_all
    packed = host.pack_workers(sw_all)
    rows = lambda idx: {k: np.ascontiguousarray(v[idx]) for k, v in packed.items()}
    cfg_rows, alt_rows, req_models = host.pack_configs(sw_all.configs)
    eng.set_configs(cfg_rows, alt_rows)
    eng.set_model_table(host.build_model_table(req_models, sw_all.model_names), len(req_models), len(sw_all.model_names))
    eng.upload_workers(rows(np.arange(cs.W0)))
    eng.upload_tasks(cs.masks, cs.created, cs.uid)
    eng.set_enabled_mask(sw_all.enabled_mask())
    eng.enable_group_events()
    flags = packed["flags"].astype(np.int64)

    def check(W, g, stats, tag):
        assert stats["n_formed"] == g["n_formed"] and stats["n_groups"] == g["n_groups"], (tag, stats)
        assert sha(_task_column(eng, W)) == g["task_sha256"], f"{tag}: per-worker tasks differ from the oracle"
        ev = eng.drain_group_events()
        assert len(ev) == g["n_events"] and events_digest(ev) == g["events_sha256"], f"{tag}: life-cycle feed differs"

    check(cs.W0, gold["cold"], eng.tick(), "cold")
    for k in range(CHURN_TICKS_PINNED):
        leave, idx_new, new_tasks = cs.step()
        eng.on_worker_status_many(leave, flags[leave] & ~E.W_HEALTHY, np.ones(len(leave), dtype=np.uint32))
        eng.append_workers(rows(idx_new))
        eng.tasks_insert_front(*new_tasks[:3])
        check(cs.W, gold["ticks"][k], eng.tick(), f"tick {k}")
    # (every tick behind the cold match but the one or two at which a quarter of the list was dead and it was compacted)
    assert CHURN_TICKS_PINNED - 2 <= eng.debug_delta_pushes() <= CHURN_TICKS_PINNED
    last = gold["ticks"][CHURN_TICKS_PINNED - 1]
    groups = [(g[0], g[1], g[2]) for g in engine_groups(eng)]
    assert len(groups) == last["n_groups"] and groups_digest(groups) == last["groups_sha256"]
    eng.close()


def test_churn_stream_with_aborted_streaming_launches():
    """every tick's streaming launch is made to give up (CARVE_STATE_ABORTED, include/pm_engine_debug.h) — the cold match after
    2,000 committed steps, the churn ticks after 25 — and the carve continues on the batch pipeline from the configuration
    it stopped in: groups, tasks and the life-cycle feed are the oracle's digests tick for tick"""
    eng = E.Engine(group_id_seed=1)
    aborts = []

    def tick():
        eng.debug_stream_abort_after(2000 if not aborts else 25)
        stats = eng.tick()
        aborts.append(eng.debug_carve_counters()["stream_aborts"])
        return stats

    _replay(tick, eng)
    assert aborts == [1] * (1 + CHURN_TICKS_PINNED), aborts
    eng.close()


def test_churn_stream_on_two_in_process_ranks():
    """the same stream through the stepwise multi-GPU tick: two engines sharing the device, the exchanges as device
    copies (tests/test_gpu_dist.py) — every rank must reproduce the oracle's digests"""
    from test_gpu_dist import _InProcExchanger, _InProcGroup
    world = 2
    grp = _InProcGroup(world)
    errs = []

    def rank_main(r):
        try:
            eng = E.Engine(group_id_seed=1)
            local = EngineLocal(eng, torch.device("cuda", 0))
            state = {}

            def reconfigure(address):
                if "se" not in state:
                    state["se"] = ShardedEngine(local, address, exchanger=_InProcExchanger(grp, r))
                else:
                    from protocol_amd.dist import shard_of
                    local.configure(r, world, shard_of(address, world).astype(np.uint8))

            _replay(lambda: state["se"].tick(), eng, reconfigure)
            eng.close()
        except BaseException as ex:  # a dead rank must not leave the other at the barrier
            errs.append((r, repr(ex)))
            grp.barrier.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
