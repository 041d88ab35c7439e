"""The streaming carve's rows are built four strides at a time through sorting networks (near_row_offer_bulk,
protocol_amd/csrc/pm_propose.inc) where the batch pipeline inserts candidate by candidate (near_row_offer).  Same keys in,
same row out: the 64 smallest keys in order (checked against numpy's sort here), the same thresholds, and the same answer
of the near-miss tracker whatever site the row's last entry sits at (checked on the device, pm_debug_row_networks).
Reference for what a row is: sort_nodes_by_proximity, crates/orchestrator/src/plugins/node_groups/mod.rs:206-255."""
import numpy as np
import pytest

from protocol_amd import engine as E

pytestmark = pytest.mark.gpu

FLOOR = 0x03B8F2B061AEA073  # bits of 1e-290: below it every key is "near" (near_window)


def _keys(rng, n_waves, n_per_wave, slot_bits, spread, holes):
    """distinct keys: (a value above the window's floor, spread over `spread` steps of the key's truncation) | slot"""
    slots = np.stack([rng.permutation(1 << slot_bits)[:n_per_wave] for _ in range(n_waves)]).astype(np.uint64)
    vals = rng.integers(0, spread, size=(n_waves, n_per_wave), dtype=np.uint64)
    keys = ((np.uint64(FLOOR >> slot_bits) + np.uint64(2) + vals) << np.uint64(slot_bits)) | slots
    if holes:
        keys[rng.random((n_waves, n_per_wave)) < holes] = np.uint64(0xFFFFFFFFFFFFFFFF)
    return keys


@pytest.mark.parametrize("n_per_wave,upto,spread,holes,n_sites", [
    (740, 1024, 1 << 40, 0.0, 4096),   # configs[1]'s rows: three batches, all through the networks
    (740, 256, 1 << 40, 0.1, 4096),    # the first batch through the networks, the rest inserted
    (2048, 1024, 1 << 40, 0.3, 4096),  # a long list: networks, then insertions
    (2048, 2048, 4096, 0.0, 7),        # keys a few steps apart, seven sites: the tracker is busy all the time
    (1024, 1024, 300, 0.5, 3),         # ... and most entries within the window of the threshold
    (50, 1024, 1 << 40, 0.0, 4096),    # fewer candidates than a row holds
    (64, 1024, 100, 0.0, 2),
    (300, 1024, 1 << 20, 0.9, 5),      # mostly holes
])
def test_rows_by_networks_equal_rows_by_insertion(n_per_wave, upto, spread, holes, n_sites):
    rng = np.random.default_rng(n_per_wave * 31 + upto + n_sites)
    slot_bits, n_waves = 13, 96
    keys = _keys(rng, n_waves, n_per_wave, slot_bits, spread, holes)
    sites = rng.integers(0, n_sites, size=1 << slot_bits, dtype=np.uint32)
    eng = E.Engine()
    bits, differing, rows = eng.debug_row_networks(keys, sites, n_per_wave, slot_bits, 1 << 20, upto)
    assert (bits, differing) == (0, 0)
    want = np.sort(keys, axis=1)[:, :64]
    if n_per_wave < 64:
        want = np.concatenate([want, np.full((n_waves, 64 - n_per_wave), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)], axis=1)
    assert np.array_equal(rows, want)
