"""The end of a configuration on sorted rows (stream_small_rows, protocol_amd/csrc/pm_stream.inc), restated in plain
Python and checked against the rule it has to reproduce.

For a wave's worth of located candidates or fewer the streaming carve's validator finishes a configuration like this:
helper waves make, for candidate i, the row of its keys to every OTHER candidate — the Haversine term's bits with the
candidate's INDEX in the low bits (the list is ascending by position, so index order is position order: the reference's
stable sort falls back to input order on a tie) — sorted ascending, and the row's inverse (where in row i candidate c
stands); a step of wave 0 takes the lowest live index as the seed (mod.rs:526-530: the first candidate with a location),
reads its row, keeps the entries that are alive (a ballot), selects the first `want` of them, and kills — through the
inverse — the candidates whose place in the row is among the selected.  Rows are made AHEAD of the steps, against whoever
was a candidate when the phase began; that is sound because candidates only disappear: the sorted remaining list the
reference would build at the seed's turn (mod.rs:234-255 over what is left) is the row minus its dead entries.

This file is that algorithm on Python integers and the statement it has to satisfy — group by group the same members, in
the same order, as filtering + a stable sort by key of what is alive at each step.  It runs on CPU and documents the
invariant; the GPU tests check the kernel itself against the oracle (tests/test_gpu_parity.py::
test_configurations_that_end_or_start_in_registers).  No reference code involved."""
import random
import struct

SB = 13  # low key bits that hold the index (PM_CARVE_SLOT_BITS)
EMPTY = (1 << 64) - 1


def bits(a: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", a))[0]


def pack_key(key_bits: int, idx: int) -> int:
    return ((key_bits >> SB) << SB) | idx


def make_row(a, i, n):
    """candidate i's row: keys to the others, ascending, EMPTY behind them (64 lanes); and its inverse"""
    row = sorted(pack_key(bits(a[i][c]), c) if c != i else EMPTY for c in range(n)) + [EMPTY] * (64 - n)
    inv = [None] * n
    for lane, k in enumerate(row):
        if k != EMPTY:
            inv[k & 63] = lane
    return row, inv


def carve_on_rows(a, n, max_s, min_s, helper_lag):
    """the steps of wave 0; rows are made by `helpers` a few candidates ahead (helper_lag = how stale the live mask a
    helper looks at may be: it passes over candidates that were dead THEN — never a later seed)"""
    alive = (1 << n) - 1
    rows = {}
    history = [alive]
    groups = []
    while True:
        n_loc = bin(alive).count("1")
        if n_loc < min_s or n_loc == 0:
            break
        want = min(max_s - 1, n_loc - 1)
        si = (alive & -alive).bit_length() - 1
        # the helpers: ascending claims against a live mask some steps old
        seen = history[max(0, len(history) - 1 - helper_lag)]
        for c in range(n):
            if c not in rows and (seen >> c) & 1:
                rows[c] = make_row(a, c, n)
        assert si in rows, "the lowest live candidate was alive in every older mask: its row has been made"
        row, inv = rows[si]
        live = [k != EMPTY and (alive >> (k & 63)) & 1 for k in row]
        rank = 0
        selm = 0
        members = [si]
        for lane in range(64):
            if live[lane]:
                if rank < want:
                    selm |= 1 << lane
                    members.append(row[lane] & 63)
                rank += 1
        if len(members) < min_s:
            break
        kill = 1 << si
        for c in range(n):
            if c != si and inv[c] is not None and (selm >> inv[c]) & 1:
                kill |= 1 << c
        assert kill == sum(1 << m for m in members), "the inverse names exactly the selected"
        alive &= ~kill
        history.append(alive)
        groups.append(members)
    return groups


def carve_by_the_rule(a, n, max_s, min_s):
    """the reference's loop on what is alive at each step: seed = first, the others by (key, input order), take max - 1"""
    alive = list(range(n))
    groups = []
    while len(alive) >= min_s and alive:
        seed = alive[0]
        others = sorted(alive[1:], key=lambda c: (bits(a[seed][c]) >> SB, c))  # (stable: ties by input order)
        members = [seed] + others[: max_s - 1]
        if len(members) < min_s:
            break
        groups.append(members)
        alive = [c for c in alive if c not in members]
    return groups


def _distances(rng, n, ties):
    """a symmetric table of Haversine-like terms; `ties`: clusters of candidates at one site (exact ties, zero inside)"""
    site = [rng.randrange(max(1, n // 3)) if ties else c for c in range(n)]
    pos = {}
    for s in set(site):
        pos[s] = (rng.random(), rng.random())
    a = [[0.0] * n for _ in range(n)]
    for i in range(n):
        for j in range(n):
            (x0, y0), (x1, y1) = pos[site[i]], pos[site[j]]
            a[i][j] = ((x0 - x1) ** 2 + (y0 - y1) ** 2) / 8.0
    return a


def test_rows_made_ahead_select_what_the_rule_selects():
    rng = random.Random(5)
    for trial in range(300):
        n = rng.randrange(1, 65)
        max_s = rng.randrange(1, 10)
        min_s = rng.randrange(1, max_s + 1)
        a = _distances(rng, n, ties=trial % 2 == 0)
        want = carve_by_the_rule(a, n, max_s, min_s)
        for lag in (0, 1, 3, 100):
            assert carve_on_rows(a, n, max_s, min_s, lag) == want, (trial, n, max_s, min_s, lag)


def test_index_in_the_low_bits_orders_like_the_position():
    """equal distances (co-located candidates): the packed key's order is the index's, i.e. the input order the
    reference's stable sort falls back to"""
    n = 40
    a = [[0.0] * n for _ in range(n)]
    row, inv = make_row(a, 7, n)
    assert [k & 63 for k in row[: n - 1]] == [c for c in range(n) if c != 7]
    assert all(inv[c] == (c if c < 7 else c - 1) for c in range(n) if c != 7)
