"""CPU model of the proposer's walk over the spatial index (cell_walk / cell_bound_key in pm_propose.inc), statement by
statement: the enumeration of a ring's runs covers the shell of cells exactly once, the lower bounds are lower bounds,
and a walk that stops by the kernel's rule has seen every candidate below the row's window — so the 64 smallest keys it
holds are the 64 smallest keys of the whole list.  (The GPU tests compare the carves; this pins the geometry.)"""
import numpy as np
import pytest

G_SIZES = (32, 64)


def cell_coord(v, g):
    t = np.maximum((v + 1.0) * (g >> 1), 0.0)
    return np.minimum(t.astype(np.int64), g - 1)


def ring_runs(r, cx, cy, cz, g):
    """the kernel's run q of ring r -> (y, z, x0, x1) or None when it leaves the grid (same arithmetic as cell_walk)"""
    side, n_full, inner = 2 * r, (8 * r if r else 1), (2 * r - 1 if r else 0)
    n_runs = n_full + 2 * inner * inner
    out = []
    for qi in range(n_runs):
        dy = dz = 0
        x0 = x1 = cx
        if r > 0:
            if qi < n_full:
                sd, t = divmod(qi, side)
                dy = (-r + t, r, r - t, -r)[sd]
                dz = (-r, -r + t, r, r - t)[sd]
                x0, x1 = cx - r, cx + r
            else:
                m = (qi - n_full) >> 1
                dy = m % inner - (r - 1)
                dz = m // inner - (r - 1)
                x0 = x1 = cx + r if (qi - n_full) & 1 else cx - r
        y, z = cy + dy, cz + dz
        if not (0 <= y < g and 0 <= z < g and x1 >= 0 and x0 < g):
            continue
        out.append((y, z, max(x0, 0), min(x1, g - 1)))
    return out


@pytest.mark.parametrize("g", G_SIZES)
def test_rings_cover_every_shell_exactly_once(g):
    rng = np.random.default_rng(3)
    for _ in range(6):
        cx, cy, cz = (int(v) for v in rng.integers(0, g, 3))
        for r in range(0, 7):
            seen = {}
            for (y, z, x0, x1) in ring_runs(r, cx, cy, cz, g):
                for x in range(x0, x1 + 1):
                    seen[(x, y, z)] = seen.get((x, y, z), 0) + 1
            want = {(x, y, z)
                    for x in range(max(cx - r, 0), min(cx + r, g - 1) + 1)
                    for y in range(max(cy - r, 0), min(cy + r, g - 1) + 1)
                    for z in range(max(cz - r, 0), min(cz + r, g - 1) + 1)
                    if max(abs(x - cx), abs(y - cy), abs(z - cz)) == r}
            assert set(seen) == want and all(v == 1 for v in seen.values()), (cx, cy, cz, r)


def bound_a(gx, gy, gz):
    gx, gy, gz = (max(v - 1e-9, 0.0) for v in (gx, gy, gz))
    return 0.25 * (gx * gx + gy * gy + gz * gz) * (1.0 - 1e-6)


def unit(lat, lon):
    la, lo = np.radians(lat), np.radians(lon)
    return np.cos(la) * np.cos(lo), np.cos(la) * np.sin(lo), np.sin(la)


def _swarm(rng, n):
    lat = np.round(25.0 + 35.0 * rng.random(n), 4)
    ul = rng.random(n) * 110.0
    lon = np.round(np.where(ul < 60.0, -125.0 + ul, -10.0 + (ul - 60.0)), 4)
    city = rng.integers(0, 32, n)
    clat, clon = np.round(25.0 + 35.0 * rng.random(32), 4), np.round(-125.0 + 60.0 * rng.random(32), 4)
    snap = rng.random(n) < 0.4
    return np.where(snap, clat[city], lat), np.where(snap, clon[city], lon)


@pytest.mark.parametrize("g,n", [(32, 9000), (64, 60000)])
def test_walk_sees_everything_below_the_window(g, n):
    rng = np.random.default_rng(g)
    lat, lon = _swarm(rng, n)
    x, y, z = unit(lat, lon)
    h = 2.0 / g
    cxs, cys, czs = cell_coord(x, g), cell_coord(y, g), cell_coord(z, g)
    lin = (czs * g + cys) * g + cxs
    order = np.argsort(lin, kind="stable")
    start = np.searchsorted(lin[order], np.arange(g * g * g + 1))
    alive = rng.random(n) < 0.6  # the candidates of the batch: part of what the index holds
    WINDOW = 1.0 + 2.0 ** -20   # (far wider than the kernel's 2^25 ulps)
    visited_total = 0
    for s in rng.choice(np.nonzero(alive)[0], 60, replace=False):
        a_all = 0.25 * ((x - x[s]) ** 2 + (y - y[s]) ** 2 + (z - z[s]) ** 2)
        cand = alive.copy()
        cand[s] = False
        brute = np.sort(a_all[cand])[:64]
        row = np.empty(0)
        tau_hi = np.inf
        cx, cy, cz = int(cxs[s]), int(cys[s]), int(czs[s])
        r, seen = 0, 0
        while True:
            if r >= 2 and bound_a((r - 1) * h, 0.0, 0.0) > tau_hi:
                break
            assert r <= 14 and r < g, "the walk of the model swarm never runs out of rings"
            for (yy, zz, x0, x1) in ring_runs(r, cx, cy, cz, g):
                xlo, xhi = x0 * h - 1.0, (x1 + 1) * h - 1.0
                ylo, zlo = yy * h - 1.0, zz * h - 1.0
                gx = max(xlo - x[s], x[s] - xhi, 0.0)
                gy = max(ylo - y[s], y[s] - (ylo + h), 0.0)
                gz = max(zlo - z[s], z[s] - (zlo + h), 0.0)
                row_lin = (zz * g + yy) * g
                ent = order[start[row_lin + x0]:start[row_lin + x1 + 1]]
                lb = bound_a(gx, gy, gz)
                if len(ent):
                    assert a_all[ent].min() >= lb, "cell_bound_key is a lower bound of every key in the run"
                if lb > tau_hi:
                    continue
                ent = ent[cand[ent]]
                seen += len(ent)
                row = np.sort(np.concatenate([row, a_all[ent]]))[:64]
                if len(row) == 64:
                    tau_hi = row[63] * WINDOW + 1e-290
            r += 1
        assert np.array_equal(row, brute), s
        # ... and everything within the window of the last entry was offered (the near-miss tracker's domain)
        visited_total += seen
    assert visited_total < 60 * int(cand.sum()) // 8, "the walk looks at a small part of the list"
