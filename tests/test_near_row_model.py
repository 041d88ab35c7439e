"""The proposer's selection and its tail certificate, restated in plain Python and checked against brute force.

`carve_propose_kernel` (protocol_amd/csrc/pm_propose.inc: NearRow, near_window, near_track, near_row_offer) keeps a
wave's 64 nearest keys sorted across its lanes, lets a candidate in only if it beats lane 63, and summarises the
candidates that came close without getting (or staying) in — smallest key + its site, smallest key at another site —
so that `tail_ok` ("every unlisted candidate within the band of the row's last entry sits at that entry's site") needs
no second sweep.  This file is the same algorithm on Python integers, stride by stride as the kernel sees them, and
the statement it has to satisfy; it runs on CPU and documents the invariant (the GPU tests check the kernel itself
against the oracle).  No reference code involved: the property is about the engine's own certificate."""
import struct

import numpy as np

SB, BAND, ULPS = 13, 2.0 ** -36, 1 << 20          # small-list geometry (pm_device.h)
SB_BIG, BAND_BIG, ULPS_BIG = 18, 2.0 ** -31, 1 << 25
NOLOC = 0x7FEFFFFFFFFFFFFF
EMPTY = (1 << 64) - 1


def bits(a: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", a))[0]


def as_double(b: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", b))[0]


def pack_key(key_bits: int, slot: int, sb: int) -> int:
    return ((key_bits >> sb) << sb) | slot


def near_window(tau: int, sb: int, ulps: int) -> int:
    if tau >= ((NOLOC >> sb) << sb):
        return tau
    hi = tau + ulps
    floor_bits = 0x03B8F2B061AEA073            # 1e-290, rounded up
    return max(hi, floor_bits) | ((1 << sb) - 1)


class NearRow:
    def __init__(self, sb, ulps):
        self.sb, self.ulps = sb, ulps
        self.key = [EMPTY] * 64                # ascending over the lanes
        self.tau = self.tau_hi = EMPTY
        self.m1 = self.m2 = EMPTY
        self.s1 = 0xFFFFFFFF

    def track(self, cands):
        for k, site in cands:                  # lane order; the kernel's change filter only skips no-ops
            if k < self.m1:
                if site != self.s1:
                    self.m2 = self.m1
                self.m1, self.s1 = k, site
            elif site != self.s1 and k < self.m2:
                self.m2 = k

    def offer(self, keys, sites, site_of_slot):
        """one 64-slot stride: keys[l] = EMPTY where lane l has no candidate"""
        if not any(k < self.tau_hi for k in keys):
            return
        acc = [k for k in keys if k < self.tau]            # judged against the threshold as of the stride's start
        if acc:
            old = list(self.key)
            for kk in acc:                                 # serial insertion, lane order
                row = self.key
                prev = [0] + row[:-1]
                self.key = [(prev[i] if prev[i] > kk else kk) if row[i] > kk else row[i] for i in range(64)]
            self.tau = self.key[63]
            self.tau_hi = near_window(self.tau, self.sb, self.ulps)
            ev = [(o, site_of_slot(o & ((1 << self.sb) - 1))) for o in old if self.tau < o < self.tau_hi]
            self.track(ev)
        self.track([(k, s) for k, s in zip(keys, sites) if self.tau < k < self.tau_hi])


def run_seed(a, sites, located, K, sb, band, ulps):
    """a[s], sites[s], located[s] per slot (slot order = input order); returns the row and the flags"""
    n = len(a)
    row = NearRow(sb, ulps)
    keys_all = [pack_key(bits(a[s]) if located[s] else NOLOC, s, sb) for s in range(n)]
    for base in range(0, n, 64):
        ks = keys_all[base:base + 64] + [EMPTY] * max(0, base + 64 - n)
        ss = list(sites[base:base + 64]) + [0] * max(0, base + 64 - n)
        row.offer(ks, ss, lambda slot: sites[slot])
    n_tot = sum(1 for k in row.key if k != EMPTY)
    n_k = min(n_tot, K)
    noloc_kb = (NOLOC >> sb) << sb
    kb = lambda k: (k >> sb) << sb
    out = {"row": row.key[:n_k], "tail_clear": False, "tail_ok": False, "defined": False}
    if n_k == K:
        e_last = row.key[K - 1]
        beyond = row.key[K] if n_tot > K else EMPTY
        if beyond == EMPTY or kb(e_last) == noloc_kb or kb(beyond) == noloc_kb:
            out["tail_clear"] = True
        else:
            a_last, a_b = as_double(kb(e_last)), as_double(kb(beyond))
            if a_b - a_last > a_b * (4.0 * band) + 1e-300:
                out["tail_clear"] = True
            else:
                out["defined"] = True
                site_last = sites[e_last & ((1 << sb) - 1)]
                band2 = a_last * (4.0 * band) + 1e-300
                bad = False
                for lane in range(K, n_tot):
                    k = row.key[lane]
                    if kb(k) != noloc_kb and as_double(kb(k)) - a_last <= band2 and sites[k & ((1 << sb) - 1)] != site_last:
                        bad = True
                other = row.m1 if row.s1 != site_last else row.m2
                if other != EMPTY and kb(other) != noloc_kb and as_double(kb(other)) - a_last <= band2:
                    bad = True
                out["tail_ok"] = not bad
                # the statement: no unlisted candidate within the band of the last entry sits at another site
                want = True
                for k in sorted(keys_all)[K:]:
                    if kb(k) != noloc_kb and as_double(kb(k)) - a_last <= band2 and sites[k & ((1 << sb) - 1)] != site_last:
                        want = False
                out["tail_ok_brute"] = want
    out["row_brute"] = sorted(keys_all)[:n_k]
    return out


def _world(rng, n, n_sites, tie_pairs, frac_noloc):
    """slots at a few hundred sites, most distances distinct, some site pairs tied to within the key truncation"""
    site_a = rng.uniform(1e-6, 0.3, n_sites)
    for _ in range(tie_pairs):                         # near-mirror sites: same `a` up to a few 1e-14 relative
        i, j = rng.integers(0, n_sites, 2)
        site_a[j] = site_a[i] * (1.0 + rng.integers(-3, 4) * 2e-14)
    big = rng.integers(0, n_sites, max(1, n_sites // 20))          # a few crowded sites (cities)
    p = np.ones(n_sites)
    p[big] = 60.0
    sites = rng.choice(n_sites, size=n, p=p / p.sum())
    located = rng.uniform(size=n) >= frac_noloc
    return site_a[sites], sites.astype(np.int64), located


def _cases():
    for geom in ("small", "big"):
        sb, band, ulps = (SB, BAND, ULPS) if geom == "small" else (SB_BIG, BAND_BIG, ULPS_BIG)
        for seed in range(12):
            rng = np.random.default_rng(1000 + seed)
            n = int(rng.integers(70, 900))
            a, sites, located = _world(rng, n, n_sites=int(rng.integers(3, 60)), tie_pairs=int(rng.integers(0, 12)),
                                       frac_noloc=float(rng.choice([0.0, 0.05, 0.5])))
            for K in (1, 7, 33, 55, 63):
                yield run_seed(a, sites, located, K, sb, band, ulps)


def test_sorted_lanes_and_tracker_equal_brute_force():
    decided = {True: 0, False: 0}
    for out in _cases():
        assert out["row"] == out["row_brute"]                      # the register IS the sorted row
        if out["defined"]:
            assert out["tail_ok"] == out["tail_ok_brute"]
            decided[out["tail_ok"]] += 1
    assert decided[True] >= 20 and decided[False] >= 20            # both outcomes of the certificate were exercised


def test_lone_node_of_another_site_behind_a_crowd():
    """a crowd beyond the row at the last entry's site and one node behind it whose key ties with the crowd's: at
    another site it must break the certificate, at the same site it must not"""
    for lone_site, expect_ok in ((1, False), (0, True)):
        n = 300
        a = np.full(n, 0.01)
        sites = np.zeros(n, dtype=np.int64)                        # slots 0 .. 298: one site, all tied
        a[n - 1] = 0.01 * (1.0 + 2e-14)                            # the last slot: same key after truncation
        sites[n - 1] = lone_site
        out = run_seed(a, sites, np.ones(n, dtype=bool), K=55, sb=SB, band=BAND, ulps=ULPS)
        assert out["defined"] and out["row"] == out["row_brute"]
        assert out["tail_ok"] == out["tail_ok_brute"] == expect_ok
