#!/usr/bin/env python3
"""A SmallCrush-style battery on the ADDRESSED stream of the counter-mode generators (quality gate before speed).

TestU01 / PractRand are not installable here (no network), so the ten tests of TestU01's SmallCrush are restated with numpy + scipy
(L'Ecuyer & Simard, "TestU01: A C library for empirical testing of random number generators", ACM TOMS 33(4), 2007, section 5 and the
user's guide of bbattery): BirthdaySpacings, Collision, Gap, SimpPoker, CouponCollector, MaxOft, WeightDistrib, MatrixRank,
HammingIndep, RandomWalk1 -- each as a chi-square / Poisson / normal test that yields a p-value -- plus three tests aimed at what a path
tracer does with the numbers (byte frequencies of every byte of every word, serial pairs across ADJACENT ADDRESSES, the unit-ball
acceptance rate of vec3.rs:54-61).

What is tested is not "the generator on a counter" in the abstract but the words the render kernels really draw, in the orders that matter:
  path    one path after the other: (y, x, s) -> events r = 0..R-1 -> blocks j = 0..J-1 -> words 0..3            (what ONE path consumes)
  pixel   fixed (event, block, word), the samples s of a pixel fastest, then x, then y                            (what a pixel's MEAN averages over)
  row     fixed (s, event, block, word), x fastest, then y                                                      (neighbouring pixels: visible correlation)
  seeds   fixed (y, x, s, event, block), consecutive 64-bit seeds                                                 (options.seed)
Generators: philox10 (rounds 1-4), philox7, pcg4d (rt_rng.h, CTR_GEN 0 / 1 / 2), and two deliberately weak ones (pcg4d cut to its first half,
an LCG) that the battery must FAIL -- a battery that passes everything proves nothing.

usage: python tools/rng_battery.py [--gens pcg4d,philox10,...] [--log2n 24] [--out profiles/r05/rng_battery.json]
A p-value outside [1e-4, 1 - 1e-4] is a FAIL, outside [1e-3, 1 - 1e-3] SUSPECT (TestU01's own thresholds are 1e-10 / 1e-4 for its reports)."""
import argparse
import json
import sys
import time

import numpy as np
from scipy import stats

U32 = np.uint32
U64 = np.uint64
M32 = U64(0xFFFFFFFF)


# ---------------------------------------------------------------------------------------------------
# Generators: numpy restatements of rt_rng.h (the oracle's C++ and the device's HIP are checked against these in tests/test_oracle_rng.py)
# ---------------------------------------------------------------------------------------------------
def mad32(a, b, c):
    return ((a.astype(U64) * b.astype(U64) + c.astype(U64)) & M32).astype(U32)


def pcg4d(x, y, z, w, rounds=2):
    k, c = np.full_like(x, 1664525), np.full_like(x, 1013904223)
    x, y, z, w = mad32(x, k, c), mad32(y, k, c), mad32(z, k, c), mad32(w, k, c)
    x = mad32(y, w, x); y = mad32(z, x, y); z = mad32(x, y, z); w = mad32(y, z, w)
    if rounds >= 2:
        x, y, z, w = x ^ (x >> U32(16)), y ^ (y >> U32(16)), z ^ (z >> U32(16)), w ^ (w >> U32(16))
        x = mad32(y, w, x); y = mad32(z, x, y); z = mad32(x, y, z); w = mad32(y, z, w)
    return x, y, z, w


def philox4x32(k0, k1, c0, c1, c2, c3, rounds):
    M0, M1, W0, W1 = U64(0xD2511F53), U64(0xCD9E8D57), U32(0x9E3779B9), U32(0xBB67AE85)
    k0, k1 = k0.copy(), k1.copy()
    for _ in range(rounds):
        p0, p1 = M0 * c0.astype(U64), M1 * c2.astype(U64)
        hi0, lo0, hi1, lo1 = (p0 >> U64(32)).astype(U32), (p0 & M32).astype(U32), (p1 >> U64(32)).astype(U32), (p1 & M32).astype(U32)
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = k0 + W0; k1 = k1 + W1
    return c0, c1, c2, c3


def block(gen, k0, k1, x, s, ray, j):
    """The 4 words of block j of the event after ray `ray` of path (key k0/k1, x, s) -- RngCtr::block of rt_rng.h.  All arguments: uint32 arrays."""
    if gen == "philox10":
        return philox4x32(k0, k1, x, s, ray, j, 10)
    if gen == "philox7":
        return philox4x32(k0, k1, x, s, ray, j, 7)
    if gen in ("pcg4d", "pcg4d_half"):
        b = pcg4d(x, s, k0, k1)                                       # the per-path base (always the full function)
        return pcg4d(b[0], b[1], b[2] + ray, b[3] + j, rounds=2 if gen == "pcg4d" else 1)
    if gen == "lcg":                                                  # a deliberately poor hash of the address: one LCG step per word of a linear index
        i = (((k0 * U32(601) + x) * U32(4099) + s) * U32(61) + ray) * U32(7) + j
        w0 = i * U32(1664525) + U32(1013904223)
        return w0, w0 * U32(1664525) + U32(1013904223), w0 * U32(22695477) + U32(1), w0 * U32(69069) + U32(12345)
    raise ValueError(gen)


def stream(gen, order, n_words, word_sel=None):
    """n_words uint32 values of the addressed stream in one of the four orders.  Image geometry as the headline config: 800 columns, 256 spp."""
    W, SPP, R, J = 800, 256, 4, 2
    n_blocks = (n_words + 3) // 4 if order == "path" else n_words
    i = np.arange(n_blocks, dtype=np.uint64)
    zero = np.zeros(n_blocks, U32)
    if order == "path":                                               # block index -> (path, ray, j); all 4 words of every block
        j = (i % U64(J)).astype(U32); r = ((i // U64(J)) % U64(R)).astype(U32); p = i // U64(J * R)
        s = (p % U64(SPP)).astype(U32); x = ((p // U64(SPP)) % U64(W)).astype(U32); y = (p // U64(SPP * W)).astype(U32)
        out = np.stack(block(gen, y, zero, x, s, r, j), axis=1).reshape(-1)
        return out[:n_words]
    sel = 1 if word_sel is None else word_sel
    if order == "pixel":                                              # one word per path: event 1, block 0
        s = (i % U64(SPP)).astype(U32); x = ((i // U64(SPP)) % U64(W)).astype(U32); y = (i // U64(SPP * W)).astype(U32)
        return block(gen, y, zero, x, s, zero + U32(1), zero)[sel]
    if order == "row":                                                # sample 0 of every pixel
        x = (i % U64(W)).astype(U32); y = (i // U64(W)).astype(U32)
        return block(gen, y, zero, x, zero, zero + U32(1), zero)[sel]
    if order == "seeds":                                              # row key = y + seed (64 bit), y = 300: consecutive seeds, the rest of the address fixed
        key = i + U64(300)
        return block(gen, (key & M32).astype(U32), (key >> U64(32)).astype(U32), zero + U32(400), zero + U32(7), zero + U32(1), zero)[sel]
    raise ValueError(order)


# ---------------------------------------------------------------------------------------------------
# Tests.  Each takes uint32 words and returns a list of (name, p-value).
# ---------------------------------------------------------------------------------------------------
def chi2_p(obs, exp):
    obs, exp = np.asarray(obs, float), np.asarray(exp, float)
    keep = exp > 0
    st = ((obs[keep] - exp[keep]) ** 2 / exp[keep]).sum()
    return float(stats.chi2.sf(st, keep.sum() - 1))


def merge_tail(obs, exp, min_exp=10.0):
    """Merges cells from the right until every expected count is >= min_exp (the usual rule for a chi-square test)."""
    obs, exp = list(map(float, obs)), list(map(float, exp))
    while len(exp) > 2 and exp[-1] < min_exp:
        e, o = exp.pop(), obs.pop()                                   # (pop FIRST: `exp[-2] += exp.pop()` would index the list before it shrinks)
        exp[-1] += e; obs[-1] += o
    return obs, exp


def t_bytes(w):
    out = []
    for b in range(4):
        c = np.bincount(((w >> U32(8 * b)) & U32(255)).astype(np.int64), minlength=256)
        out.append((f"byte{b}_frequency", chi2_p(c, np.full(256, len(w) / 256.0))))
    return out


def t_serial_pairs(w):
    """Adjacent addresses: the top 10 bits of word i against word i+1, non-overlapping: 2^20 cells."""
    a = (w[0::2] >> U32(22)).astype(np.int64); b = (w[1::2] >> U32(22)).astype(np.int64)
    n = min(len(a), len(b))
    c = np.bincount(a[:n] * 1024 + b[:n], minlength=1 << 20)
    return [("serial_pairs_top10", chi2_p(c, np.full(1 << 20, n / float(1 << 20))))]


def t_birthday_spacings(w):
    """smarsa_BirthdaySpacings: n points in d = 2^30 cells (the top 30 bits), lambda = n^3 / 4d; the number of repeated spacings, summed over
    N replications, is Poisson(N * lambda)."""
    n, d, lam = 4096, float(1 << 30), 4096.0 ** 3 / (4.0 * (1 << 30))
    reps = len(w) // n
    pts = np.sort((w[:reps * n] >> U32(2)).astype(np.int64).reshape(reps, n), axis=1)
    sp = np.sort(np.diff(pts, axis=1), axis=1)
    y = int((np.diff(sp, axis=1) == 0).sum())
    mu = reps * lam
    p_right = float(stats.poisson.sf(y - 1, mu)); p_left = float(stats.poisson.cdf(y, mu))
    return [("birthday_spacings_30bit", min(1.0, 2 * min(p_left, p_right)) if min(p_left, p_right) < 0.5 else p_left)]


def t_collision(w):
    """sknuth_Collision: n balls into k = 2^24 urns (the top 24 bits), n = k / 16 per replication: collisions ~ normal (exact mean / variance of
    the occupancy problem), summed over replications."""
    k = 1 << 24; n = k // 16
    reps = len(w) // n
    tot, cols = 0.0, 0
    for r in range(reps):
        u = np.unique(w[r * n:(r + 1) * n] >> U32(8))
        cols += n - len(u)
    # E[empty] = k (1 - 1/k)^n; collisions = n - (k - empty)
    e1 = k * (1.0 - 1.0 / k) ** n
    e2 = k * (k - 1) * (1.0 - 2.0 / k) ** n
    var = e2 + e1 - e1 * e1
    mean = n - k + e1
    z = (cols - reps * mean) / np.sqrt(reps * var)
    return [("collision_24bit", float(2 * stats.norm.sf(abs(z))))]


def t_gap(w):
    """sknuth_Gap: gaps between visits of u in [0, 1/16)."""
    u = (w >> U32(28)) == 0
    idx = np.flatnonzero(u)
    gaps = np.diff(idx) - 1
    p, tmax = 1.0 / 16, 80
    c = np.bincount(np.minimum(gaps, tmax), minlength=tmax + 1)
    exp = np.array([p * (1 - p) ** t for t in range(tmax)] + [(1 - p) ** tmax]) * len(gaps)
    return [("gap_1_16", chi2_p(c, exp))]


def t_poker(w):
    """sknuth_SimpPoker: hands of 8 nibbles... here 5 draws of d = 16 values (top nibble of 5 consecutive words): distinct values per hand."""
    d, k = 16, 5
    n = len(w) // k
    v = (w[:n * k] >> U32(28)).astype(np.int64).reshape(n, k)
    sv = np.sort(v, axis=1)
    distinct = 1 + (np.diff(sv, axis=1) != 0).sum(axis=1)
    c = np.bincount(distinct, minlength=k + 1)[1:]
    # P(r distinct) = d (d-1) ... (d-r+1) S(k, r) / d^k, S = Stirling numbers of the second kind
    S = {1: 1, 2: 15, 3: 25, 4: 10, 5: 1}
    exp = []
    for r in range(1, k + 1):
        f = 1.0
        for t in range(r):
            f *= (d - t)
        exp.append(f * S[r] / d ** k * n)
    return [("simple_poker_16x5", chi2_p(c, exp))]


def t_coupon(w):
    """sknuth_CouponCollector with d = 8 (the top 3 bits): length of the segments needed to see all 8 values."""
    d = 8
    v = (w >> U32(29)).astype(np.int64)
    v = v[:1 << 22]                                                   # a Python loop: bounded
    lens = []
    seen, cnt, start = 0, 0, 0
    full = (1 << d) - 1
    for i, x in enumerate(v.tolist()):
        b = 1 << x
        if not seen & b:
            seen |= b; cnt += 1
            if cnt == d:
                lens.append(i - start + 1); seen = 0; cnt = 0; start = i + 1
    lens = np.array(lens)
    tmax = 60
    c = np.bincount(np.minimum(lens, tmax), minlength=tmax + 1)[d:]
    # P(length = t) = d!/d^t * S(t-1, d-1); computed by the recurrence on "t draws cover exactly j values"
    probs = []
    cover = np.zeros(d + 1); cover[0] = 1.0
    pt_prev_full = 0.0
    for t in range(1, tmax + 1):
        new = np.zeros(d + 1)
        for j in range(d + 1):
            if cover[j] == 0:
                continue
            new[j] += cover[j] * j / d
            if j < d:
                new[j + 1] += cover[j] * (d - j) / d
        cover = new
        probs.append(cover[d] - pt_prev_full); pt_prev_full = cover[d]
    exp = np.array(probs[d - 1:tmax - 1] + [1.0 - sum(probs[:tmax - 1])]) * len(lens)
    o, e = merge_tail(c, exp)
    return [("coupon_collector_8", chi2_p(o, e))]


def t_max_of_t(w):
    """sknuth_MaxOft, t = 8: the maximum of 8 uniforms has CDF x^8, so max^8 is uniform: chi-square over 64 cells + Anderson-Darling-like KS."""
    t = 8
    n = len(w) // t
    u = (w[:n * t].astype(np.float64) + 0.5) / 4294967296.0
    m = u.reshape(n, t).max(axis=1) ** t
    c = np.bincount(np.minimum((m * 64).astype(np.int64), 63), minlength=64)
    sub = m[:1 << 20]
    return [("max_of_8_chi2", chi2_p(c, np.full(64, n / 64.0))), ("max_of_8_ks", float(stats.kstest(sub, "uniform").pvalue))]


def t_weight_distrib(w):
    """svaria_WeightDistrib: among k = 256 consecutive uniforms, how many fall into [0, 1/4): Binomial(256, 1/4)."""
    k = 256
    n = len(w) // k
    hits = ((w[:n * k] >> U32(30)) == 0).reshape(n, k).sum(axis=1)
    lo, hi = 40, 90
    c = np.bincount(np.clip(hits, lo, hi) - lo, minlength=hi - lo + 1)
    pm = stats.binom.pmf(np.arange(lo, hi + 1), k, 0.25)
    pm[0] = stats.binom.cdf(lo, k, 0.25); pm[-1] = stats.binom.sf(hi - 1, k, 0.25)
    o, e = merge_tail(c, pm * n)
    o, e = merge_tail(o[::-1], e[::-1])
    return [("weight_distribution_256", chi2_p(o, e))]


def gf2_rank(rows, nbits):
    """Rank over GF(2) of matrices given as arrays of Python ints... vectorised over MANY matrices: rows is (m, L) uint64 with nbits <= 64 columns."""
    rows = rows.copy()
    m, L = rows.shape
    rank = np.zeros(m, np.int64)
    used = np.zeros((m, L), bool)
    for bit in range(nbits - 1, -1, -1):
        mask = U64(1) << U64(bit)
        has = ((rows & mask) != 0) & ~used
        piv = has.argmax(axis=1)
        any_ = has.any(axis=1)
        pr = rows[np.arange(m), piv]
        elim = ((rows & mask) != 0) & any_[:, None]
        elim[np.arange(m), piv] = False
        rows = np.where(elim, rows ^ pr[:, None], rows)
        used[np.arange(m)[any_], piv[any_]] = True
        rank += any_
    return rank


def t_matrix_rank(w):
    """smarsa_MatrixRank: 32 x 32 binary matrices from 32 consecutive words, and 64 x 64 from pairs: P(rank = L - d)."""
    out = []
    for L, name in ((32, "matrix_rank_32"), (64, "matrix_rank_64")):
        per = L * (L // 32)
        m = min(len(w) // per, 200000 if L == 32 else 60000)
        ww = w[:m * per].astype(U64)
        rows = ww.reshape(m, L) if L == 32 else ((ww[0::2] << U64(32)) | ww[1::2]).reshape(m, L)
        r = gf2_rank(rows, L)
        # probabilities of rank L, L-1, <= L-2 for a large random square binary matrix
        p0 = np.prod([1 - 0.5 ** i for i in range(1, L + 1)])
        p1 = 2 * (1 - 0.5 ** L) * p0 / 1.0
        p1 = p0 * (1 - 0.5 ** L) ** 2 / (1 - 0.5) / (1 - 0.5) * 0.5                    # = 0.5776 for L >= ~20
        p2 = 1 - p0 - p1
        c = [(r == L).sum(), (r == L - 1).sum(), (r <= L - 2).sum()]
        out.append((name, chi2_p(c, np.array([p0, p1, p2]) * m)))
    return out


def t_hamming_indep(w):
    """sstring_HammingIndep: the Hamming weights of consecutive (non-overlapping) word pairs are independent Binomial(32, 1/2)."""
    bits = np.unpackbits(w.view(np.uint8)).reshape(-1, 32).sum(axis=1).astype(np.int64)
    a, b = bits[0::2], bits[1::2]
    n = min(len(a), len(b))
    lo, hi = 8, 24
    ai, bi = np.clip(a[:n], lo, hi) - lo, np.clip(b[:n], lo, hi) - lo
    k = hi - lo + 1
    c = np.bincount(ai * k + bi, minlength=k * k).reshape(k, k)
    pm = stats.binom.pmf(np.arange(lo, hi + 1), 32, 0.5)
    pm[0] = stats.binom.cdf(lo, 32, 0.5); pm[-1] = stats.binom.sf(hi - 1, 32, 0.5)
    exp = np.outer(pm, pm) * n
    corr = float(np.corrcoef(a[:n], b[:n])[0, 1])
    z = corr * np.sqrt(n)
    return [("hamming_indep_chi2", chi2_p(c.reshape(-1), exp.reshape(-1))), ("hamming_weight_correlation", float(2 * stats.norm.sf(abs(z))))]


def t_random_walk(w):
    """swalk_RandomWalk1, L = 128 steps from the bits of 4 consecutive words: the final position and the maximum, chi-square against the exact laws."""
    L = 128
    n = len(w) // 4
    n = min(n, 1 << 21)
    bits = np.unpackbits(w[:n * 4].view(np.uint8)).reshape(n, L).astype(np.int8) * 2 - 1
    pos = np.cumsum(bits, axis=1, dtype=np.int16)
    final, mx = pos[:, -1].astype(np.int64), np.maximum(pos.max(axis=1), 0).astype(np.int64)
    # final position H = 2K - L, K ~ Binomial(L, 1/2)
    kk = (final + L) // 2
    lo, hi = 44, 84
    c = np.bincount(np.clip(kk, lo, hi) - lo, minlength=hi - lo + 1)
    pm = stats.binom.pmf(np.arange(lo, hi + 1), L, 0.5); pm[0] = stats.binom.cdf(lo, L, 0.5); pm[-1] = stats.binom.sf(hi - 1, L, 0.5)
    out = [("random_walk_final", chi2_p(c, pm * n))]
    # maximum M: P(M >= m) = P(S_L >= m) + P(S_L > m)  (reflection principle)
    def sf_S(m):                                                       # P(S_L >= m)
        return stats.binom.sf((m + L + 1) // 2 - 1, L, 0.5)
    mmax = 40
    pge = np.array([sf_S(m) + sf_S(m + 1) for m in range(mmax + 2)]); pge[0] = 1.0
    pmx = pge[:-1] - pge[1:]; pmx[-1] = pge[mmax]
    c2 = np.bincount(np.minimum(mx, mmax), minlength=mmax + 1)
    o, e = merge_tail(c2, pmx * n)
    out.append(("random_walk_maximum", chi2_p(o, e)))
    return out


def t_unit_ball(w):
    """vec3.rs:54-61 with the words as the kernels use them: words 1..3 of a block -> [-1, 1)^3 (23 bits each); accept when |p|^2 < 1.
    P = pi/6 up to the lattice (2^23 points per axis: error far below the test's resolution)."""
    n = len(w) // 4
    b = w[:n * 4].reshape(n, 4)
    f = [(b[:, k] >> U32(9)).astype(np.float64) / (1 << 22) - 1.0 for k in (1, 2, 3)]
    acc = (f[0] * f[0] + f[1] * f[1] + f[2] * f[2]) < 1.0
    p = np.pi / 6
    z = (acc.sum() - n * p) / np.sqrt(n * p * (1 - p))
    # runs of accept / reject across consecutive blocks (tries j, j+1 of one event are consecutive blocks)
    a = acc.astype(np.int64)
    pair = np.bincount(a[0::2][:n // 2] * 2 + a[1::2][:n // 2], minlength=4)
    m = n // 2
    exp = np.array([(1 - p) ** 2, (1 - p) * p, p * (1 - p), p * p]) * m
    return [("unit_ball_acceptance", float(2 * stats.norm.sf(abs(z)))), ("unit_ball_consecutive_tries", chi2_p(pair, exp))]


TESTS = [t_bytes, t_serial_pairs, t_birthday_spacings, t_collision, t_gap, t_poker, t_coupon, t_max_of_t, t_weight_distrib, t_matrix_rank,
         t_hamming_indep, t_random_walk]


TWO_SIDED = ("birthday_spacings_30bit", "collision_24bit", "hamming_weight_correlation", "unit_ball_acceptance")   # p = 2 min(left, right): only small values speak


def verdict(p, name=""):
    e = p if name in TWO_SIDED else min(p, 1.0 - p)                # a chi-square p-value near 1 is "too good to be true" and counts as well
    return "FAIL" if e < 1e-4 else "suspect" if e < 1e-3 else "ok"


def run(gen, log2n, orders):
    res = {}
    for order in orders:
        n = 1 << (log2n if order in ("path", "pixel") else min(log2n, 22))
        t0 = time.time()
        w = stream(gen, order, n)
        r = []
        for t in TESTS + ([t_unit_ball] if order == "path" else []):
            r += t(w)
        res[order] = {"n_words": int(n), "seconds": round(time.time() - t0, 1), "tests": {k: float(v) for k, v in r}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gens", default="philox10,philox7,pcg4d,pcg4d_half,lcg")
    ap.add_argument("--log2n", type=int, default=24, help="log2 of the words per stream order (path / pixel); row / seeds use at most 2^22")
    ap.add_argument("--orders", default="path,pixel,row,seeds")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    doc = {"log2n": args.log2n, "thresholds": {"FAIL": 1e-4, "suspect": 1e-3}, "generators": {}}
    for gen in args.gens.split(","):
        res = run(gen, args.log2n, args.orders.split(","))
        fails = [(o, k, p) for o, d in res.items() for k, p in d["tests"].items() if verdict(p, k) == "FAIL"]
        susp = [(o, k, p) for o, d in res.items() for k, p in d["tests"].items() if verdict(p, k) == "suspect"]
        total = sum(len(d["tests"]) for d in res.values())
        doc["generators"][gen] = {"orders": res, "n_tests": total, "fail": [f"{o}/{k} p={p:.3g}" for o, k, p in fails], "suspect": [f"{o}/{k} p={p:.3g}" for o, k, p in susp]}
        print(f"{gen:12s} {total} p-values: {len(fails)} FAIL, {len(susp)} suspect" + ("".join(f"\n      FAIL    {o:6s} {k:28s} p = {p:.3g}" for o, k, p in fails[:12]))
              + ("".join(f"\n      suspect {o:6s} {k:28s} p = {p:.3g}" for o, k, p in susp[:6])), flush=True)
    if args.out:
        json.dump(doc, open(args.out, "w"), indent=1)
    return doc


if __name__ == "__main__":
    main()
