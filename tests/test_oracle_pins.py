"""CPU: independent pins of the two scipy-1.2.1 p-value conventions the oracle restates.

oracle/gen_golden.py binds `ks_2samp` / `mannwhitneyu` to the oracle's own restatements when it runs the reference's
glue, so the golden KS / MWU p-values alone would test the oracle against itself.  Here the same conventions are
checked against things that are NOT the oracle:
  * KS  (call site myDetect.py:341 -> scipy 1.2.1 ks_2samp): p = kstwobign.sf((en + 0.12 + 0.11/en) * D) — the literal
    1.2.1 expression evaluated with the container's scipy.stats.kstwobign (unchanged since), and an mpmath evaluation
    of the Kolmogorov series 2 sum (-1)^(k-1) exp(-2 k^2 x^2) at 50 digits;
  * MWU (call site myDetect.py:331 -> scipy 1.2.1 mannwhitneyu, use_continuity=True, alternative=None):
    p = norm.sf(|z|) = erfc(|z| / sqrt 2) / 2 in mpmath, z from exact rational rank sums and tie term; U = min(U1, U2);
    and scipy 1.15.3's two-sided asymptotic p / 2 where the two conventions coincide.
The C restatement (oracle/nanomod_oracle.c) is held to the same pins."""
import fractions
import math

import numpy as np
import pytest
import scipy.stats

import nanomod_oracle as orc

mpmath = pytest.importorskip('mpmath')
mpmath.mp.dps = 50


def _pairs(n_pairs=320, seed=20261002):
    """ragged 3-decimal pairs: null-like, shifted (p down to < 1e-100), near-identical (D -> 0) and tie-heavy ones"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n_pairs):
        n0, n1 = (int(v) for v in rng.integers(5, 400, size=2))
        kind = i % 8
        if kind == 0:                                     # far apart: p < 1e-100 for the larger sizes
            n0 += 300; n1 += 300
            a = np.round(rng.normal(0, 1, n0), 3); b = np.round(rng.normal(6.0, 1, n1), 3)
        elif kind == 1:                                   # D -> 0: b is a with one value moved
            a = np.round(rng.normal(0, 1, n0), 3); b = a.copy(); b[0] += 0.001
        elif kind == 2:                                   # coarse grid: long tie runs inside and across the groups
            a = np.round(rng.normal(0, 1, n0), 1); b = np.round(rng.normal(0.1, 1, n1), 1)
        elif kind == 3:                                   # moderate shift: 1e-30 < p < 1e-3
            a = np.round(rng.normal(0, 1, n0), 3); b = np.round(rng.normal(1.0, 1.2, n1), 3)
        else:
            a = np.round(rng.normal(0, 1, n0), 3); b = np.round(rng.normal(0.2, 1.1, n1), 3)
        out.append((a, b))
    return out


def _kolmogorov_mp(x):
    x = mpmath.mpf(x)
    if x <= 0:
        return mpmath.mpf(1)
    s = mpmath.mpf(0)
    for k in range(1, 200):
        t = mpmath.exp(-2 * k * k * x * x)
        s += t if k % 2 else -t
        if t < mpmath.mpf(10) ** -60:
            break
    return min(mpmath.mpf(1), max(mpmath.mpf(0), 2 * s))


def _mwu_exact(a, b):
    """(min U, z^2 as an exact rational) of scipy 1.2.1's mannwhitneyu from integer counts"""
    F = fractions.Fraction
    ka = np.rint(np.asarray(a) * 1000).astype(np.int64); kb = np.rint(np.asarray(b) * 1000).astype(np.int64)
    n1, n2 = len(ka), len(kb)
    allv = np.concatenate([ka, kb])
    vals, cnt = np.unique(allv, return_counts=True)
    below = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    rank = {int(v): F(int(2 * lo + c + 1), 2) for v, lo, c in zip(vals, below, cnt)}     # average rank of a tie group
    r1 = sum(rank[int(v)] for v in ka)
    u1 = n1 * n2 + F(n1 * (n1 + 1), 2) - r1
    u2 = n1 * n2 - u1
    n = n1 + n2
    T = 1 - F(int(np.sum(cnt.astype(object) ** 3 - cnt.astype(object))), n ** 3 - n)
    var = T * n1 * n2 * (n + 1) / 12
    num = max(u1, u2) - F(n1 * n2, 2) - F(1, 2)
    return min(u1, u2), num, var


def test_ks_p_value_is_the_1_2_1_expression_and_the_kolmogorov_series():
    worst_sf = worst_mp = worst_x = 0.0
    tiny = small_d = 0
    for a, b in _pairs():
        d, p = orc.ks_2samp(a, b)
        n0, n1 = len(a), len(b)
        # D itself: independent exact rational max |c0/n0 - c1/n1| over the pooled points
        pooled = np.concatenate([a, b])
        c0 = np.searchsorted(np.sort(a), pooled, side='right'); c1 = np.searchsorted(np.sort(b), pooled, side='right')
        num = int(np.max(np.abs(c0 * n1 - c1 * n0)))
        assert abs(d - num / (n0 * n1)) <= 2.3e-16
        en = np.sqrt(n0 * n1 / float(n0 + n1))
        lit = float(scipy.stats.kstwobign.sf((en + 0.12 + 0.11 / en) * d))      # scipy 1.2.1 ks_2samp, verbatim
        # the series at the double argument the 1.2.1 expression forms (pins the special function), and at the
        # argument carried in 50 digits (pins the expression; the double argument's rounding, ~1e-16, is amplified
        # by 4 x^2 <~ 3e3 in the far tail)
        ser = _kolmogorov_mp((en + 0.12 + 0.11 / en) * d)
        ser_x = _kolmogorov_mp((mpmath.mpf(en) + mpmath.mpf('0.12') + mpmath.mpf('0.11') / mpmath.mpf(en)) * mpmath.mpf(d))
        if lit > 0:
            worst_sf = max(worst_sf, abs(p - lit) / lit)
        if ser > mpmath.mpf(10) ** -300:
            worst_mp = max(worst_mp, float(abs(mpmath.mpf(p) - ser) / ser))
            worst_x = max(worst_x, float(abs(mpmath.mpf(p) - ser_x) / ser_x))
        else:
            assert p < 1e-299
        tiny += p < 1e-100
        small_d += d < 0.01
    assert tiny >= 10 and small_d >= 10, (tiny, small_d)                         # the pins cover p < 1e-100 and D -> 0
    assert worst_sf == 0.0, worst_sf                                             # the same special function: bit for bit
    assert worst_mp <= 1e-13, worst_mp
    assert worst_x <= 1e-12, worst_x


def test_mwu_p_value_is_the_legacy_one_sided_normal_tail():
    worst_mp = worst_sp = 0.0
    for a, b in _pairs(seed=77):
        u, p = orc.mannwhitneyu(a, b)
        umin, num, var = _mwu_exact(a, b)
        assert u == float(umin)                                                  # statistic = min(U1, U2), exact
        z = mpmath.mpf(num.numerator) / mpmath.mpf(num.denominator) / mpmath.sqrt(mpmath.mpf(var.numerator) / mpmath.mpf(var.denominator))
        pm = mpmath.erfc(abs(z) / mpmath.sqrt(2)) / 2                            # norm.sf(|z|)
        if pm > mpmath.mpf(10) ** -300:
            worst_mp = max(worst_mp, float(abs(mpmath.mpf(p) - pm) / pm))
        r = scipy.stats.mannwhitneyu(a, b, use_continuity=True, alternative='two-sided', method='asymptotic')
        assert u == min(r.statistic, len(a) * len(b) - r.statistic)
        if r.pvalue < 1.0 and r.pvalue > 1e-290:
            worst_sp = max(worst_sp, abs(p - r.pvalue / 2) / p)
    assert worst_mp <= 1e-12, worst_mp          # z carries ~1e-16 relative error, amplified by z^2 ~ 1e3 in the far tail
    assert worst_sp <= 1e-12, worst_sp


def test_c_restatement_meets_the_same_pins():
    oracle_c = pytest.importorskip('oracle_c', reason='make -C oracle')
    pairs = _pairs(n_pairs=160, seed=5)
    sig0 = np.concatenate([a for a, _ in pairs]).astype(np.float32); sig1 = np.concatenate([b for _, b in pairs]).astype(np.float32)
    off0 = np.zeros(len(pairs) + 1, np.int64); off0[1:] = np.cumsum([len(a) for a, _ in pairs])
    off1 = np.zeros(len(pairs) + 1, np.int64); off1[1:] = np.cumsum([len(b) for _, b in pairs])
    out = oracle_c.detect_batch(sig0, off0, sig1, off1, np.zeros(len(pairs), np.int32), 0, 2.0, 'ks', tests=7, threads=1)
    for i in range(len(pairs)):
        a = sig0[off0[i]:off0[i + 1]].astype(np.float64); b = sig1[off1[i]:off1[i + 1]].astype(np.float64)   # what the C side saw
        n0, n1 = len(a), len(b)
        d = out['ks_d'][i]
        en = np.sqrt(n0 * n1 / float(n0 + n1))
        lit = max(float(scipy.stats.kstwobign.sf((en + 0.12 + 0.11 / en) * d)), orc.DBL_MIN)
        assert abs(out['ks_p'][i] - lit) <= 1e-12 * lit, (i, out['ks_p'][i], lit)
        r = scipy.stats.mannwhitneyu(a, b, use_continuity=True, alternative='two-sided', method='asymptotic')
        assert out['mwu_u'][i] == min(r.statistic, n0 * n1 - r.statistic)
        if 1e-290 < r.pvalue < 1.0:
            assert abs(out['mwu_p'][i] - r.pvalue / 2) <= 1e-11 * out['mwu_p'][i], (i, out['mwu_p'][i], r.pvalue / 2)
