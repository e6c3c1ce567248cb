"""CPU oracle for NanoMod's per-base two-sample testing hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`nanomod_amd/`) may
import this module; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` use it, and only as the checker.

What is restated
----------------
The reference hot path is `bin/scripts/myDetect.py:317-414` (glue) calling four
functions of **scipy 1.2.1** (pinned in `env.py27nanomod.yml:99`; the scipy
source is NOT vendored under /root/reference and the container ships scipy
1.15.3 whose `ks_2samp` / `mannwhitneyu` defaults differ).  This module restates

* the glue: `m_min_float`/`m_max_float` (myDetect.py:317-325), `getKStest`
  default branch (myDetect.py:327-343,363), `pos_check` (myDetect.py:366-371),
  `get_combin_pvalue` (myDetect.py:379-414), the `mtest2` position walk
  (myDetect.py:416-445) and the `save_test` line format (myDetect.py:522-538);
* the scipy 1.2.1 algorithms (published source, tag v1.2.1,
  scipy/stats/stats.py): `ks_2samp`, `mannwhitneyu(alternative=None)`,
  `ttest_ind(equal_var=False)`, `combine_pvalues` (fisher / stouffer),
  `tiecorrect`, `rankdata(method='average')`.

Pinning status: PARTIALLY PINNED.  The reference has no tests, fixtures or
golden vectors for this path.  `oracle/gen_golden.py` executes
the reference's own glue lines (myDetect.py:317-414, read from /root/reference
at generation time) with `ttest_ind`/`combine_pvalues` bound to the container's
real scipy (semantics unchanged since 1.2.1) and `ks_2samp`/`mannwhitneyu`
bound to the restatements below, and this module is checked against those
outputs and against the known-answer anchors KAT-1..4 of SURVEY.md §8c
(tests/test_oracle_golden.py).  The two restated scipy functions themselves are
cross-checked against scipy 1.15.3 where the semantics coincide (KS D exactly;
MWU == two-sided asymptotic p / 2 and min(U1,U2)); their 1.2.1 p-value
conventions are "parity unpinned" in the strict sense (scipy 1.2.1 cannot be
run here).

Numerical primitives used from the container (all unchanged in meaning since
1.2.1): `scipy.special.kolmogorov`, `ndtr`, `ndtri`, `stdtr`, `chdtrc`.
"""
from __future__ import annotations

import math
import sys

import numpy as np
from scipy import special as sc

DBL_MIN = sys.float_info.min   # 2.2250738585072014e-308
DBL_MAX = sys.float_info.max   # 1.7976931348623157e+308


class AllIdenticalError(ValueError):
    """scipy 1.2.1 mannwhitneyu raises ValueError('All numbers are identical in
    mannwhitneyu') when the tie correction T is 0; the reference does not catch
    it (myDetect.py:331) so a detect run aborts."""


# --------------------------------------------------------------------------
# myDetect.py:317-325
def m_min_float(fv):
    if fv < DBL_MIN:
        return DBL_MIN
    return fv


def m_max_float(fv):
    if fv > DBL_MAX:
        return DBL_MAX
    return fv


# --------------------------------------------------------------------------
# scipy 1.2.1 stats.ks_2samp (call site myDetect.py:341)
def ks_2samp(a, b):
    d1 = np.sort(np.asarray(a, dtype=np.float64))
    d2 = np.sort(np.asarray(b, dtype=np.float64))
    n1 = d1.shape[0]
    n2 = d2.shape[0]
    data_all = np.concatenate([d1, d2])
    cdf1 = np.searchsorted(d1, data_all, side='right') / (1.0 * n1)
    cdf2 = np.searchsorted(d2, data_all, side='right') / (1.0 * n2)
    d = np.max(np.absolute(cdf1 - cdf2))
    en = np.sqrt(n1 * n2 / float(n1 + n2))
    # kstwobign.sf == special.kolmogorov (scipy/stats/_continuous_distns.py)
    prob = float(sc.kolmogorov((en + 0.12 + 0.11 / en) * d))
    return float(d), prob


def ks_counts(a, b):
    """Exact integer form of the KS statistic: max |c0*n1 - c1*n0| over pooled
    values (c = searchsorted(side='right')).  D == this / (n0*n1) up to one
    fp64 rounding of the reference's float CDF subtraction."""
    d1 = np.sort(np.asarray(a, dtype=np.float64))
    d2 = np.sort(np.asarray(b, dtype=np.float64))
    data_all = np.concatenate([d1, d2])
    c0 = np.searchsorted(d1, data_all, side='right').astype(np.int64)
    c1 = np.searchsorted(d2, data_all, side='right').astype(np.int64)
    return int(np.max(np.abs(c0 * d2.shape[0] - c1 * d1.shape[0])))


# --------------------------------------------------------------------------
# scipy 1.2.1 stats.rankdata(method='average') / tiecorrect / mannwhitneyu
def rankdata_average(x):
    arr = np.ravel(np.asarray(x, dtype=np.float64))
    sorter = np.argsort(arr, kind='mergesort')
    inv = np.empty(sorter.size, dtype=np.intp)
    inv[sorter] = np.arange(sorter.size, dtype=np.intp)
    arr = arr[sorter]
    obs = np.r_[True, arr[1:] != arr[:-1]]
    dense = obs.cumsum()[inv]
    count = np.r_[np.nonzero(obs)[0], len(obs)]
    return .5 * (count[dense] + count[dense - 1] + 1)


def tiecorrect(rankvals):
    arr = np.sort(rankvals)
    idx = np.nonzero(np.r_[True, arr[1:] != arr[:-1], True])[0]
    cnt = np.diff(idx).astype(np.float64)
    size = np.float64(arr.size)
    return 1.0 if size < 2 else 1.0 - (cnt ** 3 - cnt).sum() / (size ** 3 - size)


def mannwhitneyu(x, y):
    """mannwhitneyu(x, y, use_continuity=True, alternative=None) of scipy 1.2.1
    (call site myDetect.py:331): legacy one-sided p, statistic min(U1, U2)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    n1 = len(x)
    n2 = len(y)
    ranked = rankdata_average(np.concatenate((x, y)))
    rankx = ranked[0:n1]
    u1 = n1 * n2 + (n1 * (n1 + 1)) / 2.0 - np.sum(rankx, axis=0)
    u2 = n1 * n2 - u1
    T = tiecorrect(ranked)
    if T == 0:
        raise AllIdenticalError('All numbers are identical in mannwhitneyu')
    sd = np.sqrt(T * n1 * n2 * (n1 + n2 + 1) / 12.0)
    meanrank = n1 * n2 / 2.0 + 0.5
    bigu = max(u1, u2)
    z = (bigu - meanrank) / sd
    p = float(sc.ndtr(-abs(z)))          # norm.sf(abs(z))
    u = min(u1, u2)
    return float(u), p


# --------------------------------------------------------------------------
# scipy stats.ttest_ind(a, b, equal_var=False) (call site myDetect.py:335)
def ttest_welch(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    n1 = a.shape[0]
    n2 = b.shape[0]
    with np.errstate(divide='ignore', invalid='ignore'):
        v1 = np.var(a, ddof=1)
        v2 = np.var(b, ddof=1)
        vn1 = v1 / n1
        vn2 = v2 / n2
        df = (vn1 + vn2) ** 2 / (vn1 ** 2 / (n1 - 1) + vn2 ** 2 / (n2 - 1))
        df = np.where(np.isnan(df), 1, df)
        denom = np.sqrt(vn1 + vn2)
        d = np.mean(a) - np.mean(b)
        t = np.divide(d, denom)
        prob = sc.stdtr(df, -np.abs(t)) * 2   # t.sf(|t|, df) * 2
    return float(t), float(prob)


# --------------------------------------------------------------------------
# scipy stats.combine_pvalues (call sites myDetect.py:393,401)
def combine_fisher(pvalues):
    pvalues = np.asarray(pvalues, dtype=np.float64)
    with np.errstate(divide='ignore'):
        Xsq = -2 * np.sum(np.log(pvalues))
    pval = float(sc.chdtrc(2 * len(pvalues), Xsq))   # chi2.sf(Xsq, 2k)
    return float(Xsq), pval


def combine_stouffer(pvalues, weights):
    pvalues = np.asarray(pvalues, dtype=np.float64)
    weights = np.asarray(weights, dtype=np.float64)
    with np.errstate(invalid='ignore'):
        Zi = -sc.ndtri(pvalues)                          # norm.isf(p)
        Z = np.dot(weights, Zi) / np.linalg.norm(weights)
    pval = float(sc.ndtr(-Z))                            # norm.sf(Z)
    return float(Z), pval


def stouffer_weights(nb, weights_dif):
    """myDetect.py:396-400: 100 in the middle, divided by WeightsDif per step."""
    mweights = [100]
    for _ in range(nb):
        mweights.insert(0, mweights[0] / weights_dif)
        mweights.append(mweights[-1] / weights_dif)
    return mweights


# --------------------------------------------------------------------------
# myDetect.py:327-343,363 (default branch; the unseeded down-sampling branch
# :345-361 is outside the parity contract, SURVEY.md §8a row A3')
def getKStest(a, b):
    st, pu = mannwhitneyu(a, b)
    pu = m_min_float(pu)
    stu = m_max_float(st)
    st, pt = ttest_welch(a, b)
    pt = m_min_float(pt)
    stt = m_max_float(st)
    st, pks = ks_2samp(a, b)
    pks = m_min_float(pks)
    stks = m_max_float(st)
    return [(stu, pu), (stt, pt), (stks, pks)]


# myDetect.py:345-361 — the down-sampling branch.  The reference draws from numpy's unseeded global RNG, so
# this restatement (taking an explicit Generator) is only statistically comparable with any other run.
def ks_downsampled(a, b, cov, iters=100, quantile=0.25, rng=None):
    rng = rng if rng is not None else np.random.default_rng()
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    p_array = np.zeros(iters); st_array = np.zeros(iters)
    for i in range(iters):
        a_temp = rng.choice(a, cov) if len(a) > cov else a
        b_temp = rng.choice(b, cov) if len(b) > cov else b
        st, pks = ks_2samp(a_temp, b_temp)
        p_array[i] = m_min_float(pks); st_array[i] = m_max_float(st)
    ind = np.argsort(p_array)[int(iters * quantile)]
    return float(st_array[ind]), float(p_array[ind])


# --------------------------------------------------------------------------
# Batch form on the build's CSR layout.
METHOD_KS, METHOD_STOUFFER, METHOD_FISHER = 0, 1, 2
STATUS_MWU_ALL_IDENTICAL = 1
STATUS_T_NAN = 2
STATUS_EMPTY = 4


def per_position_tests(sig0, off0, sig1, off1):
    """Returns dict of fp64 arrays mwu_u, mwu_p, t_t, t_p, ks_d, ks_p and uint8
    status.  Where the reference would raise (all identical), MWU is reported
    as (nan, nan) with STATUS_MWU_ALL_IDENTICAL set."""
    npos = len(off0) - 1
    out = {k: np.empty(npos, dtype=np.float64)
           for k in ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p')}
    status = np.zeros(npos, dtype=np.uint8)
    for i in range(npos):
        a = np.asarray(sig0[off0[i]:off0[i + 1]], dtype=np.float64)
        b = np.asarray(sig1[off1[i]:off1[i + 1]], dtype=np.float64)
        try:
            st, pu = mannwhitneyu(a, b)
            pu = m_min_float(pu)
            st = m_max_float(st)
        except AllIdenticalError:
            st, pu = math.nan, math.nan
            status[i] |= STATUS_MWU_ALL_IDENTICAL
        out['mwu_u'][i], out['mwu_p'][i] = st, pu
        st, pt = ttest_welch(a, b)
        if math.isnan(pt):
            status[i] |= STATUS_T_NAN
        out['t_t'][i], out['t_p'][i] = m_max_float(st), m_min_float(pt)
        st, pks = ks_2samp(a, b)
        out['ks_d'][i], out['ks_p'][i] = m_max_float(st), m_min_float(pks)
    out['status'] = status
    return out


def combine_track(ks_d, ks_p, run_id, nb, weights_dif, method):
    """myDetect.py:373-414 on a whole track.  `run_id[i]==run_id[j]` restates
    pos_check (myDetect.py:366-371).  method==METHOD_KS is the caller's business
    (mtest2 skips the combine, myDetect.py:443)."""
    npos = len(ks_p)
    st = np.empty(npos, dtype=np.float64)
    pv = np.empty(npos, dtype=np.float64)
    if nb == 0:                       # myDetect.py:413 returns the KS tuple
        return np.array(ks_d, dtype=np.float64), np.array(ks_p, dtype=np.float64)
    w = stouffer_weights(nb, weights_dif)
    for i in range(npos):
        win = []
        for j in range(i - nb, i + nb + 1):
            if j < 0 or j > npos - 1 or run_id[j] != run_id[i]:
                win.append(1.0)
            else:
                win.append(ks_p[j])
        if method == METHOD_FISHER:
            s, p = combine_fisher(win)
        else:
            s, p = combine_stouffer(win, w)
        pv[i] = m_min_float(p)
        st[i] = m_max_float(s)
    return st, pv


def detect_batch(sig0, off0, sig1, off1, run_id, nb=2, weights_dif=2.0,
                 method=METHOD_STOUFFER):
    out = per_position_tests(sig0, off0, sig1, off1)
    if method != METHOD_KS:
        out['comb_st'], out['comb_p'] = combine_track(
            out['ks_d'], out['ks_p'], run_id, nb, weights_dif, method)
    return out


# --------------------------------------------------------------------------
# myDetect.py:532-536 line format
def format_sign_test_line(chrom, strand, pos0, base, n0, n1, rec, with_comb):
    s = '%s %s %d %s %d %d %.3f %.3E %.3f %.3E %.3f %.3E' % (
        chrom, strand, pos0 + 1, base, n0, n1,
        rec[0][0], rec[0][1], rec[1][0], rec[1][1], rec[2][0], rec[2][1])
    if with_comb:
        s += ' %.3f %.3E\n' % (rec[3][0], rec[3][1])
    else:
        s += '\n'
    return s
