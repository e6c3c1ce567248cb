#!/usr/bin/env python3
"""Differential fuzz of the HIP path against the C oracle (run on the GPU box: python tests/fuzz_gpu.py [rounds]).
Random ragged batches: group sizes from 1 to ~9000 with random size ranges per batch, continuous / gridded /
heavily tied values on and off the milli-unit grid, float32 values beside / beyond the grid, optional int16 input (also heavily tied and spread over the whole domain), event-like rows (a level per position, narrow spread: the counting form), planted NaN / infinite samples (NMOD_STATUS_NONFINITE), both test masks.  Test infrastructure, like everything under oracle/."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nanomod_amd as nm            # noqa: E402
import oracle_c                     # noqa: E402
import helpers as H                 # noqa: E402

L = nm._lib


def run_round(rng):
    """one random batch through the HIP path (both test masks) against the oracle; returns a description"""
    hi0 = int(rng.choice([3, 8, 40, 64, 65, 130, 260, 600, 1100, 2048, 2500, 9000]))
    hi1 = int(rng.choice([3, 8, 40, 64, 65, 130, 260, 600, 1100, 2048, 2500, 9000]))
    lo0 = int(rng.integers(1, hi0 + 1)) if rng.random() < 0.5 else 1
    lo1 = int(rng.integers(1, hi1 + 1)) if rng.random() < 0.5 else 1
    mode = rng.choice(['cont', 'grid2', 'grid0', 'grid3', 'offt', 'gmix', 'i16', 'i16t', 'i16w', 'f64', 'f64near', 'nonf', 'evt', 'evt16', 'evt64', 'evto', 'evto16', 'evto64'])
    if rng.random() < 0.08:                   # both groups above 1 024 samples, event-like: the value-domain counting form (rank_count_value.hpp)
        lo0, lo1 = int(rng.integers(1000, 1400)), int(rng.integers(1000, 1400)); hi0, hi1 = int(rng.integers(lo0, 2049)), int(rng.integers(lo1, 2049))
        mode = rng.choice(['evt', 'evt16', 'evt64', 'evto', 'evto16', 'evto64', 'i16t', 'grid2'])
    npos = int(rng.integers(1, (400 if max(hi0, hi1) <= 600 else 40) // (4 if mode.startswith('f64') or mode in ('evt64', 'evto64') else 1) + 1))
    n0 = rng.integers(lo0, hi0 + 1, npos); n1 = rng.integers(lo1, hi1 + 1, npos)
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum(n1)
    a = rng.normal(0, 1, off0[-1]); b = rng.normal(rng.choice([0.0, 0.1, 1.0]), rng.choice([1.0, 0.3, 2.0]), off1[-1])
    if mode == 'grid2':
        a, b = np.round(a, 2), np.round(b, 2)
    elif mode == 'grid0':
        a, b = np.round(a, 0), np.round(b, 0)
    elif mode == 'grid3':                     # float32 on the milli-unit grid of real events (the counters of the WIDE form)
        a, b = np.round(a, 3), np.round(b, 3)
    elif mode == 'offt':                      # heavy ties OFF the milli-unit grid: the bitmap form's exact table or its redo list
        step = float(rng.choice([0.0137, 0.25001, 0.7003]))
        a, b = np.round(a / step) * step, np.round(b / step) * step
    elif mode == 'gmix':                      # on the grid, except some samples: off it by chance, by one float32 ulp, or far away
        a, b = np.round(a, 3), np.round(b, 3)
        for v in (a, b):
            k = rng.random(len(v)) < rng.choice([0.0005, 0.01, 0.3])
            v[k] = rng.choice([1.0, 7.0, 33.0, 4e4]) * rng.normal(0, 1, int(k.sum()))
    elif mode in ('evt', 'evt16', 'evt64', 'evto', 'evto16', 'evto64'):   # event-like rows: a level per position, a narrow spread, the milli-unit grid (the counting forms)
        lev = rng.uniform(-3, 3, npos)
        lev0 = np.repeat(lev, n0); lev1 = np.repeat(lev, n1)
        sp = float(rng.choice([0.02, 0.1, 0.2, 0.35, 0.6]))
        a = np.round(lev0 + sp * rng.normal(0, 1, len(a)), 3); b = np.round(lev1 + sp * rng.normal(rng.choice([0.0, 1.0]), 1, len(b)), 3)
        if mode.startswith('evto'):           # ... with outliers: reads anywhere in the +-5 unit clip range (the counting forms' tail lists), some of them repeated
            frac = float(rng.choice([0.001, 0.01, 0.05, 0.2]))
            for v in (a, b):
                hit = rng.random(len(v)) < frac
                v[hit] = np.round(rng.uniform(-5, 5, int(hit.sum())), int(rng.choice([3, 1, 0])))
    if mode == 'i16t':                        # int16 with heavy ties: up to hundreds of copies of a value (8-bit counters wrap)
        a, b = np.round(a, 1), np.round(b, int(rng.integers(0, 3)))
    elif mode == 'i16w':                      # int16 over most of the domain: many count windows
        a, b = np.clip(a * 8, -32.7, 32.7), np.clip(b * 8, -32.7, 32.7)
    if mode.startswith('i16') or mode in ('evt16', 'evto16'):
        s0 = np.round(a * 1000).astype(np.int16); s1 = np.round(b * 1000).astype(np.int16)
        r0, r1 = s0.astype(np.float64) / 1000, s1.astype(np.float64) / 1000
    elif mode in ('evt64', 'evto64'):         # the reference's own rows: float64 values k / 1000.0 (the float64 front end: integer keys)
        s0 = np.round(a * 1000) / 1000.0; s1 = np.round(b * 1000) / 1000.0
        r0, r1 = s0, s1
    elif mode in ('f64', 'f64near'):
        # float64 input: arbitrary doubles; 'f64near' plants doubles that differ below float32 resolution and exact copies
        s0 = a.copy(); s1 = b.copy()
        if mode == 'f64near':
            k = min(len(s0), len(s1)) // 2
            idx = rng.integers(0, len(s0), k)
            s1[:k] = s0[idx] * (1.0 + rng.choice([0.0, 2.0 ** -30, -2.0 ** -40], k))
        r0, r1 = s0, s1
    else:
        s0 = a.astype(np.float32); s1 = b.astype(np.float32)
        if mode == 'gmix' and len(s0) > 3 and len(s1) > 3:
            s0[::97] = np.nextafter(s0[::97], np.float32(1e9)); s1[1::53] = np.nextafter(s1[1::53], np.float32(-1e9))
        r0, r1 = s0, s1
    rid = np.cumsum(rng.random(npos) < 0.1).astype(np.int32)
    nb = int(rng.integers(0, 4)); method = str(rng.choice(['stouffer', 'fisher']))
    if mode == 'nonf':
        return nonfinite_round(rng, s0, off0, s1, off1, rid, nb, method, (lo0, hi0, lo1, hi1))
    if s0.dtype == np.float64:                # the C oracle takes float32 / int16: the numpy restatement on the doubles
        import nanomod_oracle as orc
        exp = orc.detect_batch(r0, off0, r1, off1, rid, nb, 2.0, orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
    else:
        exp = oracle_c.detect_batch(s0, off0, s1, off1, rid, nb, 2.0, method, threads=0)
    # float64 rows: through the device's float64 front end, or (where every sample of a chunk is k / 1000.0) narrowed to int16 on the host
    hf = L.FLAG_NO_HOST_NARROW if (s0.dtype == np.float64 and rng.random() < 0.5) else 0
    got = nm.detect_host(s0, off0, s1, off1, rid, nb=nb, weights_dif=2.0, method=method, flags=hf)
    ident = (exp['status'] & 1) != 0          # MWU all identical: U / p NaN on both sides
    assert np.all(np.isnan(got['mwu_u'][ident])) and np.all(np.isnan(exp['mwu_u'][ident]))
    got['mwu_u'][ident] = 0.0; exp['mwu_u'][ident] = 0.0
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    assert np.array_equal(got['status'], exp['status'])
    ks = nm.detect_host(s0, off0, s1, off1, rid, nb=nb, weights_dif=2.0, method=method, tests=L.TEST_KS, flags=hf)
    H.assert_close_stat(ks['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    H.assert_close_p(ks['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    return 'npos %d sizes [%d..%d] x [%d..%d] %s nb %d %s identical %d' % (npos, lo0, hi0, lo1, hi1, mode, nb, method, ident.sum())


def nonfinite_round(rng, s0, off0, s1, off1, rid, nb, method, sizes):
    """NaN / +-inf planted in a few positions: with NMOD_FLAG_CHECK_FINITE exactly those positions carry NMOD_STATUS_NONFINITE
    (both test masks); without it NaN and -inf are flagged by the moments whenever all tests run; every other position keeps
    the oracle's per-position numbers"""
    npos = len(rid)
    exp = oracle_c.detect_batch(s0, off0, s1, off1, rid, 0, 2.0, method, threads=0)
    b0, b1 = s0.copy(), s1.copy()
    planted = np.zeros(npos, bool); sure = np.zeros(npos, bool)
    for pos in rng.choice(npos, size=min(npos, 1 + npos // 7), replace=False):
        g = int(rng.integers(0, 2)); v = rng.choice([np.nan, -np.inf, np.inf])
        arr, off = (b1, off1) if g else (b0, off0)
        arr[off[pos] + int(rng.integers(0, off[pos + 1] - off[pos]))] = v
        planted[pos] = True; sure[pos] = sure[pos] or not (v == np.inf)
    keep = ~planted
    for tests in (L.TEST_ALL, L.TEST_KS):
        for flags in (L.FLAG_CHECK_FINITE, 0):
            got = nm.detect_host(b0, off0, b1, off1, rid, nb=0, weights_dif=2.0, method=method, tests=tests, flags=flags)
            nf = (got['status'] & L.STATUS_NONFINITE) != 0
            if flags:
                assert np.array_equal(nf, planted)
            else:
                assert not nf[keep].any() and (tests != L.TEST_ALL or nf[sure].all())
            assert np.array_equal(got['ks_d'][keep], exp['ks_d'][keep])
            H.assert_close_p(got['ks_p'][keep], exp['ks_p'][keep], 1e-9, 'ks_p')
            if tests == L.TEST_ALL:
                ident = ((exp['status'] & 1) != 0) | planted
                assert np.array_equal(got['mwu_u'][~ident], exp['mwu_u'][~ident])
    return 'npos %d sizes [%d..%d] x [%d..%d] nonf planted %d' % (npos, sizes[0], sizes[1], sizes[2], sizes[3], planted.sum())


def run(rounds, seed=12345, log=print):
    master = np.random.default_rng(seed)
    for it in range(rounds):
        log('round %d ok: %s' % (it, run_round(np.random.default_rng(master.integers(1 << 62)))))


if __name__ == '__main__':
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 12345,
        log=lambda m: print(m, flush=True))
    print('fuzz ok')
