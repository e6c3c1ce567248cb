"""CPU: the C restatement (oracle/nanomod_oracle.c) against the Python oracle and the golden fixtures."""
import numpy as np
import pytest

import helpers as H
import nanomod_oracle as orc

oracle_c = pytest.importorskip('oracle_c', reason='make -C oracle')


def _batch(rng, npos, lo, hi, grid):
    ca, cb = [], []
    for i in range(npos):
        a = rng.normal(0, 1, rng.integers(lo, hi + 1))
        b = rng.normal(0.5 if i % 5 == 0 else 0, 1.2, rng.integers(lo, hi + 1))
        if grid:
            a, b = np.round(a, 2), np.round(b, 2)
        ca.append(a.astype(np.float32)); cb.append(b.astype(np.float32))
    o0 = np.zeros(npos + 1, np.int64); o0[1:] = np.cumsum([len(c) for c in ca])
    o1 = np.zeros(npos + 1, np.int64); o1[1:] = np.cumsum([len(c) for c in cb])
    return np.concatenate(ca), o0, np.concatenate(cb), o1, (np.arange(npos) // 13).astype(np.int32)


@pytest.mark.parametrize('grid', [False, True])
@pytest.mark.parametrize('method', ['stouffer', 'fisher'])
def test_c_matches_python_oracle(grid, method):
    rng = np.random.default_rng(11 + grid)
    s0, o0, s1, o1, rid = _batch(rng, 300, 3, 400, grid)
    exp = orc.detect_batch(s0, o0, s1, o1, rid, 2, 2.0, orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
    got = oracle_c.detect_batch(s0, o0, s1, o1, rid, 2, 2.0, method)
    H.compare_outputs(got, exp, True)
    assert np.array_equal(got['ks_d'], exp['ks_d']) and np.array_equal(got['status'], exp['status'])


def test_c_edge_cases_and_int16():
    s0 = np.array([0.5] * 6 + [1.0] * 5, np.float32); o0 = np.array([0, 6, 11])
    s1 = np.array([0.5] * 7 + [2.0] * 5, np.float32); o1 = np.array([0, 7, 12])
    g = oracle_c.detect_batch(s0, o0, s1, o1, np.zeros(2, np.int32), method='ks')
    e = orc.detect_batch(s0, o0, s1, o1, np.zeros(2, np.int32), method=orc.METHOD_KS)
    assert np.array_equal(g['status'], e['status']) and g['status'][0] == 3
    assert g['t_t'][1] == -np.inf and g['t_p'][1] == orc.DBL_MIN and g['ks_d'][0] == 0 and g['ks_p'][0] == 1
    rng = np.random.default_rng(5)
    f0, o0, f1, o1, rid = _batch(rng, 100, 5, 200, False)
    k0 = np.rint(f0 * 1000.0).astype(np.int16); k1 = np.rint(f1 * 1000.0).astype(np.int16)
    g = oracle_c.detect_batch(k0, o0, k1, o1, rid)
    e = orc.detect_batch(k0 / 1000.0, o0, k1 / 1000.0, o1, rid)
    H.compare_outputs(g, e, True)


def test_c_welch_sums_like_numpy():
    """np.mean / np.var under scipy's ttest_ind sum pairwise (numpy's add.reduce): on rows that share a large level the order of
    summation is visible in t at 1e-13; the C restatement follows numpy's order and gives the Python oracle's t bit for bit"""
    for dt in ('i16', 'f32'):
        for n in (5, 8, 37, 128, 129, 200, 255, 1000):
            P = 120
            a = H.synth_events_ref(11, 0, P, 0, n, 0, 0, 100, dt); b = H.synth_events_ref(11, 0, P, 1, n, 0, 0, 100, dt)
            off = np.arange(0, (P + 1) * n, n, dtype=np.int64)
            got = oracle_c.detect_batch(a.reshape(-1), off, b.reshape(-1), off, np.zeros(P, np.int32), 2, 2.0, 'fisher', tests=7)
            sc = 1000.0 if dt == 'i16' else 1.0
            tt = np.array([orc.ttest_welch(a[i].astype(np.float64) / sc, b[i].astype(np.float64) / sc)[0] for i in range(P)])
            assert np.array_equal(tt, got['t_t']), (dt, n)
    # the gate the GPU tests use for t on such rows (helpers.t_abs_gate) is ~1e-13 there and 2e-14 around zero
    g = H.t_abs_gate(a.reshape(-1), off, b.reshape(-1), off)
    assert g.shape == (P,) and 2e-14 < np.median(g) < 2e-12
    z = np.random.default_rng(1).normal(0, 1, 800).astype(np.float32)
    assert H.t_abs_gate(z[:400], np.array([0, 200, 400]), z[400:], np.array([0, 200, 400])).max() < 3e-14


def test_c_special_function_tails():
    """p-values down to DBL_MIN: extreme separation, tiny windows"""
    n = 2000
    a = np.linspace(0, 1, n).astype(np.float32); b = (np.linspace(0, 1, n) + 5).astype(np.float32)
    o = np.array([0, n])
    g = oracle_c.detect_batch(a, o, b, o, np.zeros(1, np.int32), method='ks')
    e = orc.detect_batch(a, o, b, o, np.zeros(1, np.int32), method=orc.METHOD_KS)
    H.compare_outputs(g, e, False)
    ks_p = np.array([1e-300, 1e-200, 2.3e-308, 0.5, 1.0, 1e-17, 0.999999])
    for m, mid in (('stouffer', orc.METHOD_STOUFFER), ('fisher', orc.METHOD_FISHER)):
        est, ep = orc.combine_track(np.zeros(7), ks_p, np.zeros(7, np.int32), 2, 2.0, mid)
        # drive the C combine through detect's track entry: reuse ks track by a tiny batch is not possible,
        # so check the primitives through a 1-run track with nb=2 on synthetic KS p-values is covered in
        # test_c_matches_python_oracle; here only assert the Python oracle's own tail behaviour is finite
        assert np.all(np.isfinite(ep)) and np.all(ep >= orc.DBL_MIN)
