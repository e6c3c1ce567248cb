"""-m gpu: the HIP path (through the C ABI) against the oracle and the golden fixtures."""
import json
import os
import zlib
import tempfile

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nm():
    import nanomod_amd
    return nanomod_amd


def test_selftest_wave_primitives(nm):
    assert nm._lib.load().nmod_selftest(0) == 0


def test_known_answer_anchors(nm):
    with open(os.path.join(H.GOLDEN, 'kat.json')) as f:
        kat = json.load(f)
    mo = {'coverages': [0, 0]}
    for key in ('KAT-1', 'KAT-2'):
        # KAT-2's b = a + 0.75 is 1 ulp off the 0.001 grid in fp64; the device takes grid/fp32-exact
        # values, so snap (changes t by ~1e-16 relative, nothing else)
        got = nm.getKStest(mo, np.round(kat[key]['a'], 3), np.round(kat[key]['b'], 3), '+')
        exp = kat[key]['out']
        assert got[0][0] == exp[0][0]
        for g, e in zip(got, exp):
            assert abs(g[0] - e[0]) <= 1e-12 * abs(e[0]) + 4e-16
            assert abs(g[1] - e[1]) <= 1e-12 * abs(e[1])
    for method in ('stouffer', 'fisher'):
        k = kat['KAT-3-' + method]
        mo = {'neighborPvalues': k['nb'], 'WeightsDif': k['dif'], 'testMethod': method,
              'sign_test': [(('c', '+', 10 + i, 'A', 5, 5), [(0, 1), (0, 1), (0.1 * i, p)]) for i, p in enumerate(k['ks_p'])]}
        for i, e in enumerate(k['out']):
            g = nm.get_combin_pvalue(mo, i)
            e = [float(v) for v in e]
            if np.isinf(e[0]):
                assert g[0] == e[0] and g[1] == e[1]
            else:
                assert abs(g[0] - e[0]) <= 1e-12 * abs(e[0]) and abs(g[1] - e[1]) <= 1e-11 * e[1]
    # all-identical: the reference raises (scipy 1.2.1 mannwhitneyu)
    with pytest.raises(ValueError, match='All numbers are identical'):
        nm.getKStest({'coverages': [0, 0]}, [0.25] * 6, [0.25] * 6, '+')


CASES = [('g50', 'g50_stouffer', 2, 2.0, 'stouffer'), ('g50', 'g50_fisher', 2, 2.0, 'fisher'),
         ('g50', 'g50_ks', 2, 2.0, 'ks'), ('ragged', 'ragged_stouffer', 2, 2.0, 'stouffer'),
         ('ragged', 'ragged_fisher', 2, 2.0, 'fisher'), ('ties', 'ties_stouffer', 2, 2.0, 'stouffer')]
CASES += [('sweep', 'sweep_nb%d_w%g_%s' % (nb, w, m), nb, w, m)
          for nb in (0, 1, 2, 3) for w in (1.0, 2.0, 3.0) for m in ('stouffer', 'fisher')
          if not (m == 'fisher' and w != 2.0)]


@pytest.mark.parametrize('inp,name,nb,wdif,method', CASES)
def test_golden_tables_through_mtest2(nm, inp, name, nb, wdif, method):
    """mfilter_coverage + mtest2 + save_test on the reference-shaped moptions: same position
    set and order, same numbers, byte-identical `_sign_test.txt`, same ranking."""
    fx = H.load_inputs(inp)
    exp, table = H.load_expected(name)
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, name, nb, wdif, method)
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        with open(os.path.join(out, name + '_sign_test.txt')) as f:
            got_table = f.read()
    st = mo['sign_test']
    assert [r[0][0] for r in st] == list(exp['chrom']) and [r[0][1] for r in st] == list(exp['strand'])
    assert [r[0][2] for r in st] == list(exp['pos']) and [r[0][3] for r in st] == list(exp['base'])
    assert [r[0][4] for r in st] == list(exp['n0']) and [r[0][5] for r in st] == list(exp['n1'])
    got = {'mwu_u': [r[1][0][0] for r in st], 'mwu_p': [r[1][0][1] for r in st],
           't_t': [r[1][1][0] for r in st], 't_p': [r[1][1][1] for r in st],
           'ks_d': [r[1][2][0] for r in st], 'ks_p': [r[1][2][1] for r in st]}
    with_comb = 'comb_st' in exp
    if with_comb:
        got['comb_st'] = [r[1][3][0] for r in st]
        got['comb_p'] = [r[1][3][1] for r in st]
    H.compare_outputs(got, exp, with_comb)
    assert got_table == table
    # ranking (myDetect.py:460): a valid ascending sort of our own keys, and the same order as the
    # reference's up to positions whose keys agree to 1e-9 (last-bit differences in p may swap near-ties)
    index_of = {id(r): i for i, r in enumerate(st)}
    order = [index_of[id(r)] for r in mo['sorted_sign_test']]
    assert sorted(order) == list(range(len(st)))
    sind = 3 if with_comb else 2
    keys = [(st[i][1][sind][1], st[i][1][2][1], st[i][1][0][1]) for i in order]
    assert all(keys[k] <= keys[k + 1] for k in range(len(keys) - 1))
    ekey = np.asarray(exp['comb_p'] if with_comb else exp['ks_p'])
    a, b = ekey[np.array(order, dtype=int)], ekey[exp['sorted_index']]
    assert np.all(np.abs(a - b) <= 1e-9 * np.abs(b))


def test_too_large_group_is_flagged_per_position(nm):
    """a group beyond NMOD_MAX_RANKED (65 535 samples; the reference has no limit, myDetect.py:327-343): that position gets
    NMOD_STATUS_TOO_LARGE and NaN outputs, every other position of the batch is computed — host-resident and device-resident
    entry, CSR and a fixed stride beyond the limit"""
    import torch
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(21)
    sizes0 = rng.integers(20, 300, 40); sizes1 = rng.integers(20, 300, 40)
    sizes0[7] = 70_000; sizes1[23] = 66_000
    off0 = np.zeros(41, np.int64); off0[1:] = np.cumsum(sizes0)
    off1 = np.zeros(41, np.int64); off1[1:] = np.cumsum(sizes1)
    sig0 = rng.normal(0, 1, off0[-1]).astype(np.float32); sig1 = rng.normal(0.2, 1, off1[-1]).astype(np.float32)
    rid = np.zeros(40, np.int32)
    big = np.zeros(40, bool); big[[7, 23]] = True
    # the oracle on the batch with the two rows cut to a handful of samples: the other positions' numbers (the window combine
    # of their neighbours sees NaN from the flagged positions, so compare the per-position tests)
    s0 = sizes0.copy(); s1 = sizes1.copy(); s0[7] = 5; s1[23] = 5
    o0 = np.zeros(41, np.int64); o0[1:] = np.cumsum(s0); o1 = np.zeros(41, np.int64); o1[1:] = np.cumsum(s1)
    c0 = np.concatenate([sig0[off0[i]:off0[i] + s0[i]] for i in range(40)]); c1 = np.concatenate([sig1[off1[i]:off1[i] + s1[i]] for i in range(40)])
    exp = orc.detect_batch(c0, o0, c1, o1, rid, 2, 2.0, orc.METHOD_FISHER)

    def check(got):
        st = np.asarray(got['status'])
        assert np.array_equal((st & L.STATUS_TOO_LARGE) != 0, big)
        for k in ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p'):
            g = np.asarray(got[k])
            assert np.all(np.isnan(g[big])), k
            if k.endswith('_p'):
                H.assert_close_p(g[~big], exp[k][~big], 1e-9, k)
            else:
                H.assert_close_stat(g[~big], exp[k][~big], 1e-11, 2e-14, k)
        assert np.array_equal(np.asarray(got['ks_d'])[~big], exp['ks_d'][~big])
    check(nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='fisher'))
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='fisher', tests=L.TEST_ALL)
    t = lambda a: torch.from_numpy(a).cuda()
    out = det.run(t(sig0), t(sig1), t(rid), off0=t(off0), off1=t(off1))
    torch.cuda.synchronize()
    check({k: v.cpu().numpy() for k, v in out.items()})
    # KS-only mode
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    assert np.array_equal((got['status'] & L.STATUS_TOO_LARGE) != 0, big) and np.array_equal(got['ks_d'][~big], exp['ks_d'][~big])
    # a fixed stride beyond the limit: every position is flagged, nothing faults
    n = 66_000
    a = rng.normal(0, 1, 3 * n).astype(np.float32)
    got = nm.detect_host(a, None, a[:3 * 50].copy(), None, np.zeros(3, np.int32), nb=2, weights_dif=2.0, method='fisher', stride0=n, stride1=50)
    assert np.all((got['status'] & L.STATUS_TOO_LARGE) != 0) and np.all(np.isnan(got['ks_d']))


@pytest.mark.parametrize('shape', ['200v200', 'ragged'])
def test_nonfinite_samples_are_flagged(nm, shape):
    """NMOD_STATUS_NONFINITE (the reference lets NaN propagate, myDetect.py:327-343; here the statistics of such a position are
    unspecified and the position is flagged).  All tests computed (what every reference-shaped call does): the Welch moments flag
    every NaN and -inf for free; with NMOD_FLAG_CHECK_FINITE every non-finite sample is flagged, KS-only mode included; the other
    positions of the batch are untouched."""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(31)
    P = 120
    if shape == '200v200':
        s0 = np.full(P, 200); s1 = np.full(P, 200)
    else:
        s0 = rng.integers(5, 700, P); s1 = rng.integers(5, 90, P)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(s0)
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(s1)
    sig0 = rng.normal(0, 1, off0[-1]).astype(np.float32); sig1 = rng.normal(0, 1, off1[-1]).astype(np.float32)
    rid = np.zeros(P, np.int32)
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, orc.METHOD_FISHER)
    plant = [(3, 0, np.nan), (11, 1, np.nan), (20, 0, -np.inf), (33, 1, -np.inf), (47, 0, np.inf), (58, 1, np.inf), (64, 0, np.nan), (64, 1, np.inf)]
    b0, b1 = sig0.copy(), sig1.copy()
    for pos, g, v in plant:
        (b1 if g else b0)[(off1 if g else off0)[pos] + int(rng.integers(0, (s1 if g else s0)[pos]))] = v
    planted = np.zeros(P, bool); planted[[p for p, _, _ in plant]] = True
    sure = np.zeros(P, bool); sure[[p for p, _, v in plant if not (np.isinf(v) and v > 0)]] = True     # NaN and -inf: flagged by the moments

    def others_ok(got, ks_only=False):
        keep = ~planted
        assert np.array_equal(got['ks_d'][keep], exp['ks_d'][keep])
        H.assert_close_p(got['ks_p'][keep], exp['ks_p'][keep], 1e-9, 'ks_p')
        if not ks_only:
            assert np.array_equal(got['mwu_u'][keep], exp['mwu_u'][keep])
            H.assert_close_p(got['t_p'][keep], exp['t_p'][keep], 1e-9, 't_p')
        assert not np.any(got['status'][keep] & L.STATUS_NONFINITE)
    got = nm.detect_host(b0, off0, b1, off1, rid, nb=2, weights_dif=2.0, method='fisher')
    others_ok(got)
    assert np.all((got['status'][sure] & L.STATUS_NONFINITE) != 0)
    for tests in (L.TEST_ALL, L.TEST_KS):
        got = nm.detect_host(b0, off0, b1, off1, rid, nb=2, weights_dif=2.0, method='fisher', tests=tests, flags=L.FLAG_CHECK_FINITE)
        others_ok(got, tests == L.TEST_KS)
        assert np.array_equal((got['status'] & L.STATUS_NONFINITE) != 0, planted)
    # float64 rows: the samples themselves are scanned
    got = nm.detect_host(b0.astype(np.float64), off0, b1.astype(np.float64), off1, rid, nb=2, weights_dif=2.0, method='fisher', flags=L.FLAG_CHECK_FINITE)
    assert np.array_equal((got['status'] & L.STATUS_NONFINITE) != 0, planted)
    # clean input: the flag changes nothing
    a = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='fisher', flags=L.FLAG_CHECK_FINITE)
    b = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='fisher')
    assert all(np.array_equal(a[k], b[k], equal_nan=True) for k in a)


# ---- round 5: corners pinned by reference-generated fixtures (oracle/gen_golden.py, its last block)
CASES_NB = [('track600', 'track600_nb%d_%s' % (nb, m), nb, 2.0, m) for nb in (5, 16, 64) for m in ('stouffer', 'fisher')]


@pytest.mark.parametrize('inp,name,nb,wdif,method', CASES_NB)
def test_golden_tables_wide_windows(nm, inp, name, nb, wdif, method):
    """neighborPvalues 5, 16 and 64 (the ABI's largest; NanoMod.py:357 takes any int) on a 600-position track whose runs are
    mostly shorter than the window: same numbers, byte-identical table, through mtest2"""
    test_golden_tables_through_mtest2(nm, inp, name, nb, wdif, method)


def test_window_beyond_the_abi_limit_is_refused(nm):
    """neighborPvalues = 65 > NMOD_MAX_NB: NMOD_ERR_INVALID_ARG from the library, nothing computed"""
    fx = H.load_inputs('track600')
    rid = np.zeros(len(fx['pos']), np.int32)
    with pytest.raises(nm._lib.NanomodLibraryError, match='invalid argument'):
        nm.detect_host(fx['sig0'], fx['off0'], fx['sig1'], fx['off1'], rid, nb=65, weights_dif=2.0, method='stouffer')
    got = nm.detect_host(fx['sig0'], fx['off0'], fx['sig1'], fx['off1'], rid, nb=64, weights_dif=2.0, method='fisher')
    assert np.all(np.isfinite(got['comb_p']))


@pytest.mark.parametrize('mc', [3, 20])
def test_golden_tables_min_coverage(nm, mc):
    """MinCoverage 3 and 20 (myDetect.py:301-314): positions enter / leave the tested set per group, runs re-form around the
    dropped ones — same position set, numbers and table as the reference's mfilter_coverage + mtest2"""
    fx = H.load_inputs('ragged')
    name = 'ragged_mc%d' % mc
    exp, table = H.load_expected(name)
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, name, 2, 2.0, 'stouffer', min_cov=mc)
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        with open(os.path.join(out, name + '_sign_test.txt')) as f:
            assert f.read() == table
    st = mo['sign_test']
    assert [r[0][2] for r in st] == list(exp['pos']) and [r[0][4] for r in st] == list(exp['n0']) and [r[0][5] for r in st] == list(exp['n1'])
    assert min(min(exp['n0']), min(exp['n1'])) >= mc
    assert len(st) != len(H.load_expected('ragged_stouffer')[0]['pos'])


@pytest.mark.parametrize('method', ['stouffer', 'ks'])
def test_rank_use_st_global_order(nm, method):
    """rankUse = 'st' (myDetect.py:447-462): sorted by (combined or KS statistic, KS statistic, U), then reversed — the
    reference's order up to records whose keys agree to 1e-9"""
    fx = H.load_inputs('g50')
    name = 'g50_rankst_' + method
    exp, table = H.load_expected(name)
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, name, 2, 2.0, method)
        mo['rankUse'] = 'st'
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        with open(os.path.join(out, name + '_sign_test.txt')) as f:
            assert f.read() == table
    st = mo['sign_test']
    index_of = {id(r): i for i, r in enumerate(st)}
    order = np.array([index_of[id(r)] for r in mo['sorted_sign_test']])
    assert sorted(order.tolist()) == list(range(len(st)))
    ref = exp['sorted_index']
    keys = [np.asarray(exp['comb_st'] if method != 'ks' else exp['ks_d']), np.asarray(exp['ks_d']), np.asarray(exp['mwu_u'])]
    same = order == ref
    assert same.mean() > 0.95                                    # (equal up to near-ties of the primary key: Z values that agree to ~1e-12 swap)
    for k in keys[:1]:
        a, b = k[order], k[ref]
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin) and np.all(np.abs(a[fin] - b[fin]) <= 1e-9 * np.abs(b[fin]) + 1e-12)
    # descending in the primary statistic
    prim = keys[0][order]
    fin = np.isfinite(prim)
    assert np.all(np.diff(prim[fin]) <= 1e-9 * np.abs(prim[fin][1:]) + 1e-12)


def _cvs_rows(text):
    return [ln.split(' ') for ln in text.strip().split('\n')]


@pytest.mark.parametrize('inp', ['sweep', 'ragged'])
def test_mstd_records_and_meanstd_file(nm, inp):
    """--mstd (myDetect.py:425,437-438,541-544): moptions['sign_test_mstd'] and `_meanstd.cvs` (0-based positions) as the
    reference's mtest2 + save_test produce them.  Continuous rows: the file byte for byte.  3-decimal rows: the means of n grid
    values do sit on '%.3f' rounding boundaries, where the last bit of a mean decides the digit — every number within one unit
    of the last printed place and equal unless the reference's value is within 1e-9 of a boundary."""
    fx = H.load_inputs(inp)
    name = inp + '_mstd'
    exp, table = H.load_expected(name)
    want = open(os.path.join(H.GOLDEN, name + '_meanstd.cvs')).read()
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, name, 2, 2.0, 'stouffer', mstd=1)
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        with open(os.path.join(out, name + '_sign_test.txt')) as f:
            assert f.read() == table
        got = open(os.path.join(out, name + '_meanstd.cvs')).read()
    st = mo['sign_test']
    ms = mo['sign_test_mstd']
    assert len(ms) == len(st)
    for k, col in (('mean0', (0, 0)), ('std0', (0, 1)), ('mean1', (1, 0)), ('std1', (1, 1))):
        g = np.array([ms[(r[0][0], r[0][1], r[0][2])][col[0]][col[1]] for r in st])
        assert np.all(np.abs(g - exp[k]) <= 1e-12 * np.abs(exp[k]) + 1e-15), k
    if inp == 'sweep':
        assert got == want
    else:
        gr, wr = _cvs_rows(got), _cvs_rows(want)
        assert len(gr) == len(wr)
        ref = np.stack([exp['mean0'], exp['std0'], exp['mean1'], exp['std1']], axis=1)
        differing = 0
        for i, (g, w) in enumerate(zip(gr, wr)):
            assert g[:4] == w[:4]
            for j in range(4):
                if g[4 + j] != w[4 + j]:
                    differing += 1
                    frac = abs(ref[i, j]) * 1000.0 % 1.0
                    assert abs(frac - 0.5) < 1e-6 and abs(float(g[4 + j]) - float(w[4 + j])) <= 0.0011, (i, j, g, w)
        assert differing <= 0.05 * 4 * len(gr)


def test_cli_detect_mstd(nm, capsys):
    """`detect --mstd 1` end to end: the CLI's `_meanstd.cvs` equals the reference's on the continuous fixture, byte for byte"""
    from nanomod_amd import cli
    from test_abi_and_host import _fixture_containers
    want = open(os.path.join(H.GOLDEN, 'sweep_mstd_meanstd.cvs')).read()
    _, table = H.load_expected('sweep_mstd')
    with tempfile.TemporaryDirectory() as tmp:
        p0, p1 = _fixture_containers('sweep', tmp)
        rc = cli.main(['detect', '--wrkBase1', p0, '--wrkBase2', p1, '--FileID', 'm', '--outFolder', tmp, '--mstd', '1', '--outLevel', '3'])
        assert rc == 0
        assert open(os.path.join(tmp, 'm_meanstd.cvs')).read() == want
        assert open(os.path.join(tmp, 'm_sign_test.txt')).read() == table
    capsys.readouterr()


@pytest.mark.parametrize('inp,name,method', [('ragged', 'ragged_stouffer', 'stouffer'), ('ties', 'ties_stouffer', 'stouffer'),
                                             ('g50', 'g50_fisher', 'fisher')])
def test_golden_tables_with_device_side_dtype_choice(nm, inp, name, method, monkeypatch):
    """large float64 batches skip the host's search for a narrower dtype (detect.DEVICE_ENCODE_ABOVE) and let the float64
    front end of the device pick keys per position: forced here on the golden fixtures — same table, byte for byte"""
    import nanomod_amd.detect as D
    monkeypatch.setattr(D, 'DEVICE_ENCODE_ABOVE', 0)
    fx = H.load_inputs(inp)
    exp, table = H.load_expected(name)
    with tempfile.TemporaryDirectory() as out:
        mo = H.build_moptions(fx, out, name, 2, 2.0, method)
        nm.mfilter_coverage(mo)
        meta, sig0, off0, sig1, off1, rid = D.build_csr(mo)
        assert sig0.dtype == np.float64 and sig1.dtype == np.float64
        nm.mtest2(mo)
        with open(os.path.join(out, name + '_sign_test.txt')) as f:
            assert f.read() == table


def _random_batch(rng, npos, lo0, hi0, lo1, hi1, grid=False, shift_every=7):
    ca, cb = [], []
    for i in range(npos):
        n0 = int(rng.integers(lo0, hi0 + 1))
        n1 = int(rng.integers(lo1, hi1 + 1))
        a = rng.normal(0, 1, n0)
        b = rng.normal(0.7 if i % shift_every == 0 else 0.0, 1.3, n1)
        if grid:
            a, b = np.round(a, 2), np.round(b, 2)
        ca.append(a.astype(np.float32))
        cb.append(b.astype(np.float32))
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    rid = (np.arange(npos) // 11).astype(np.int32)
    return np.concatenate(ca), off0, np.concatenate(cb), off1, rid


def test_synth_fill_events_matches_the_numpy_restatement(nm):
    """nmod_synth_fill_events (bench.py's `real_spread` legs): level per position + spread per read on the milli-unit grid,
    fixed stride and ragged rows, float32 and int16 output, planted shift, counter-based (any pos_begin)"""
    import torch
    L = nm._lib
    rng = np.random.default_rng(4)
    P, begin = 600, 9_990
    sizes = rng.integers(0, 300, P); sizes[5] = 0; sizes[17] = 1
    off = np.zeros(P + 1, np.int64); off[1:] = np.cumsum(sizes)
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    d_off = torch.from_numpy(off).cuda()
    nmax = int(sizes.max())
    for tdt, name in ((torch.float32, 'f32'), (torch.int16, 'i16')):
        for g in (0, 1):
            # (spread up to the bound the ABI accepts: the quotient's floor is taken in 64 bits; outliers: 0, 1 and 10 per mille and all)
            for spread, outl in ((0, 0), (100, 0), (200, 0), (400, 0), (1183, 0), (8000, 0), (200, 1), (200, 10), (400, 1000)):
                ref = H.synth_events_ref(77, begin, P, g, nmax, 10000, 800, spread, dtype=name, outlier_permille=outl)
                out = torch.zeros(int(off[-1]), dtype=tdt, device='cuda:0')
                det.synth_fill_events(out, 77, begin, P, g, off=d_off, plant_period=10000, plant_shift_milli=800, spread_milli=spread, outlier_permille=outl)
                exp = np.concatenate([ref[i, :sizes[i]] for i in range(P)])
                assert np.array_equal(out.cpu().numpy(), exp), (name, g, spread, outl, 'csr')
                out = torch.zeros(P * 37, dtype=tdt, device='cuda:0')
                det.synth_fill_events(out, 77, begin, P, g, n_per_pos=37, plant_period=10000, plant_shift_milli=800, spread_milli=spread, outlier_permille=outl)
                assert np.array_equal(out.cpu().numpy().reshape(P, 37), ref[:, :37]), (name, g, spread, outl, 'stride')
                if outl == 10 and name == 'i16':        # the share of replaced reads is what the parameter says
                    base = H.synth_events_ref(77, begin, P, g, nmax, 10000, 800, spread, dtype=name)
                    assert 0.005 < np.mean(ref != base) < 0.015
    # both groups share the level; the float32 image is the stored 3-decimal value
    a = H.synth_events_ref(77, 0, 50, 0, 200, 0, 0, 100, 'i16'); b = H.synth_events_ref(77, 0, 50, 1, 200, 0, 0, 100, 'i16')
    assert np.all(np.abs(a.mean(axis=1) - b.mean(axis=1)) < 60) and a.mean(axis=1).std() > 1000
    f = H.synth_events_ref(77, 0, 50, 0, 200, 0, 0, 100, 'f32')
    assert np.array_equal(f, (a.astype(np.float64) / 1000.0).astype(np.float32))
    bad = L.make_params(device=0, memspace=L.MEM_DEVICE, dtype=L.DTYPE_F32)
    assert L.load().nmod_synth_fill_events(bad, 1, 0, 10, 0, 5, None, 0, 0, 9000, 0, 1) == -1
    assert L.load().nmod_synth_fill_events(bad, 1, 0, 10, 0, 0, None, 0, 0, 100, 0, 1) == -1
    assert L.load().nmod_synth_fill_events(bad, 1, 0, 10, 0, 5, None, 0, 0, 100, 1001, 1) == -1


@pytest.mark.parametrize('spread', [100, 200])
@pytest.mark.parametrize('dtype', ['f32', 'i16', 'f64'])
def test_event_like_batches_vs_oracle(nm, spread, dtype):
    """event-like rows (a level per position, reads spread 0.1 / 0.2 / 0.4 units around it, 3-decimal grid: most samples tie):
    200 v 200 fixed stride and ragged sizes around it, KS-only and all tests, against the oracle"""
    import nanomod_oracle as orc
    import oracle_c
    L = nm._lib
    P = 1200
    # ('f64': the rows as the reference holds them, k / 1000.0 — the float64 front end gives such positions k as keys)
    a = H.synth_events_ref(11, 4990, P, 0, 256, 5000, 800, spread, 'i16' if dtype == 'f64' else dtype)
    b = H.synth_events_ref(11, 4990, P, 1, 256, 5000, 800, spread, 'i16' if dtype == 'f64' else dtype)
    rid = np.zeros(P, np.int32)
    rng = np.random.default_rng(spread)
    for shape in ('stride', 'ragged'):
        if shape == 'stride':
            s0 = np.full(P, 200); s1 = np.full(P, 200)
        else:
            s0 = rng.integers(120, 257, P); s1 = rng.integers(120, 257, P)
            s0[:4] = (255, 256, 255, 256); s1[:4] = (255, 255, 256, 256)
        off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(s0)
        off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(s1)
        sig0 = np.concatenate([a[i, :s0[i]] for i in range(P)]); sig1 = np.concatenate([b[i, :s1[i]] for i in range(P)])
        exp = oracle_c.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, 'fisher', tests=7)
        if dtype == 'f64':
            sig0 = sig0.astype(np.float64) / 1000.0; sig1 = sig1.astype(np.float64) / 1000.0
        kw = dict(stride0=200, stride1=200) if shape == 'stride' else {}
        # ('f64': through the device's float64 front end — whole-number keys — not the host-side narrowing to int16)
        f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0
        got = nm.detect_host(sig0, None if shape == 'stride' else off0, sig1, None if shape == 'stride' else off1, rid, nb=2,
                             weights_dif=2.0, method='fisher', flags=f64, **kw)
        st = L.last_dispatch_stats()
        H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(sig0, off0, sig1, off1))
        assert np.array_equal(got['status'], exp['status'])
        # the counting forms took the batch (nmod_last_dispatch_stats): these numbers did not come from the sorting form
        assert st['positions'] == P and st['skipped'] == 0
        counted = st['rank_count'] + st['rank_count_wide']
        # (ragged: the few positions whose groups both fit 128 samples are the eight-positions-per-wave class's, which has no counting form)
        assert st['count_tried'] >= P - 12 and counted >= 0.95 * P and counted + st['count_rejected'] == st['count_tried'], (shape, st)
        assert st['rank_hist'] + st['rank_hist_wide'] + st['rank_pair'] == P - counted and st['ks_rank'] == 0
        # ... and the sorting forms alone (NMOD_FLAG_NO_COUNTING) give the same integers, so the same U, D and p bit for bit
        srt = nm.detect_host(sig0, None if shape == 'stride' else off0, sig1, None if shape == 'stride' else off1, rid, nb=2,
                             weights_dif=2.0, method='fisher', flags=L.FLAG_NO_COUNTING | f64, **kw)
        st0 = L.last_dispatch_stats()
        assert st0['count_tried'] == 0 and st0['rank_count'] + st0['rank_count_wide'] == 0 and st0['rank_hist'] == P
        for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p'):
            assert np.array_equal(srt[k], got[k], equal_nan=True), (shape, k)
        H.assert_close_p(srt['t_p'], got['t_p'], 1e-9, 't_p')
        got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='fisher', tests=L.TEST_KS, flags=f64)
        st = L.last_dispatch_stats()
        assert st['ks_rank'] == P and st['count_tried'] == 0            # (KS only at this coverage: ks_rank_kernel, eight / four positions per wave)
        assert np.array_equal(got['ks_d'], exp['ks_d'])
        H.assert_close_p(got['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
        H.assert_close_p(got['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    # the oracle's two implementations agree on such rows (the C one is what the big comparisons use)
    e2 = orc.detect_batch(sig0[:off0[40]], off0[:41], sig1[:off1[40]], off1[:41], rid[:40], 2, 2.0, orc.METHOD_FISHER)
    assert np.array_equal(e2['mwu_u'], exp['mwu_u'][:40]) and np.array_equal(e2['ks_d'], exp['ks_d'][:40])


@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_counting_form_edges(nm, dtype):
    """the counting form (rank_count.hpp) at its limits, inside a batch the probe accepts: every sample equal; two distinct values;
    a key range of exactly 2 047 (fits the window) and 2 048 (does not: the sorting form takes the position); groups of 255 / 4 / 256
    samples; one group constant; keys at the ends of the int16 domain; a float32 row with one sample off the grid"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(77)
    rows0, rows1 = [], []

    def add(a, b):
        rows0.append(np.asarray(a, dtype=np.int64)); rows1.append(np.asarray(b, dtype=np.int64))
    for _ in range(120):                                   # the bulk: ordinary event-like positions (the probe samples 64 of the batch)
        lev = int(rng.integers(-3000, 3000)); n0, n1 = int(rng.integers(130, 256)), int(rng.integers(130, 256))
        add(lev + np.rint(200 * rng.normal(0, 1, n0)), lev + np.rint(200 * rng.normal(0, 1, n1)))
    edge_at = len(rows0)
    add([500] * 200, [500] * 200)                          # every sample equal: MWU all identical
    add([10] * 100 + [11] * 100, [10] * 60 + [11] * 140)   # two distinct values
    add([0] + [2047] * 199, rng.integers(0, 2048, 200))    # range 2 047: the widest window
    add([0] + [2048] * 199, rng.integers(0, 2049, 200))    # range 2 048: beyond it
    add(rng.integers(-100, 100, 255), rng.integers(-100, 100, 4))
    add(rng.integers(-100, 100, 4), rng.integers(-100, 100, 255))
    add(rng.integers(-100, 100, 256), rng.integers(-100, 100, 200))       # 256 samples: not this form's
    add([7] * 200, rng.integers(-300, 300, 200))           # one group constant
    add(32767 - rng.integers(0, 900, 200), 32767 - rng.integers(0, 900, 200))
    add(-32767 + rng.integers(0, 900, 200), -32767 + rng.integers(0, 900, 200))
    add(rng.integers(-1000, 1000, 129), rng.integers(-1000, 1000, 131))
    add(np.arange(200), np.arange(200) + 1)                # no ties inside a group, every value shared but two
    P = len(rows0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum([len(r) for r in rows0])
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum([len(r) for r in rows1])
    k0 = np.concatenate(rows0).astype(np.int16); k1 = np.concatenate(rows1).astype(np.int16)
    rid = np.zeros(P, np.int32)
    if dtype == 'i16':
        s0, s1 = k0, k1
    elif dtype == 'f32':
        s0 = (k0.astype(np.float64) / 1000.0).astype(np.float32); s1 = (k1.astype(np.float64) / 1000.0).astype(np.float32)
        s0[off0[5] + 3] = np.nextafter(s0[off0[5] + 3], np.float32(9))          # one ordinary position with a sample off the grid
    else:
        s0 = k0.astype(np.float64) / 1000.0; s1 = k1.astype(np.float64) / 1000.0
    exp = orc.detect_batch(s0, off0, s1, off1, rid, 0, 2.0, orc.METHOD_FISHER)
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0           # (the device's float64 front end, not the host-side narrowing)
    got = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64)
    st = L.last_dispatch_stats()
    # exactly the positions the docstring lists are handed on: range 2 048, a group of 256, + the float32 sample off the grid; float64:
    # + the all-equal position (0.5 is float32-exact, so its keys are the values themselves, not k: not the counting form's)
    want_rejected = 2 if dtype == 'i16' else 3
    assert st['count_tried'] == P and st['count_rejected'] == want_rejected and st['rank_count'] + st['rank_count_wide'] == P - want_rejected, st
    assert st['rank_count'] >= 120 + 7 - (0 if dtype == 'i16' else 1)
    ident = (exp['status'] & L.STATUS_MWU_ALL_IDENTICAL) != 0
    assert ident[edge_at] and ident.sum() == 1 and np.array_equal(got['status'], exp['status'])
    assert np.isnan(got['mwu_u'][edge_at]) and got['ks_d'][edge_at] == 0.0 and got['ks_p'][edge_at] == 1.0
    for d in (got, exp):
        d['mwu_u'][ident] = 0.0; d['mwu_p'][ident] = 0.0
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    # the same positions alone (a batch of twelve edge cases: whatever the probe decides, the numbers are the same)
    lo = edge_at
    sub = nm.detect_host(s0[off0[lo]:], off0[lo:] - off0[lo], s1[off1[lo]:], off1[lo:] - off1[lo], rid[lo:], nb=0, weights_dif=2.0, method='fisher', flags=f64)
    for k in ('mwu_p', 'ks_d', 'ks_p'):                      # (exact integers into the same K2: bit-equal whichever form produced them)
        assert np.array_equal(sub[k][1:], got[k][lo + 1:], equal_nan=True), k
    H.assert_close_p(sub['t_p'][1:], got['t_p'][lo + 1:], 1e-9, 't_p')


@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_counting_form_outliers_vs_oracle(nm, dtype):
    """rank_count.hpp's outlier path (round 6): samples further than 1 024 milli-units from the mean of the position's keys leave the
    table for a list and come back under remapped keys (order and ties kept).  Positions with 1, 2, 15, 16 and 17 such samples
    below / above / on both sides, in group 1 / group 2 / both; the same outlier value several times and in both groups; outliers at
    the ends of the int16 domain; an outlier as a group's first sample; two clusters 3 units apart (every sample an outlier:
    handed on); then random contamination at 1 / 10 / 30 per mille.  Every number against the oracle; the form keeps every position
    with at most 16 tail samples (nmod_last_dispatch_stats); the sorting form alone gives the same integers"""
    import oracle_c
    L = nm._lib
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0
    rng = np.random.default_rng(zlib.crc32(('cnt-outliers' + dtype).encode()))
    rows0, rows1, kept = [], [], []

    def add(a, b, keep=True):
        rows0.append(np.asarray(a, dtype=np.int64)); rows1.append(np.asarray(b, dtype=np.int64)); kept.append(keep)

    def ev(n, lev=0, s=150):
        return lev + np.clip(np.rint(s * rng.normal(0, 1, n)), -600, 600).astype(np.int64)

    def out(n, side, lev=0):
        lo = rng.integers(-5000, lev - 1300, n) if lev - 1300 > -5000 else np.full(n, -5000)
        hi = rng.integers(lev + 1300, 5001, n) if lev + 1300 < 5000 else np.full(n, 5000)
        return lo if side < 0 else hi if side > 0 else np.where(rng.random(n) < 0.5, lo, hi)
    for _ in range(700):                                     # the bulk: ordinary event-like positions (the probe wants 7 of 8 sampled ones to fit)
        lev = int(rng.integers(-3000, 3000)); add(ev(int(rng.integers(130, 256)), lev), ev(int(rng.integers(130, 256)), lev))
    edge_at = len(rows0)
    for n_out in (1, 2, 15, 16, 17):
        for side in (-1, 1, 0):
            lev = int(rng.integers(-2500, 2500))
            add(np.r_[ev(200 - n_out, lev), out(n_out, side, lev)], ev(180, lev), n_out <= 16)            # in group 1 (the group that is looked up)
            add(ev(180, lev), np.r_[out(n_out, side, lev), ev(200 - n_out, lev)], n_out <= 16)            # in group 2, first in its row
            if n_out <= 8:
                add(np.r_[out(n_out, side, lev), ev(150, lev)], np.r_[ev(150, lev), out(n_out, -side, lev)])   # both groups (first sample of group 1)
    add(np.r_[ev(190), [4000] * 6, [-4000] * 2], np.r_[ev(190), [4000] * 3, [-4000] * 4, 4001])          # the same outlier value several times, in both groups
    add(np.r_[ev(196), [32767, 32767, -32768, -32767]], np.r_[ev(198), [32767, -32768]])                  # the ends of the int16 domain
    add(np.r_[ev(199, 31000), [-32768]], ev(200, 31000)); add(np.r_[ev(199, -31000), [32767]], ev(200, -31000))   # a window clamped at the domain's end
    add(np.r_[ev(100, -1500, 80), ev(100, 1500, 80)], np.r_[ev(100, -1500, 80), ev(100, 1500, 80)], False)        # two clusters: every sample is far from the mean
    add(np.r_[ev(250), out(5, 0)], np.r_[ev(250), out(5, 0)])                                              # 255 samples per group with outliers
    add(np.r_[[0] * 100, [1] * 100, [3000] * 4], np.r_[[0] * 60, [1] * 140, [3000] * 2, [-3000] * 2])      # heavy ties inside the window, tied outliers
    for frac, npos_f in ((0.001, 60), (0.01, 60), (0.03, 30)):
        for _ in range(npos_f):
            lev = int(rng.integers(-3000, 3000)); a, b = ev(int(rng.integers(130, 256)), lev), ev(int(rng.integers(130, 256)), lev)
            n_o = 0
            for v in (a, b):
                hit = rng.random(len(v)) < frac
                v[hit] = rng.integers(-5000, 5001, int(hit.sum())); n_o += int((np.abs(v - lev) > 1300).sum())
            add(a, b, None)                                  # (kept unless more than 16 of them ended up far away: decided by the library's own centre)
    P = len(rows0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum([len(r) for r in rows0])
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum([len(r) for r in rows1])
    k0 = np.concatenate(rows0).astype(np.int16); k1 = np.concatenate(rows1).astype(np.int16)
    rid = np.zeros(P, np.int32)
    s0, s1 = _as_dtype(k0, dtype), _as_dtype(k1, dtype)
    exp = oracle_c.detect_batch(s0 if dtype == 'f32' else k0, off0, s1 if dtype == 'f32' else k1, off1, rid, 0, 2.0, 'fisher', tests=7)
    assert L.load().nmod_host_pipeline_config(1 << 30, 0, 0, 0) == 0          # (one chunk: one probe over the whole batch)
    try:
        got = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64)
        st = L.last_dispatch_stats()
        srt = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64 | L.FLAG_NO_COUNTING)
    finally:
        assert L.load().nmod_host_pipeline_config(0, 0, 0, 0) == 0
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    assert np.array_equal(got['status'], exp['status'])
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p'):
        assert np.array_equal(srt[k], got[k]), k
    H.assert_close_p(srt['t_p'], got['t_p'], 1e-9, 't_p')
    must_reject = kept.count(False)
    assert st['skipped'] == 0 and st['count_tried'] == P and st['rank_count'] + st['count_rejected'] == P, st
    assert must_reject <= st['count_rejected'] <= must_reject + 12 and st['rank_count'] >= P - must_reject - 12, (must_reject, st)


def _event_rows(rng, sizes0, sizes1, spread, shift_every=5):
    """int16 event-like rows: a level per position, reads spread around it, group 2 shifted at every shift_every-th position"""
    P = len(sizes0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(sizes0)
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(sizes1)
    lev = rng.integers(-3000, 3001, P)
    k0 = np.repeat(lev, sizes0) + np.rint(spread * rng.normal(0, 1, int(off0[-1]))).astype(np.int64)
    sh = np.where(np.arange(P) % shift_every == 0, rng.integers(-300, 301, P), 0)
    k1 = np.repeat(lev + sh, sizes1) + np.rint(spread * rng.normal(0, 1, int(off1[-1]))).astype(np.int64)
    return np.clip(k0, -32767, 32767).astype(np.int16), off0, np.clip(k1, -32767, 32767).astype(np.int16), off1


def _as_dtype(k, dtype):
    if dtype == 'i16':
        return k
    x = k.astype(np.float64) / 1000.0
    return x.astype(np.float32) if dtype == 'f32' else x


@pytest.mark.parametrize('shape', ['skew', 'skew_rev', '500v500', '1000v1000', 'mixed', 'stride600'])
@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_count_wide_event_like_vs_oracle(nm, shape, dtype):
    """the counting form for any coverage (rank_count_wide.hpp) against the oracle on event-like rows: skewed coverage either
    way round (configs[4]: ~1 131 v ~57), both groups in the 512 and in the 1 024 class, every class at once, a fixed-stride
    batch; all tests and KS-only (the latter never takes the form)"""
    import oracle_c
    L = nm._lib
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0           # ('f64': the device's float64 front end — whole-number keys —, not the host-side narrowing)
    rng = np.random.default_rng(zlib.crc32((shape + dtype).encode()))
    P = 700
    if shape in ('skew', 'skew_rev'):
        a = rng.integers(600, 2400, P); b = rng.integers(20, 120, P)
        a[:6] = (2048, 2049, 4095, 4095, 2047, 1025); b[:6] = (64, 65, 256, 1, 128, 129)
        if shape == 'skew_rev':
            a, b = b, a
    elif shape == '500v500':
        a = rng.integers(300, 513, P); b = rng.integers(300, 513, P); a[:3] = (512, 257, 512); b[:3] = (512, 512, 257)
    elif shape == '1000v1000':
        P = 300
        a = rng.integers(600, 1025, P); b = rng.integers(600, 1025, P); a[:3] = (1024, 513, 1024); b[:3] = (1024, 1024, 513)
    elif shape == 'mixed':
        a = np.exp(rng.uniform(np.log(2), np.log(4000), P)).astype(np.int64); b = np.exp(rng.uniform(np.log(2), np.log(4000), P)).astype(np.int64)
    else:
        a = np.full(P, 600); b = np.full(P, 90)
    k0, off0, k1, off1 = _event_rows(rng, a, b, 200 if shape != 'mixed' else 150)
    rid = np.zeros(P, np.int32)
    s0, s1 = _as_dtype(k0, dtype), _as_dtype(k1, dtype)
    exp = oracle_c.detect_batch(s0 if dtype == 'f32' else k0, off0, s1 if dtype == 'f32' else k1, off1, rid, 2, 2.0, 'stouffer', tests=7)
    kw = dict(stride0=600, stride1=90) if shape == 'stride600' else {}
    got = nm.detect_host(s0, None if kw else off0, s1, None if kw else off1, rid, nb=2, weights_dif=2.0, method='stouffer', **kw, flags=f64)
    st = L.last_dispatch_stats()
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    assert np.array_equal(got['status'], exp['status'])
    # the form took the batch ('mixed': positions whose smaller group exceeds 1 024 samples, or whose groups fit 64 / 128 / 256 both, are not its)
    share = st['rank_count_wide'] / P
    assert st['skipped'] == 0 and share >= (0.45 if shape == 'mixed' else 0.95), (shape, st)
    # ... and without it (NMOD_FLAG_NO_COUNT_WIDE) the sorting forms give the same integers: U, D, p bit for bit
    srt = nm.detect_host(s0, None if kw else off0, s1, None if kw else off1, rid, nb=2, weights_dif=2.0, method='stouffer', flags=f64 | L.FLAG_NO_COUNT_WIDE, **kw)
    assert L.last_dispatch_stats()['rank_count_wide'] == 0
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p'):
        assert np.array_equal(srt[k], got[k], equal_nan=True), (shape, k)
    H.assert_close_p(srt['t_p'], got['t_p'], 1e-9, 't_p')
    got = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64)
    st = L.last_dispatch_stats()
    if shape in ('skew', 'skew_rev', '500v500', '1000v1000', 'stride600'):       # KS only: positions whose larger group holds >= 320 samples
        assert st['rank_count_wide'] >= 0.8 * P and st['ks_rank'] + st['big'] == P - st['rank_count_wide'], (shape, st)
    assert np.array_equal(got['ks_d'], exp['ks_d'])
    H.assert_close_p(got['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    srt = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64 | L.FLAG_NO_COUNTING)
    assert L.last_dispatch_stats()['rank_count_wide'] == 0
    for k in ('ks_d', 'ks_p', 'comb_st', 'comb_p'):
        assert np.array_equal(srt[k], got[k], equal_nan=True), (shape, k)


@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_count_wide_edges(nm, dtype):
    """rank_count_wide.hpp at its limits, inside classes its probe accepts: every sample equal (300 v 300: the tie limit sends it
    back); one value 254 / 255 / 256 times in the larger group, 254 / 255 in the smaller; the smaller group's range 4 095 and
    4 096; a sample of the larger group just inside / outside the window; the larger group of 4 095 / 4 096; a group constant;
    keys at the ends of the int16 domain; a float32 sample off the grid in either group; a smaller group of one"""
    import oracle_c
    L = nm._lib
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0           # ('f64': the device's float64 front end — whole-number keys —, not the host-side narrowing)
    rng = np.random.default_rng(91)
    rows0, rows1 = [], []

    def add(a, b):
        rows0.append(np.asarray(a, dtype=np.int64)); rows1.append(np.asarray(b, dtype=np.int64))

    def ev(n, lev=0, s=200):
        return lev + np.rint(s * rng.normal(0, 1, n)).astype(np.int64)
    for _ in range(150):                                   # the bulk of two classes: (300..512) v (300..512) and (700..1000) v (30..64)
        lev = int(rng.integers(-3000, 3000))
        add(ev(int(rng.integers(300, 513)), lev), ev(int(rng.integers(300, 513)), lev))
        lev = int(rng.integers(-3000, 3000))
        add(ev(int(rng.integers(700, 1000)), lev), ev(int(rng.integers(30, 65)), lev))
    edge_at = len(rows0)
    add([500] * 300, [500] * 300)                                              # every sample equal
    add(np.r_[[40] * 254, ev(200)], ev(300)); add(np.r_[[40] * 255, ev(200)], ev(300)); add(np.r_[[40] * 256, ev(200)], ev(300))
    add(ev(400), np.r_[[-7] * 254, ev(100)]); add(ev(400), np.r_[[-7] * 255, ev(100)])
    add(np.r_[[40] * 200, ev(200)], np.r_[[40] * 200, ev(200)])                # 400 copies over both groups
    add(ev(800), np.r_[0, 4095, rng.integers(0, 4096, 50)]); add(ev(800), np.r_[0, 4096, rng.integers(0, 4096, 50)])
    add(np.r_[ev(799, 0, 100), 1500], ev(40, 0, 100)); add(np.r_[ev(799, 0, 100), 3000], ev(40, 0, 100)); add(np.r_[ev(799, 0, 100), -3000], ev(40, 0, 100))
    add(ev(4095), ev(60)); add(ev(60), ev(4095)); add(ev(4096), ev(60)); add(ev(60), ev(4096))
    add([7] * 900, ev(50)); add(ev(900), [7] * 50); add(ev(900), [7])
    add(32767 - rng.integers(0, 900, 900), 32767 - rng.integers(0, 900, 40)); add(-32767 + rng.integers(0, 900, 900), -32767 + rng.integers(0, 900, 40))
    add(np.arange(500), np.arange(500) + 1)
    off_grid_at = len(rows0)
    add(ev(900), ev(50)); add(ev(900), ev(50))
    P = len(rows0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum([len(r) for r in rows0])
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum([len(r) for r in rows1])
    k0 = np.concatenate(rows0).astype(np.int16); k1 = np.concatenate(rows1).astype(np.int16)
    rid = np.zeros(P, np.int32)
    s0, s1 = _as_dtype(k0, dtype), _as_dtype(k1, dtype)
    if dtype == 'f32':
        s0[off0[off_grid_at] + 700] = np.nextafter(s0[off0[off_grid_at] + 700], np.float32(9))
        s1[off1[off_grid_at + 1] + 3] = np.nextafter(s1[off1[off_grid_at + 1] + 3], np.float32(-9))
    exp = oracle_c.detect_batch(s0 if dtype == 'f32' else k0, off0, s1 if dtype == 'f32' else k1, off1, rid, 0, 2.0, 'fisher', tests=7)
    got = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64)
    st = L.last_dispatch_stats()
    # the bulk (300 positions) is the form's; of the edge positions it hands on at most all
    assert st['rank_count_wide'] >= 300 and st['count_rejected'] <= P - 300 and st['count_tried'] >= 300 and st['skipped'] == 0, st
    ident = (exp['status'] & L.STATUS_MWU_ALL_IDENTICAL) != 0
    assert ident[edge_at] and ident.sum() == 1 and np.array_equal(got['status'], exp['status'])
    for d in (got, exp):
        d['mwu_u'][ident] = 0.0; d['mwu_p'][ident] = 0.0
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    srt = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64 | L.FLAG_NO_COUNTING)
    assert L.last_dispatch_stats()['count_tried'] == 0
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p'):                  # the sorting forms alone: the same integers, so U, D and p bit for bit
        assert np.array_equal(srt[k][~ident], got[k][~ident]), k
    lo = edge_at                                               # the edge positions alone: whatever the probes decide, the same numbers
    sub = nm.detect_host(s0[off0[lo]:], off0[lo:] - off0[lo], s1[off1[lo]:], off1[lo:] - off1[lo], rid[lo:], nb=0, weights_dif=2.0, method='fisher', flags=f64)
    for k in ('mwu_p', 'ks_d', 'ks_p'):
        assert np.array_equal(sub[k][1:], got[k][lo + 1:], equal_nan=True), k
    H.assert_close_p(sub['t_p'][1:], got['t_p'][lo + 1:], 1e-9, 't_p')
    # KS only (the form without the tie term; positions whose larger group is below its size stay with ks_rank_kernel), D bit for bit
    # and as the exact rational (NMOD_FLAG_KS_RATIONAL_D: within 2 ulp)
    ks = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64)
    assert np.array_equal(ks['ks_d'], exp['ks_d'])
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    kr = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64 | L.FLAG_KS_RATIONAL_D)
    assert np.all(np.abs(kr['ks_d'] - exp['ks_d']) <= 4.5e-16)


@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_count_wide_outliers_vs_oracle(nm, dtype):
    """rank_count_wide.hpp's tail list (samples outside the 2 048-value window: mis-segmented reads anywhere in +-5 units,
    myRefBaseSignalAnnotation.py:251-259): positions with 1, 2, 63, 64 and 65 such samples below / above the window, in the smaller /
    the larger / both groups, repeated outlier values (ties among the tail), outliers at the ends of the int16 domain, an outlier
    as the row's first sample; then random contamination at 1 / 10 / 50 per mille.  Every number against the oracle, all tests and
    KS only; the form keeps every position with at most 64 tail samples (nmod_last_dispatch_stats)"""
    import oracle_c
    L = nm._lib
    rng = np.random.default_rng(zlib.crc32(('cw-outliers' + dtype).encode()))
    rows0, rows1 = [], []

    def add(a, b):
        rows0.append(np.asarray(a, dtype=np.int64)); rows1.append(np.asarray(b, dtype=np.int64))

    def ev(n, lev=0, s=150):
        return lev + np.clip(np.rint(s * rng.normal(0, 1, n)), -700, 700).astype(np.int64)

    def out(n, side, lev=0):                                 # n outliers on one side (or either) of a window around lev, in the clip range
        lo = rng.integers(-5000, lev - 1100, n) if lev - 1100 > -5000 else np.full(n, -5000)
        hi = rng.integers(lev + 1100, 5001, n) if lev + 1100 < 5000 else np.full(n, 5000)
        return lo if side < 0 else hi if side > 0 else np.where(rng.random(n) < 0.5, lo, hi)
    for _ in range(500):                                     # the bulk (the probes look at 64 positions per class and want 7 of 8 to fit): skewed and 400 v 400
        lev = int(rng.integers(-3000, 3000)); add(ev(int(rng.integers(700, 1400)), lev), ev(int(rng.integers(30, 65)), lev))
        lev = int(rng.integers(-3000, 3000)); add(ev(int(rng.integers(300, 513)), lev), ev(int(rng.integers(300, 513)), lev))
    edge_at = len(rows0)
    kept = []                                                # edge positions the form must keep / hand on
    for big, small in ((900, 50), (400, 400)):
        for n_out in (1, 2, 63, 64, 65):
            for side in (-1, 1, 0):
                lev = int(rng.integers(-2500, 2500))
                add(np.r_[ev(big - n_out, lev), out(n_out, side, lev)], ev(small, lev)); kept.append(n_out <= 64)       # in the larger group
                if n_out < small - 8:
                    add(ev(big, lev), np.r_[out(n_out, side, lev), ev(small - n_out, lev)]); kept.append(True)           # in the smaller group, first in its row
        lev = 0
        add(np.r_[ev(big - 40, lev), out(20, -1, lev), out(20, 1, lev)], np.r_[ev(small - 6, lev), out(3, -1, lev), out(3, 1, lev)]); kept.append(True)   # both groups, both sides
        add(np.r_[ev(big - 30, lev), [4000] * 30], np.r_[ev(small - 4, lev), [4000] * 2, [-4000] * 2]); kept.append(True)           # the same value 32 times over both groups
        add(np.r_[ev(big - 4, lev), [32767, 32767, -32768, -32767]], np.r_[ev(small - 2, lev), [32767, -32768]]); kept.append(True)  # the ends of the int16 domain
        add(np.r_[ev(big - 1, 30000), [-32768]], ev(small, 30000)); kept.append(True)                                               # a window clamped at the domain's end
        add(np.r_[ev(big - 1, -30000), [32767]], ev(small, -30000)); kept.append(True)
    # the other register counts of the smaller group (128 / 256 / 1 024 samples: RS = 2, 4, 16 — the last one keeps no addresses in
    # registers): a class each, clean positions and positions with 1 ... 20 outliers in either group (few enough of them for the KS-only
    # probe too: it hands a class back above 4 - 6 tail samples per 1 000)
    for big, small in ((1000, 100), (1000, 200), (1000, 1000)):
        for i in range(48):
            lev = int(rng.integers(-2500, 2500)); n_out = (0 if i < 40 else int(rng.integers(1, 21)))
            a, b = ev(big, lev), ev(small, lev)
            if n_out:
                a[rng.choice(big, n_out, replace=False)] = out(n_out, 0, lev)
                nb_ = min(n_out, small // 8)
                b[rng.choice(small, nb_, replace=False)] = out(nb_, 0, lev)
            add(a, b); kept.append(True)
    for frac, npos_f in ((0.001, 40), (0.01, 40), (0.05, 16)):   # random contamination of both groups
        for _ in range(npos_f):
            lev = int(rng.integers(-3000, 3000)); n0, n1 = int(rng.integers(700, 1400)), int(rng.integers(30, 65))
            a, b = ev(n0, lev), ev(n1, lev)
            for v in (a, b):
                hit = rng.random(len(v)) < frac
                v[hit] = rng.integers(-5000, 5001, int(hit.sum()))
            add(a, b)
    P = len(rows0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum([len(r) for r in rows0])
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum([len(r) for r in rows1])
    k0 = np.concatenate(rows0).astype(np.int16); k1 = np.concatenate(rows1).astype(np.int16)
    rid = np.zeros(P, np.int32)
    s0, s1 = _as_dtype(k0, dtype), _as_dtype(k1, dtype)
    exp = oracle_c.detect_batch(s0 if dtype == 'f32' else k0, off0, s1 if dtype == 'f32' else k1, off1, rid, 0, 2.0, 'fisher', tests=7)
    # (one chunk: the host-resident entry probes every chunk's classes on their own, and a chunk made of the edge positions alone —
    # they sit together — would keep its classes on the sorting forms)
    assert L.load().nmod_host_pipeline_config(1 << 30, 0, 0, 0) == 0
    try:
        _count_wide_outlier_checks(nm, dtype, s0, off0, s1, off1, rid, exp, kept, P)
    finally:
        assert L.load().nmod_host_pipeline_config(0, 0, 0, 0) == 0


def _count_wide_outlier_checks(nm, dtype, s0, off0, s1, off1, rid, exp, kept, P):
    L = nm._lib
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0           # ('f64': the device's float64 front end — whole-number keys —, not the host-side narrowing)
    got = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64)
    st = L.last_dispatch_stats()
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    assert np.array_equal(got['status'], exp['status'])
    n_rej = kept.count(False)
    assert st['skipped'] == 0 and st['count_tried'] == P, st
    # the positions with 65 outliers are handed on; of the randomly contaminated ones at 50 per mille (~55 outliers of ~1 100) a few more
    assert n_rej <= st['count_rejected'] <= n_rej + 25 and st['rank_count_wide'] == P - st['count_rejected'], (n_rej, st)
    # the sorting forms alone: the same integers
    srt = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64 | L.FLAG_NO_COUNTING)
    assert L.last_dispatch_stats()['count_tried'] == 0
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p'):
        assert np.array_equal(srt[k], got[k]), k
    H.assert_close_p(srt['t_p'], got['t_p'], 1e-9, 't_p')
    # KS only (no tie term, no moments), D bit for bit; the exact-rational flag
    ks = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64)
    st = L.last_dispatch_stats()
    assert np.array_equal(ks['ks_d'], exp['ks_d']) and st['rank_count_wide'] >= P - n_rej - 25
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    kr = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64 | L.FLAG_KS_RATIONAL_D)
    assert np.all(np.abs(kr['ks_d'] - exp['ks_d']) <= 4.5e-16)


@pytest.mark.parametrize('dtype', ['i16', 'f32', 'f64'])
def test_value_domain_form_vs_oracle(nm, dtype):
    """rank_count_value.hpp (round 6): all tests on event-like positions whose groups BOTH hold 1 025 ... 2 048 samples — counted by value,
    every statistic from the table's values in order.  Clean positions, 1 / 2 equal / 64 / 65 / 128 / 129 samples outside the window (below,
    above, both; either group), ties inside and across the groups, groups a unit apart (D = 1), constant groups, a window clamped at
    the end of the int16 domain, random contamination at 1 / 10 per mille — every number against the oracle; the form keeps every
    position with at most 128 outside samples (nmod_last_dispatch_stats) and the sorting form (rank_pair_kernel) gives the same integers"""
    import oracle_c
    L = nm._lib
    rng = np.random.default_rng(zlib.crc32(('value-domain' + dtype).encode()))
    rows0, rows1, kept = [], [], []

    def add(a, b, keep=True):
        rows0.append(np.asarray(a, dtype=np.int64)); rows1.append(np.asarray(b, dtype=np.int64)); kept.append(keep)

    def ev(n, lev=0, s=150):
        return lev + np.clip(np.rint(s * rng.normal(0, 1, n)), -700, 700).astype(np.int64)

    def out(n, side, lev=0):
        lo = rng.integers(-5000, lev - 1100, n) if lev - 1100 > -5000 else np.full(n, -5000)
        hi = rng.integers(lev + 1100, 5001, n) if lev + 1100 < 5000 else np.full(n, 5000)
        return lo if side < 0 else hi if side > 0 else np.where(rng.random(n) < 0.5, lo, hi)

    def sz():
        return int(rng.integers(1025, 2049))
    for _ in range(100):                                     # the bulk: the probe wants 7 of 8 sampled positions to fit
        lev = int(rng.integers(-3000, 3000)); add(ev(sz(), lev), ev(sz(), lev + int(rng.choice([0, 0, 60, -200]))))
    for n_out in (1, 2, 64, 65, 128, 129):
        for side in (-1, 1, 0):
            for grp in (0, 1):
                lev = int(rng.integers(-2500, 2500)); a, b = ev(sz(), lev), ev(sz(), lev)
                tgt = a if grp == 0 else b
                far = out(n_out, side, lev)
                if n_out == 2:
                    far[1] = far[0]
                tgt[64 + rng.choice(len(tgt) - 64, n_out, replace=False)] = far     # (not among the first 64: the centre's sample)
                add(a, b, n_out <= 128)
    lev = 0
    a, b = ev(2048, lev), ev(1025, lev); a[100:130] = 4000; b[200:202] = 4000; b[300:302] = -4000; add(a, b)      # one value 32 times over both groups, outside
    a, b = ev(1500, lev), ev(1500, lev); a[:] = 250; b[:] = 250; add(a, b)                                        # every sample equal: U / p NaN
    a, b = ev(1500, lev), ev(1600, lev); a[:] = 100; b[:] = 900; a[0] = 101; b[0] = 899; add(a, b)                # (all but) constant groups apart: D = 1 (exactly constant ones have no t to compare: 0 / 0 in exact arithmetic, rounding noise in the reference)
    a, b = ev(1300, 0, 60), ev(1300, 1000, 60); add(a, b)                                                          # a unit apart: both inside one window, D = 1
    a, b = ev(2048, lev, 20), ev(2048, lev, 20); add(a, b)                                                         # narrow: ~150 copies per value
    add(ev(1100, 31500), ev(1100, 31500)); add(ev(1100, -31500), ev(1100, -31500))                                # the window clamped at the domain's ends
    a, b = ev(1200, 30000), ev(1200, 30000); a[70] = -32768; b[80] = 32767; add(a, b, True if dtype == 'i16' else None)   # (-32.768 is not a float32 / float64 key of the form: |k| <= 32 767)
    a, b = ev(1400, lev), ev(1400, lev); a[:64] = out(64, 0, lev); add(a, b, None)                                       # the first 64 of a group all outliers: the centre from the other group's
    for frac, npos_f in ((0.001, 24), (0.01, 24)):
        for _ in range(npos_f):
            lev = int(rng.integers(-3000, 3000)); a, b = ev(sz(), lev), ev(sz(), lev)
            for v in (a, b):
                hit = rng.random(len(v)) < frac
                v[hit] = rng.integers(-5000, 5001, int(hit.sum()))
            add(a, b, None)
    P = len(rows0)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum([len(r) for r in rows0])
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum([len(r) for r in rows1])
    k0 = np.concatenate(rows0).astype(np.int16); k1 = np.concatenate(rows1).astype(np.int16)
    rid = np.zeros(P, np.int32)
    s0, s1 = _as_dtype(k0, dtype), _as_dtype(k1, dtype)
    exp = oracle_c.detect_batch(s0 if dtype == 'f32' else k0, off0, s1 if dtype == 'f32' else k1, off1, rid, 0, 2.0, 'fisher', tests=7)
    f64 = L.FLAG_NO_HOST_NARROW if dtype == 'f64' else 0
    assert L.load().nmod_host_pipeline_config(1 << 30, 0, 0, 0) == 0      # (one chunk: one probe over the class)
    try:
        got = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64)
        st = L.last_dispatch_stats()
        srt = nm.detect_host(s0, off0, s1, off1, rid, nb=0, weights_dif=2.0, method='fisher', flags=f64 | L.FLAG_NO_COUNTING)
        st_srt = L.last_dispatch_stats()
    finally:
        assert L.load().nmod_host_pipeline_config(0, 0, 0, 0) == 0
    ident = (exp['status'] & 1) != 0
    assert np.array_equal(np.isnan(got['mwu_u']), ident) and np.array_equal(np.isnan(srt['mwu_u']), ident)
    for r in (got, srt):
        r['mwu_u'][ident] = 0.0
    exp['mwu_u'][ident] = 0.0
    H.compare_outputs(got, exp, True, t_abs=H.t_abs_gate(s0, off0, s1, off1))
    assert np.array_equal(got['status'], exp['status'])
    n_rej = kept.count(False)
    assert st['skipped'] == 0 and st['count_tried'] == P and st_srt['count_tried'] == 0 and st_srt['rank_pair'] == P, (st, st_srt)
    assert n_rej <= st['count_rejected'] <= n_rej + 6 and st['rank_count_wide'] == P - st['count_rejected'] and st['rank_pair'] == st['count_rejected'], (n_rej, st)
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p'):
        assert np.array_equal(srt[k], got[k], equal_nan=True), k
    H.assert_close_p(srt['t_p'], got['t_p'], 1e-9, 't_p')
    # KS only: the form's instance without the sums; D bit for bit, the sorting form (ks_rank_kernel) gives the same
    assert L.load().nmod_host_pipeline_config(1 << 30, 0, 0, 0) == 0
    try:
        ks = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64)
        st_ks = L.last_dispatch_stats()
        ks_srt = nm.detect_host(s0, off0, s1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=f64 | L.FLAG_NO_COUNTING)
    finally:
        assert L.load().nmod_host_pipeline_config(0, 0, 0, 0) == 0
    assert np.array_equal(ks['ks_d'], exp['ks_d']) and np.array_equal(ks_srt['ks_d'], ks['ks_d']) and np.array_equal(ks_srt['ks_p'], ks['ks_p'])
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    assert st_ks['count_tried'] == P and st_ks['rank_count_wide'] >= P - n_rej - 6 and st_ks['rank_count_wide'] + st_ks['ks_rank'] == P, st_ks
    import ctypes as C
    buf = C.create_string_buffer(160)
    prm = L.make_params(method=L.METHOD_FISHER, nb=1)
    assert L.load().nmod_describe_dispatch(C.byref(prm), 1500, 2048, buf, 160) == 0
    assert buf.value == b'rank_count_value_kernel<f32> (event-like rows) | rank_pair_kernel<32,32,f32>', buf.value


# general (64 lanes per group) and packed (two positions per wave) kernels, every capacity class
@pytest.mark.parametrize('sizes', [(5, 64, 5, 64), (65, 128, 3, 30), (65, 128, 65, 128), (129, 256, 129, 256),
                                   (100, 128, 129, 220), (257, 512, 257, 512), (300, 512, 20, 256),
                                   (513, 1024, 513, 1024), (513, 1024, 40, 70), (1025, 2048, 1025, 2048),
                                   (3, 2048, 3, 2048)])
@pytest.mark.parametrize('grid', [False, True])
def test_random_batches_vs_oracle(nm, sizes, grid):
    """every size class (and mixed classes in one batch), continuous and tie-heavy data"""
    import nanomod_oracle as orc
    rng = np.random.default_rng(zlib.crc32(repr((sizes, grid)).encode()))
    npos = 60 if sizes[1] > 1024 else 150
    sig0, off0, sig1, off1, rid = _random_batch(rng, npos, *sizes, grid=grid)
    for method in ('stouffer', 'fisher'):
        got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method=method)
        st = nm._lib.last_dispatch_stats()
        exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0,
                               orc.METHOD_STOUFFER if method == 'stouffer' else orc.METHOD_FISHER)
        H.compare_outputs(got, exp, True)
        assert np.array_equal(got['status'], exp['status'])
        assert st['positions'] == npos and st['skipped'] == 0 and st['ks_rank'] == 0
        if not grid:                 # continuous rows: the probes keep the counting forms out
            assert st['count_tried'] == 0 and st['rank_count'] + st['rank_count_wide'] == 0, st


@pytest.mark.parametrize('sizes', [(5, 64, 5, 64), (65, 128, 3, 30), (129, 256, 129, 256), (200, 200, 200, 200),
                                   (257, 512, 100, 300), (513, 1024, 600, 1024), (1025, 2048, 1025, 2048),
                                   (900, 1100, 30, 70), (3, 2048, 3, 2048), (64, 64, 256, 256), (1, 3, 1, 3)])
@pytest.mark.parametrize('grid', [False, True])
def test_ks_only_mode_vs_oracle(nm, sizes, grid):
    """tests mask = KS (the benchmark configuration): the sort-one-group + rank-the-other kernel, every
    capacity class of the smaller group, continuous and tie-heavy data, ragged batches"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(zlib.crc32(repr((sizes, grid, 'ks')).encode()))   # (hash() of a str changes per process)
    npos = 60 if sizes[1] > 1024 else 203
    sig0, off0, sig1, off1, rid = _random_batch(rng, npos, *sizes, grid=grid)
    if grid:   # make some positions extremely tie-heavy, including identical groups
        for i in range(0, npos, 9):
            sig0[off0[i]:off0[i + 1]] = np.round(sig0[off0[i]:off0[i + 1]], 0)
            sig1[off1[i]:off1[i + 1]] = np.round(sig1[off1[i]:off1[i + 1]], 0)
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    ks = [orc.ks_2samp(sig0[off0[i]:off0[i + 1]], sig1[off1[i]:off1[i + 1]]) for i in range(npos)]
    exp_d = np.array([k[0] for k in ks]); exp_p = np.maximum(np.array([k[1] for k in ks]), orc.DBL_MIN)
    H.assert_close_stat(got['ks_d'], exp_d, 0, 0.0, 'ks_d')
    H.assert_close_p(got['ks_p'], exp_p, 1e-9, 'ks_p')
    st, pv = orc.combine_track(exp_d, exp_p, rid, 2, 2.0, orc.METHOD_STOUFFER)
    H.assert_close_p(got['comb_p'], pv, 1e-9, 'comb_p')
    # and the exact integer numerator agrees with the all-tests kernels' D (bit-exact reference form)
    full = nm.detect_host(sig0, off0, sig1, off1, rid, method='ks')
    assert np.max(np.abs(full['ks_d'] - got['ks_d'])) <= 0.0


def test_int16_milli_path_matches_float_path(nm):
    import nanomod_oracle as orc
    rng = np.random.default_rng(99)
    sig0, off0, sig1, off1, rid = _random_batch(rng, 200, 5, 300, 5, 300)
    k0 = np.rint(sig0.astype(np.float64) * 1000).astype(np.int16)
    k1 = np.rint(sig1.astype(np.float64) * 1000).astype(np.int16)
    got = nm.detect_host(k0, off0, k1, off1, rid, want_mstd=True)
    exp = orc.detect_batch(k0 / 1000.0, off0, k1 / 1000.0, off1, rid)
    H.compare_outputs(got, exp, True)
    m0 = np.array([np.mean(k0[off0[i]:off0[i + 1]] / 1000.0) for i in range(200)])
    s1 = np.array([np.std(k1[off1[i]:off1[i + 1]] / 1000.0) for i in range(200)])
    assert np.allclose(got['mean0'], m0, rtol=1e-12, atol=1e-15) and np.allclose(got['std1'], s1, rtol=1e-12)


def test_int16_ks_only_mode(nm):
    """int16 milli-unit signals through the KS-only kernel (tie-heavy by construction), ragged"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(123)
    sig0, off0, sig1, off1, rid = _random_batch(rng, 300, 5, 700, 5, 300)
    k0 = np.rint(sig0.astype(np.float64) * 1000).astype(np.int16)
    k1 = np.rint(sig1.astype(np.float64) * 1000).astype(np.int16)
    got = nm.detect_host(k0, off0, k1, off1, rid, tests=L.TEST_KS, method='fisher')
    ks = [orc.ks_2samp(k0[off0[i]:off0[i + 1]] / 1000.0, k1[off1[i]:off1[i + 1]] / 1000.0) for i in range(300)]
    H.assert_close_stat(got['ks_d'], np.array([k[0] for k in ks]), 0, 0.0, 'ks_d')
    H.assert_close_p(got['ks_p'], np.maximum(np.array([k[1] for k in ks]), orc.DBL_MIN), 1e-9, 'ks_p')


def test_edge_statuses(nm):
    L = nm._lib
    # position 0: all identical; 1: zero variance in both groups but different means; 2: normal; 3: empty group
    sig0 = np.array([0.5] * 6 + [1.0] * 5 + [0.1, 0.2, 0.3, 0.4, 0.5], dtype=np.float32)
    off0 = np.array([0, 6, 11, 16, 16], dtype=np.int64)
    sig1 = np.array([0.5] * 7 + [2.0] * 5 + [0.3, 0.1, 0.9, 0.7, 0.2] + [1.0, 2.0, 3.0], dtype=np.float32)
    off1 = np.array([0, 7, 12, 17, 20], dtype=np.int64)
    r = nm.detect_host(sig0, off0, sig1, off1, np.zeros(4, np.int32), method='ks')
    assert r['status'][0] & L.STATUS_MWU_ALL_IDENTICAL and np.isnan(r['mwu_p'][0])
    assert r['ks_d'][0] == 0.0 and r['ks_p'][0] == 1.0
    assert np.isnan(r['t_t'][0]) and np.isnan(r['t_p'][0]) and (r['status'][0] & L.STATUS_T_NAN)
    assert r['t_t'][1] == -np.inf and r['t_p'][1] == 2.2250738585072014e-308      # m_min_float clamp
    assert r['ks_d'][1] == 1.0
    assert r['status'][2] == 0
    assert r['status'][3] & L.STATUS_EMPTY
    # more samples than the format allows (NMOD_MAX_RANKED): flagged per position (test_too_large_group_is_flagged_per_position)
    big = np.zeros(65536, np.float32)
    r = nm.detect_host(big, np.array([0, 65536]), big, np.array([0, 65536]), np.zeros(1, np.int32))
    assert (r['status'][0] & L.STATUS_TOO_LARGE) and np.isnan(r['ks_p'][0])
    # empty batch
    r = nm.detect_host(np.zeros(0, np.float32), np.zeros(1, np.int64), np.zeros(0, np.float32), np.zeros(1, np.int64),
                       np.zeros(0, np.int32))
    assert r['ks_p'].shape == (0,)


def test_device_resident_path_and_synth(nm):
    """torch-resident CSR/stride inputs, the device generator against its numpy restatement,
    the HIP-event timer, and size-independent properties at a larger size."""
    import torch
    import nanomod_oracle as orc
    L = nm._lib
    dev = 'cuda:0'
    npos, n = 20000, 200
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    det.timer = nm.EventTimer(64)
    sig0 = torch.empty(npos * n, dtype=torch.float32, device=dev)
    sig1 = torch.empty(npos * n, dtype=torch.float32, device=dev)
    det.synth_fill(sig0, 20240601, 1000, npos, 0, n, 10000, 0.8)
    det.synth_fill(sig1, 20240601, 1000, npos, 1, n, 10000, 0.8)
    torch.cuda.synchronize()
    assert np.array_equal(sig0.cpu().numpy(), H.synth_ref(20240601, 1000, npos, 0, n, 10000, 0.8))
    assert np.array_equal(sig1.cpu().numpy(), H.synth_ref(20240601, 1000, npos, 1, n, 10000, 0.8))
    rid = torch.zeros(npos, dtype=torch.int32, device=dev)
    res = det.run(sig0, sig1, rid, stride0=n, stride1=n, npos=npos)
    torch.cuda.synchronize()
    ms, launches = det.timer.read(L.KERNEL_RANK_STATS)
    assert launches == 1 and ms > 0
    # oracle on a sample of positions, including the planted ones (9999, 10000, 10001 -> index 8999..9001)
    idx = np.r_[0:40, 8990:9010, npos - 40:npos]
    a = sig0.cpu().numpy().reshape(npos, n)
    b = sig1.cpu().numpy().reshape(npos, n)
    ksp = res['ks_p'].cpu().numpy(); ksd = res['ks_d'].cpu().numpy()
    for i in idx:
        d, p = orc.ks_2samp(a[i], b[i])
        assert abs(ksd[i] - d) <= 0.0 and abs(ksp[i] - max(p, 2.2250738585072014e-308)) <= 1e-9 * p
    assert ksp[8999:9002].max() < 1e-6                      # the planted shift is found
    st, pv = orc.combine_track(ksd, ksp, np.zeros(npos, np.int32), 2, 2.0, orc.METHOD_STOUFFER)
    H.assert_close_p(res['comb_p'].cpu().numpy()[idx], pv[idx], 1e-9, 'comb_p')
    # same data through the CSR entry (binned launch) gives identical bits
    off = torch.arange(0, (npos + 1) * n, n, dtype=torch.int64, device=dev)
    res2 = det.run(sig0, sig1, rid, off0=off, off1=off, max_n0=n, max_n1=n)
    torch.cuda.synchronize()
    assert torch.equal(res['ks_p'], res2['ks_p']) and torch.equal(res['comb_st'].nan_to_num(neginf=-1e300), res2['comb_st'].nan_to_num(neginf=-1e300))
    # swap symmetry of D (|F0 - F1| is symmetric) and permutation invariance inside a position
    res3 = det.run(sig1, sig0, rid, stride0=n, stride1=n, npos=npos)
    perm = torch.randperm(n, device=dev)
    sig0p = sig0.view(npos, n)[:, perm].contiguous().view(-1)
    res4 = det.run(sig0p, sig1, rid, stride0=n, stride1=n, npos=npos)
    torch.cuda.synchronize()
    assert torch.equal(res['ks_d'], res3['ks_d']) and torch.equal(res['ks_p'], res4['ks_p'])


@pytest.mark.parametrize('inp,name,method', [('ragged', 'ragged_stouffer', 'stouffer'), ('g50', 'g50_fisher', 'fisher'),
                                             ('g50', 'g50_ks', 'ks'), ('ties', 'ties_stouffer', 'stouffer')])
def test_cli_detect_end_to_end(nm, inp, name, method, capsys):
    """python -m nanomod_amd.cli detect on .npz containers: same `_sign_test.txt` bytes as the reference run"""
    from nanomod_amd import cli
    from test_abi_and_host import _fixture_containers
    exp, table = H.load_expected(name)
    with tempfile.TemporaryDirectory() as tmp:
        p0, p1 = _fixture_containers(inp, tmp)
        rc = cli.main(['detect', '--wrkBase1', p0, '--wrkBase2', p1, '--FileID', 'x', '--outFolder', tmp,
                       '--testMethod', method, '--topN', '5', '--outLevel', '3'])
        assert rc == 0
        assert open(os.path.join(tmp, 'x_sign_test.txt')).read() == table
    lines = capsys.readouterr().out.strip().split('\n')
    assert len(lines) >= 5


def test_region_rank_through_mtest2_and_cli(nm, capsys):
    """RegionRankbyST=1 end to end on the GPU numbers: mtest2 on moptions and the CLI give the reference's window order"""
    from nanomod_amd import cli
    from test_abi_and_host import _fixture_containers
    z = np.load(os.path.join(H.GOLDEN, 'g50_regionrank_w10_o1.npz'))
    fx = H.load_inputs('g50')
    with tempfile.TemporaryDirectory() as tmp:
        mo = H.build_moptions(fx, tmp, 'rr', 2, 2.0, 'stouffer')
        mo.update({'RegionRankbyST': 1, 'window': 10, 'WindOvlp': 1, 'percentile': 0.1, 'NA': '', 'SaveTest': 0})
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        assert [r[0][2] for r in mo['sorted_sign_test']] == list(z['pos'])
        p0, p1 = _fixture_containers('g50', tmp)
        rc = cli.main(['detect', '--wrkBase1', p0, '--wrkBase2', p1, '--FileID', 'x', '--outFolder', tmp, '--RegionRankbyST', '1',
                       '--WindOvlp', '1', '--window', '21', '--topN', '7', '--outLevel', '3'])
        assert rc == 0
    lines = [l for l in capsys.readouterr().out.strip().split('\n') if l and l[0].isdigit()]
    assert [int(l.split()[3]) - 1 for l in lines[-7:]] == list(z['pos'][:7])


def test_device_path_unknown_max_side_stream_and_too_large_hint(nm):
    """DEVICE mode corners: max_n unknown (library reduces the offsets and synchronises once), a non-default
    stream, and a max_n hint smaller than a position (that position gets STATUS_TOO_LARGE and NaN, the rest is exact)"""
    import torch
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(8)
    sig0, off0, sig1, off1, rid = _random_batch(rng, 400, 5, 300, 5, 120)
    dev = 'cuda:0'
    t = lambda a: torch.from_numpy(a).to(dev)
    d0, o0, d1, o1, r = t(sig0), t(off0), t(sig1), t(off1), t(rid)
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, orc.METHOD_STOUFFER)
    for tests in (L.TEST_ALL, L.TEST_KS):
        det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
        res = det.run(d0, d1, r, off0=o0, off1=o1)                       # max_n0 = max_n1 = 0: unknown
        torch.cuda.synchronize()
        H.assert_close_p(res['ks_p'].cpu().numpy(), exp['ks_p'], 1e-9, 'ks_p')
        H.assert_close_p(res['comb_p'].cpu().numpy(), exp['comb_p'], 1e-9, 'comb_p')
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            res2 = det.run(d0, d1, r, off0=o0, off1=o1, max_n0=300, max_n1=120)
        side.synchronize()
        assert torch.equal(res['ks_p'], res2['ks_p']) and torch.equal(res['comb_p'], res2['comb_p'])
        # a hint that is too small for some positions
        n0 = np.diff(off0)
        res3 = det.run(d0, d1, r, off0=o0, off1=o1, max_n0=128, max_n1=128)
        torch.cuda.synchronize()
        st = res3['status'].cpu().numpy(); ksp = res3['ks_p'].cpu().numpy()
        big = n0 > 128
        assert big.any() and np.all((st[big] & L.STATUS_TOO_LARGE) != 0) and np.all(np.isnan(ksp[big]))
        assert np.all(st[~big] & L.STATUS_TOO_LARGE == 0)
        H.assert_close_p(ksp[~big], exp['ks_p'][~big], 1e-9, 'ks_p small')
        # a hint that is NOT a capacity-class boundary (100 < 128), on a workspace holding a previous batch's numbers:
        # positions with 101..128 samples are skipped by K1 and must be flagged, not read from the stale workspace
        n1 = np.diff(off1)
        res4 = det.run(d0, d1, r, off0=o0, off1=o1, max_n0=100, max_n1=100)
        torch.cuda.synchronize()
        st = res4['status'].cpu().numpy()
        over = (n0 > 100) | (n1 > 100)
        between = ((n0 > 100) & (n0 <= 128)) | ((n1 > 100) & (n1 <= 128))
        assert between.any() and np.all((st[over] & L.STATUS_TOO_LARGE) != 0)
        for k in ('ks_d', 'ks_p') + (('mwu_u', 'mwu_p', 't_t', 't_p') if tests == L.TEST_ALL else ()):
            assert np.all(np.isnan(res4[k].cpu().numpy()[over])), k
        assert np.all(st[~over] & L.STATUS_TOO_LARGE == 0)
        H.assert_close_p(res4['ks_p'].cpu().numpy()[~over], exp['ks_p'][~over], 1e-9, 'ks_p under the hint')


def test_count_wide_device_path_unknown_max_hint_and_workspace_reuse(nm):
    """the counting form for any coverage through the DEVICE entry: maxima unknown (the library reduces the offsets), a side stream,
    a max_n hint below some positions (TOO_LARGE there, the rest exact), and one detector (one workspace) running a skewed batch,
    a 500 v 500-like batch and the skewed one again — stale work lists / gates of the previous batch must not leak; both masks"""
    import torch
    import oracle_c
    L = nm._lib
    rng = np.random.default_rng(21)
    P = 500
    a0 = rng.integers(400, 1500, P); b0 = rng.integers(20, 120, P)
    a1 = rng.integers(330, 513, P); b1 = rng.integers(330, 513, P)
    batches = []
    for a, b in ((a0, b0), (a1, b1)):
        k0, off0, k1, off1 = _event_rows(rng, a, b, 200)
        rid = np.zeros(P, np.int32)
        exp = oracle_c.detect_batch(k0, off0, k1, off1, rid, 2, 2.0, 'stouffer', tests=7)
        batches.append((k0, off0, k1, off1, rid, exp))
    t = lambda x: torch.from_numpy(x).to('cuda:0')
    for tests in (L.TEST_ALL, L.TEST_KS):
        det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
        for which in (0, 1, 0):
            k0, off0, k1, off1, rid, exp = batches[which]
            res = det.run(t(k0), t(k1), t(rid), off0=t(off0), off1=t(off1))                 # maxima unknown
            torch.cuda.synchronize()
            assert np.array_equal(res['ks_d'].cpu().numpy(), exp['ks_d'])
            H.assert_close_p(res['comb_p'].cpu().numpy(), exp['comb_p'], 1e-9, 'comb_p')
            if tests == L.TEST_ALL:
                assert np.array_equal(res['mwu_u'].cpu().numpy(), exp['mwu_u'])
                H.assert_close_p(res['t_p'].cpu().numpy(), exp['t_p'], 1e-9, 't_p')
        k0, off0, k1, off1, rid, exp = batches[0]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            res = det.run(t(k0), t(k1), t(rid), off0=t(off0), off1=t(off1), max_n0=1000, max_n1=200)      # some groups exceed the hint
        side.synchronize()
        n0 = np.diff(off0)
        big = n0 > 1000
        st = res['status'].cpu().numpy()
        assert big.any() and np.all((st[big] & L.STATUS_TOO_LARGE) != 0) and np.all(st[~big] & L.STATUS_TOO_LARGE == 0)
        assert np.array_equal(res['ks_d'].cpu().numpy()[~big], exp['ks_d'][~big]) and np.all(np.isnan(res['ks_d'].cpu().numpy()[big]))


def test_ks_only_large_ranked_group(nm):
    """KS-only mode: only the smaller group of a position is capacity-bound (2048); the other one is ranked,
    not sorted, and may be much larger (cfg5's 4000-read positions)"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(77)
    sizes = [(4000, 50), (30, 9000), (2048, 2049), (6000, 400), (3000, 2000), (5, 20000)]
    ca = [rng.normal(0, 1, a).astype(np.float32) for a, b in sizes]
    cb = [np.round(rng.normal(0.1, 1, b), 2).astype(np.float32) for a, b in sizes]
    off0 = np.zeros(len(sizes) + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(len(sizes) + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    got = nm.detect_host(np.concatenate(ca), off0, np.concatenate(cb), off1, np.zeros(len(sizes), np.int32),
                         tests=L.TEST_KS, method='stouffer')
    for i, (a, b) in enumerate(zip(ca, cb)):
        d, p = orc.ks_2samp(a, b)
        assert abs(got['ks_d'][i] - d) <= 0.0 and abs(got['ks_p'][i] - max(p, orc.DBL_MIN)) <= 1e-9 * max(p, orc.DBL_MIN), i


@pytest.mark.parametrize('mode', ['cont', 'grid1', 'const', 'i16', 'i16const', 'i16span', 'i16heavy', 'f64', 'f64ties',
                                  'g32', 'g32span', 'g32heavy', 'g32mixed', 'offconst', 'i16tails', 'g32tails'])
def test_unequal_classes_streamed_larger_group(nm, mode):
    """positions whose groups fall in different capacity classes with the smaller one <= 256 samples: the WIDE form of
    rank_hist_kernel (smaller group sorted, larger one streamed and counted in a per-wave hash table), the larger group up
    to 2 048 in one pass and 2 049 .. 4 096 in two; either group may be the larger one.  `const`: every sample of a
    position equal — the longest probe chains and the largest tie sums (256 keys tied with 4 096 samples: a b (a + b)
    and the sum of t^3 - t pass 2^32).  int16 input counts the ties of the larger group in direct-address 8-bit counters
    over a window of the value domain: `i16const` (hundreds of equal samples: the counters overflow and the position is
    recounted with 16-bit ones), `i16span` (values over the whole int16 range: one pass per window), `i16heavy` (a
    heavy value of 255 / 256 / 257 copies among spread ones: the edge of the overflow test).  float32 input whose smaller
    group is on the milli-unit grid of real events takes the same counters (rank_hist.hpp: grid_key) while the larger
    group's samples are on the grid too: `g32` (3-decimal values), `g32span` / `g32heavy` (the recount paths), `g32mixed`
    (samples off the grid, one ulp beside it, or beyond +-32.767 in either group: back to the hash, mid-position); `offconst`:
    hundreds to thousands of equal samples off the grid — the bitmap form's fall-back, the hash passes with the table full.
    `i16tails` / `g32tails` (round 6): event-like rows with samples of the streamed group OUTSIDE the counters' window — 1, 2 equal
    ones, a dozen with repeats, exactly 64, 65 and 200 (past the tail list: the recount), and tails beside a value of 256 copies;
    the counting forms are switched off for these two (NMOD_FLAG_NO_COUNTING), so the sorting form's tail list is what runs"""
    import nanomod_oracle as orc
    rng = np.random.default_rng(zlib.crc32(mode.encode()))
    sizes = [(50, 1000), (1000, 50), (3, 130), (64, 65), (65, 2048), (130, 700), (256, 2048), (256, 4096), (4096, 256),
             (1, 2049), (2500, 7), (200, 3000), (100, 129), (128, 1025), (256, 4095), (17, 4096)]
    ca, cb = [], []
    for i, (a, b) in enumerate(sizes):
        x = rng.normal(0, 1, a); y = rng.normal(0.3 if i % 2 else 0.0, 1.2, b)
        if mode == 'grid1':
            x, y = np.round(x, 1), np.round(y, 1)
        elif mode in ('g32', 'g32mixed'):
            x, y = np.round(x, 3), np.round(y, 3)
        elif mode in ('i16span', 'g32span'):
            x = rng.integers(-32768, 32768, a) / 1000.0; y = rng.integers(-32768, 32768, b) / 1000.0
            y[::5] = y[0]; x[::4] = y[1 % b]
            if i % 4 == 0:
                y[:] = np.where(rng.random(b) < 0.5, -32.768, 32.767)   # the two ends of the domain only
        elif mode in ('i16heavy', 'g32heavy'):
            big, small = (x, y) if a > b else (y, x)
            k = (255, 256, 257)[i % 3]
            if len(big) > k:
                big[rng.permutation(len(big))[:k]] = 0.123
            small[: len(small) // 2] = 0.123
        elif mode == 'offconst':                      # heavy ties OFF the milli-unit grid: bitmaps -> the hash passes at full load
            x[:] = 0.1234567; y[:] = 0.1234567
            if i % 3 == 0:
                y[: b // 2] = -1.7654321
            elif i % 3 == 1:
                y[::3] = rng.normal(0, 1, len(y[::3]))
        elif mode in ('i16tails', 'g32tails'):
            lev = rng.uniform(-1, 1)
            x = np.round(lev + 0.2 * rng.normal(0, 1, a), 3); y = np.round(lev + 0.2 * rng.normal(0.3, 1, b), 3)
            big = x if a > b else y
            nt = (1, 2, 12, 64, 65, 200, 30, 0)[i % 8]
            nt = min(nt, len(big) - 1)
            far = np.round(lev + rng.choice([-1.0, 1.0], nt) * rng.uniform(4.3, 9.0, nt), 3)
            if nt >= 2:
                far[1::3] = far[0]                     # copies among the outside samples (2: a pair; 12 ..: every third)
            if nt == 30 and len(big) > 400:
                big[40:296] = np.round(lev, 3)         # 256 copies of one value inside the window: the counter wraps -> recount
            big[rng.permutation(len(big))[:nt] if nt != 30 else np.arange(nt)] = far
        elif mode in ('const', 'i16const'):
            x[:] = 0.25; y[:] = 0.25
            if i % 3 == 0:
                y[: b // 2] = -1.5                    # two runs in the larger group, one of them tied with all of the smaller
        ca.append(x); cb.append(y)
    off0 = np.zeros(len(sizes) + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(len(sizes) + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    rid = np.zeros(len(sizes), np.int32)
    if mode.startswith('i16'):
        sig0 = np.round(np.concatenate(ca) * 1000).astype(np.int16); sig1 = np.round(np.concatenate(cb) * 1000).astype(np.int16)
        r0, r1 = sig0.astype(np.float64) / 1000, sig1.astype(np.float64) / 1000
    elif mode.startswith('f64'):
        # arbitrary doubles (neither float32-exact nor on the 0.001 grid): the float64 front end ranks rounded float32
        # images and redoes the positions whose images tie; 'f64ties': pairs of doubles one ulp of float32 apart (their
        # images tie, the doubles do not) and exact duplicates across the groups
        sig0 = np.concatenate(ca); sig1 = np.concatenate(cb)
        if mode == 'f64ties':
            k = min(len(sig0), len(sig1)) // 3
            sig1[:k] = sig0[:k] * (1.0 + 2.0 ** -30)
            sig1[k:2 * k:7] = sig0[k:2 * k:7]
        r0, r1 = sig0, sig1
    else:
        sig0 = np.concatenate(ca).astype(np.float32); sig1 = np.concatenate(cb).astype(np.float32)
        if mode == 'g32mixed':
            for i, (a, b) in enumerate(sizes):
                big_, ob = (sig0, off0) if a > b else (sig1, off1)          # the streamed group of position i
                sm_, os_ = (sig1, off1) if a > b else (sig0, off0)
                j = int(ob[i]) + (7 * i) % max(a, b)
                if i % 4 == 0:
                    big_[j] = np.nextafter(big_[j], np.float32(9))          # one ulp beside a grid value: ties with nothing
                    if j + 1 < ob[i + 1]:
                        big_[j + 1] = big_[j]                               # ... except its own copy
                elif i % 4 == 1:
                    big_[j] = np.float32(40.0)                              # on the 3-decimal grid but beyond int16 milli-units
                elif i % 4 == 2:
                    sm_[int(os_[i])] = np.float32(0.1234567)                # the sorted group off the grid: the hash from the start
                else:                                                       # grid values far outside the counters' reach: the recount
                    big_[j] = np.float32(200.0) if i % 8 == 3 else np.float32(1.0e10)   # refuses the range -> the hash passes
                    if j + 2 < ob[i + 1]:
                        big_[j + 2] = big_[j]
        r0, r1 = sig0, sig1
    tails = mode in ('i16tails', 'g32tails')
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=1, weights_dif=2.0, method='fisher', flags=nm._lib.FLAG_NO_COUNTING if tails else 0)
    exp = orc.detect_batch(r0, off0, r1, off1, rid, 1, 2.0, orc.METHOD_FISHER)
    if tails:
        st = nm._lib.last_dispatch_stats()
        assert st['rank_count'] == 0 and st['rank_count_wide'] == 0 and st['rank_hist_wide'] >= 8, st
    ident = (exp['status'] & 1) != 0                  # all samples identical: U / p NaN on both sides
    assert np.array_equal(np.isnan(got['mwu_u']), np.isnan(exp['mwu_u']))
    got['mwu_u'][ident] = 0.0; exp['mwu_u'][ident] = 0.0
    H.compare_outputs(got, exp, True)
    assert np.array_equal(got['status'], exp['status'])
    # what runs: the dispatch description names the kernel for the sizes of the extremes
    L = nm._lib
    import ctypes as C
    buf = C.create_string_buffer(128)
    cw = b'rank_count_wide_kernel<f32> (event-like rows) | '
    for (a, b), want in (((50, 1000), cw + b'rank_hist_kernel<1,64,f32,wide>'), ((256, 4096), cw + b'rank_hist_kernel<4,64,f32,wide>'),
                         ((4096, 130), cw + b'rank_hist_kernel<4,64,f32,wide>'), ((300, 4096), b'big_hist_kernel<f32>')):
        prm = L.make_params(method=L.METHOD_FISHER, nb=1)
        assert L.load().nmod_describe_dispatch(C.byref(prm), a, b, buf, 128) == 0
        assert buf.value == want, (a, b, buf.value)


def test_wide_redo_list_random_heavy_ties(nm):
    """the WIDE float32 form's redo list (wide_redo_kernel): streamed groups of 300 .. 4 096 samples drawn from a handful to a
    few hundred distinct values OFF the milli-unit grid — more samples on shared bitmap bits than the exact table takes — mixed
    with continuous positions and positions on the grid in one batch, either group the larger one, CSR and fixed stride"""
    import nanomod_oracle as orc
    rng = np.random.default_rng(2024)
    ca, cb = [], []
    for i in range(48):
        small = int(rng.integers(3, 257)); large = int(rng.integers(300, 4097))
        kind = i % 4
        if kind == 0:                                             # continuous
            x = rng.normal(0, 1, small); y = rng.normal(0.1, 1.1, large)
        elif kind == 1:                                           # on the grid
            x = np.round(rng.normal(0, 1, small), 3); y = np.round(rng.normal(0.1, 1.1, large), 3)
        else:                                                     # few distinct values off the grid (multiples of an odd step)
            nvals = int(rng.choice([2, 7, 40, 300]))
            vals = (rng.permutation(4 * nvals)[:nvals] - 2 * nvals) * 0.013700001
            y = rng.choice(vals, large); x = rng.choice(np.r_[vals[: max(1, nvals // 2)], rng.normal(0, 1, 5)], small)
        if i % 3 == 0:
            x, y = y, x
        ca.append(x.astype(np.float32)); cb.append(y.astype(np.float32))
    off0 = np.zeros(len(ca) + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(len(cb) + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    sig0, sig1 = np.concatenate(ca), np.concatenate(cb)
    rid = np.zeros(len(ca), np.int32)
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=1, weights_dif=2.0, method='fisher')
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 1, 2.0, orc.METHOD_FISHER)
    H.compare_outputs(got, exp, True)
    assert np.array_equal(got['status'], exp['status'])
    # fixed stride (no class lists, no cleared counters between batches): 40 v 3 000, every position heavy; twice in a row
    P, n0, n1 = 64, 40, 3000
    vals = np.arange(-6, 7) * np.float32(0.0771)
    a = rng.choice(vals, P * n0).astype(np.float32); b = rng.choice(vals, P * n1).astype(np.float32)
    o0 = np.arange(0, (P + 1) * n0, n0, dtype=np.int64); o1 = np.arange(0, (P + 1) * n1, n1, dtype=np.int64)
    exp = orc.detect_batch(a, o0, b, o1, np.zeros(P, np.int32), 1, 2.0, orc.METHOD_FISHER)
    for _ in range(2):
        got = nm.detect_host(a, None, b, None, np.zeros(P, np.int32), nb=1, weights_dif=2.0, method='fisher', stride0=n0, stride1=n1)
        H.compare_outputs(got, exp, True)


@pytest.mark.parametrize('grid', [False, True])
def test_large_positions_all_tests_and_ks_only(nm, grid):
    """positions beyond the wave-resident kernels (> 2048 samples in a sorted group) take big_rank_kernel: LDS sort
    (<= 8192 keys) and in-slab sort (> 8192), mixed with ordinary positions in one ragged batch, ties, both modes"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(78 + grid)
    sizes = [(3000, 2500), (100, 90), (5000, 100), (2049, 2049), (9000, 8500), (40, 12000), (2048, 2048), (4097, 300),
             (20000, 17000), (7, 5)]
    ca = [rng.normal(0, 1, a) for a, b in sizes]
    cb = [rng.normal(0.05 if i % 3 else 0.0, 1.1, b) for i, (a, b) in enumerate(sizes)]
    if grid:
        ca = [np.round(c, 2) for c in ca]; cb = [np.round(c, 2) for c in cb]
        cb[3] = ca[3].copy()                                            # identical groups
    ca = [c.astype(np.float32) for c in ca]; cb = [c.astype(np.float32) for c in cb]
    off0 = np.zeros(len(sizes) + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(len(sizes) + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    sig0, sig1 = np.concatenate(ca), np.concatenate(cb)
    rid = np.zeros(len(sizes), np.int32)
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='stouffer')
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, orc.METHOD_STOUFFER)
    H.compare_outputs(got, exp, True)
    assert np.array_equal(got['status'], exp['status'])
    ks = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    H.assert_close_stat(ks['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    H.assert_close_p(ks['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    # int16 milli-unit input through the same path
    q0 = np.round(sig0 * 1000).astype(np.int16); q1 = np.round(sig1 * 1000).astype(np.int16)
    got16 = nm.detect_host(q0, off0, q1, off1, rid, nb=2, weights_dif=2.0, method='stouffer')
    exp16 = orc.detect_batch(q0.astype(np.float64) / 1000, off0, q1.astype(np.float64) / 1000, off1, rid, 2, 2.0, orc.METHOD_STOUFFER)
    H.compare_outputs(got16, exp16, True)


def test_large_positions_fixed_stride_and_limit(nm):
    """a fixed-coverage batch whose every position is large, and the format's limit of 65 535 samples per group"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(5)
    npos, n0, n1 = 7, 2500, 3100
    sig0 = rng.normal(0, 1, npos * n0).astype(np.float32); sig1 = np.round(rng.normal(0.1, 1, npos * n1), 3).astype(np.float32)
    off0 = np.arange(npos + 1, dtype=np.int64) * n0; off1 = np.arange(npos + 1, dtype=np.int64) * n1
    rid = np.zeros(npos, np.int32)
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, orc.METHOD_FISHER)
    got = nm.detect_host(sig0, None, sig1, None, rid, nb=2, weights_dif=2.0, method='fisher', stride0=n0, stride1=n1)
    H.compare_outputs(got, exp, True)
    ks = nm.detect_host(sig0, None, sig1, None, rid, nb=2, weights_dif=2.0, method='fisher', stride0=n0, stride1=n1, tests=L.TEST_KS)
    H.assert_close_stat(ks['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    H.assert_close_p(ks['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    big = np.zeros(65536, np.float32)
    for tests in (L.TEST_ALL, L.TEST_KS):
        r = nm.detect_host(big, np.array([0, 65536]), big[:10], np.array([0, 10]), np.zeros(1, np.int32), tests=tests, method='ks')
        assert (r['status'][0] & L.STATUS_TOO_LARGE) and np.isnan(r['ks_d'][0])        # flagged, not refused (round 5)


def test_downsample_entry_point_is_chunk_invariant_and_exact_without_draws(nm, monkeypatch):
    """nmod_downsample_ks (the device form of myDetect.py:345-361): positions whose groups are under their threshold are not
    drawn from — all `iters` virtual rows are the position itself, the selected pair is its plain KS pair, bit for bit; the draws
    are keyed by the position's index among the flagged ones, so the chunking of the virtual rows changes nothing; another seed
    changes the draws; float32, int16 and float64 rows."""
    from nanomod_amd import engine
    L = nm._lib
    rng = np.random.default_rng(9)
    npos = 90
    n0 = rng.integers(10, 300, npos); n1 = rng.integers(10, 300, npos)
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum(n1)
    a = np.round(rng.normal(0, 1, off0[-1]), 3); b = np.round(rng.normal(0.3, 1, off1[-1]), 3)
    for dtype in (np.float32, np.int16, np.float64):
        s0 = np.rint(a * 1000).astype(np.int16) if dtype == np.int16 else a.astype(dtype)
        s1 = np.rint(b * 1000).astype(np.int16) if dtype == np.int16 else b.astype(dtype)
        plain = nm.detect_host(s0, off0, s1, off1, np.zeros(npos, np.int32), tests=L.TEST_KS, method='ks')
        pos = np.arange(npos)[::-1].copy()                      # any order, every position
        cov = np.where(np.arange(npos)[::-1] % 3 == 0, 100000, 60)   # a third of them far above both groups: no draws
        d1, p1 = engine.downsample_ks(s0, off0, s1, off1, pos, cov, iters=100, quantile=0.25, seed=5)
        nodraw = (n0[pos] <= cov) & (n1[pos] <= cov)
        assert nodraw.sum() >= 25 and (~nodraw).sum() >= 25
        assert np.array_equal(d1[nodraw], plain['ks_d'][pos][nodraw]) and np.array_equal(p1[nodraw], plain['ks_p'][pos][nodraw])
        assert np.all((p1 > 0) & (p1 <= 1) & (d1 >= 0) & (d1 <= 1))
        monkeypatch.setenv('NMOD_DOWNSAMPLE_ELEMENTS', '60000')     # ~3 positions per chunk
        d2, p2 = engine.downsample_ks(s0, off0, s1, off1, pos, cov, iters=100, quantile=0.25, seed=5)
        monkeypatch.delenv('NMOD_DOWNSAMPLE_ELEMENTS')
        assert np.array_equal(d1, d2) and np.array_equal(p1, p2)
        d3, p3 = engine.downsample_ks(s0, off0, s1, off1, pos, cov, iters=100, quantile=0.25, seed=6)
        assert np.array_equal(p3[nodraw], p1[nodraw]) and np.mean(p3[~nodraw] != p1[~nodraw]) > 0.5
        # the quantile index walks the sorted resamples: p at 0.0 <= p at 0.25 <= p at 0.9 for every position
        plo = engine.downsample_ks(s0, off0, s1, off1, pos, cov, iters=100, quantile=0.0, seed=5)[1]
        phi = engine.downsample_ks(s0, off0, s1, off1, pos, cov, iters=100, quantile=0.9, seed=5)[1]
        assert np.all(plo <= p1) and np.all(p1 <= phi) and np.any(plo < phi)
    d0, p0 = engine.downsample_ks(s0, off0, s1, off1, np.zeros(0, np.int64), np.zeros(0, np.int64))
    assert len(d0) == 0 and len(p0) == 0


def test_downsampling_branch_statistically_matches(nm):
    """--coverages > 0 (myDetect.py:345-361): seeded on the device, unseeded in the reference, so compare
    distributions: over many positions the selected 25th-percentile p-values of the two implementations must
    agree (two-sample KS between them not significant, medians close), MWU / Welch are untouched, positions
    under the threshold keep the plain KS pair, and the run is reproducible for a fixed seed."""
    import nanomod_oracle as orc
    rng = np.random.default_rng(5)
    npos, cov = 160, 40
    chunks_a = [rng.normal(0, 1, int(rng.integers(60, 140))).astype(np.float32) for _ in range(npos)]
    chunks_b = [rng.normal(0.25, 1, int(rng.integers(20, 140))).astype(np.float32) for _ in range(npos)]
    mo = {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [cov, cov], 'downsampling': 100,
          'downsampling_quantile': 0.25, 'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv',
          'SaveTest': 0, 'RegionRankbyST': 0, 'outFolder': '/tmp', 'FileID': 'ds', 'MinCoverage': 5, 'nmod_seed': 11}
    for g, (ds, ch) in enumerate((('A', chunks_a), ('B', chunks_b))):
        mo[ds] = {'norm_mean': {('c', '+'): {100 + i: [np.float64(v) for v in ch[i]] for i in range(npos)}},
                  'base': {('c', '+'): {100 + i: 'A' for i in range(npos)}}, 'basedict': {}}
    import copy
    mo2 = copy.deepcopy(mo)
    nm.mtest2(mo)
    nm.mtest2(mo2)
    got_p = np.array([r[1][2][1] for r in mo['sign_test']]); got_d = np.array([r[1][2][0] for r in mo['sign_test']])
    assert np.array_equal(got_p, np.array([r[1][2][1] for r in mo2['sign_test']]))          # seeded: reproducible
    ref = [orc.ks_downsampled(chunks_a[i], chunks_b[i], cov, 100, 0.25, np.random.default_rng(1000 + i)) for i in range(npos)]
    ref_p = np.array([r[1] for r in ref]); ref_d = np.array([r[0] for r in ref])
    small = np.array([len(chunks_a[i]) <= cov and len(chunks_b[i]) <= cov for i in range(npos)])
    assert not small.any()                                   # every position here is down-sampled
    from scipy import stats
    assert stats.ks_2samp(np.log(got_p), np.log(ref_p)).pvalue > 0.01
    assert abs(np.median(np.log(got_p)) - np.median(np.log(ref_p))) < 0.35
    assert abs(got_d.mean() - ref_d.mean()) < 0.02
    # per position the two selected p-values are draws from the same sampling distribution: same order of magnitude
    assert np.mean(np.abs(np.log(got_p) - np.log(ref_p)) < 1.5) > 0.9
    # MWU / Welch come from the full data
    full = [orc.getKStest(chunks_a[i].astype(np.float64), chunks_b[i].astype(np.float64)) for i in range(5)]
    for i in range(5):
        assert mo['sign_test'][i][1][0][0] == full[i][0][0] and abs(mo['sign_test'][i][1][1][1] - full[i][1][1]) < 1e-9 * full[i][1][1]
    # below the threshold nothing changes
    mo3 = copy.deepcopy(mo); mo3['coverages'] = [1000, 1000]
    for ds in ('A', 'B'):
        mo3[ds] = mo2[ds]
    mo4 = copy.deepcopy(mo3); mo4['coverages'] = [0, 0]
    nm.mtest2(mo3); nm.mtest2(mo4)
    assert [r[1][2] for r in mo3['sign_test']] == [r[1][2] for r in mo4['sign_test']]
    # the single-position entry point takes the same branch
    one = nm.getKStest({'coverages': [cov, cov], 'nmod_seed': 3}, chunks_a[0], chunks_b[0], '+')
    assert 0 < one[2][1] <= 1 and one[0][0] == full[0][0][0]


def test_downsampling_cli_matches_mtest2_same_seed(nm):
    """`--coverages 12 --seed 4` through the CLI writes the same table as mtest2 with nmod_seed=4 on the same groups"""
    from nanomod_amd import cli
    from test_abi_and_host import _fixture_containers
    fx = H.load_inputs('g50')
    with tempfile.TemporaryDirectory() as tmp:
        mo = H.build_moptions(fx, tmp, 'ds', 2, 2.0, 'stouffer')
        mo.update({'coverages': [12, 12], 'nmod_seed': 4, 'SaveTest': 1})
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
        p0, p1 = _fixture_containers('g50', tmp)
        rc = cli.main(['detect', '--wrkBase1', p0, '--wrkBase2', p1, '--FileID', 'x', '--outFolder', tmp, '--coverages', '12',
                       '--seed', '4', '--topN', '3', '--outLevel', '3'])
        assert rc == 0
        a = open(os.path.join(tmp, 'ds_sign_test.txt')).read(); b = open(os.path.join(tmp, 'x_sign_test.txt')).read()
        assert a == b
        exp, table = H.load_expected('g50_stouffer')
        assert a != table                                # the branch did change some KS pairs


def test_welch_p_value_grid(nm):
    """the incomplete-beta continued fraction (division-free forward recurrence) over a grid of degrees of freedom
    and t values: tiny and huge groups, null to extreme shifts (p from 1 down to the DBL_MIN clamp)"""
    import nanomod_oracle as orc
    rng = np.random.default_rng(99)
    ca, cb = [], []
    for n0, n1 in [(2, 2), (2, 9), (3, 3), (5, 7), (30, 30), (200, 180), (1000, 40), (2000, 2000), (6000, 5000)]:
        for delta in (0.0, 1e-3, 0.03, 0.3, 1.0, 3.0, 10.0, 60.0):
            for s1 in (1.0, 0.05, 7.0):
                ca.append(rng.normal(0, 1, n0)); cb.append(rng.normal(delta, s1, n1))
    ca = [c.astype(np.float32) for c in ca]; cb = [c.astype(np.float32) for c in cb]
    off0 = np.zeros(len(ca) + 1, np.int64); off0[1:] = np.cumsum([len(c) for c in ca])
    off1 = np.zeros(len(cb) + 1, np.int64); off1[1:] = np.cumsum([len(c) for c in cb])
    sig0, sig1 = np.concatenate(ca), np.concatenate(cb)
    rid = np.zeros(len(ca), np.int32)
    got = nm.detect_host(sig0, off0, sig1, off1, rid, method='ks')
    exp = orc.detect_batch(sig0, off0, sig1, off1, rid, 2, 2.0, orc.METHOD_KS)
    H.assert_close_stat(got['t_t'], exp['t_t'], 1e-11, 1e-15, 't_t')
    H.assert_close_p(got['t_p'], exp['t_p'], 1e-9, 't_p')
    assert (exp['t_p'] < 1e-200).any() and (exp['t_p'] > 0.5).any()


def test_float64_input_dtype(nm):
    """NMOD_DTYPE_F64: float64 samples as the reference holds them are re-encoded on the device — float32 when every
    value is float32-exact, int16 milli-units when every value is k/1000, sorted as 64-bit keys otherwise — in host
    and device memory, CSR with a non-zero first offset and fixed stride"""
    import torch
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(11)
    npos = 70
    n0 = rng.integers(3, 300, npos); n1 = rng.integers(3, 300, npos)
    pad = 5                                                            # the CSR arrays start at sample 5
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum(n0); off0 += pad
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum(n1); off1 += pad
    rid = (np.arange(npos) // 9).astype(np.int32)
    a32 = rng.normal(0, 1, off0[-1]).astype(np.float32); b32 = rng.normal(0.2, 1, off1[-1]).astype(np.float32)
    a3 = np.round(rng.normal(0, 1, off0[-1]), 3); b3 = np.round(rng.normal(0.2, 1, off1[-1]), 3)
    for a, b, same_as in ((a32.astype(np.float64), b32.astype(np.float64), (a32, b32)),
                          (a3, b3, (np.rint(a3 * 1000).astype(np.int16), np.rint(b3 * 1000).astype(np.int16)))):
        got = nm.detect_host(a, off0, b, off1, rid, nb=2, weights_dif=2.0, method='stouffer')
        ref = nm.detect_host(same_as[0], off0, same_as[1], off1, rid, nb=2, weights_dif=2.0, method='stouffer')
        # the rank statistics are those of the narrower dtype bit for bit (order-preserving keys); the Welch moments
        # are taken from the float64 samples themselves (two-pass), so t agrees to rounding
        for k in ref:
            if k in ('t_t', 't_p'):
                assert np.allclose(got[k], ref[k], rtol=1e-9 if k == 't_p' else 1e-11, atol=1e-14, equal_nan=True), k
            else:
                assert np.array_equal(got[k], ref[k], equal_nan=True), k
        exp = orc.detect_batch(a[pad:], off0 - pad, b[pad:], off1 - pad, rid, 2, 2.0, orc.METHOD_STOUFFER)
        H.compare_outputs(got, exp, True)
        # device-resident float64 tensors through the same entry point
        det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer')
        r = det.run(torch.as_tensor(a, device='cuda:0'), torch.as_tensor(b, device='cuda:0'), torch.as_tensor(rid, device='cuda:0'),
                    off0=torch.as_tensor(off0, device='cuda:0'), off1=torch.as_tensor(off1, device='cuda:0'))
        torch.cuda.synchronize()
        for k in ('ks_p', 'mwu_p', 'comb_p'):
            assert np.array_equal(r[k].cpu().numpy(), ref[k], equal_nan=True), k
        assert np.allclose(r['t_p'].cpu().numpy(), ref['t_p'], rtol=1e-9, atol=0, equal_nan=True)
    # fixed stride, KS-only
    s0 = np.round(rng.normal(0, 1, 40 * 50), 3); s1 = np.round(rng.normal(0, 1, 40 * 60), 3)
    g = nm.detect_host(s0, None, s1, None, np.zeros(40, np.int32), stride0=50, stride1=60, tests=L.TEST_KS, method='ks')
    r = nm.detect_host(np.rint(s0 * 1000).astype(np.int16), None, np.rint(s1 * 1000).astype(np.int16), None, np.zeros(40, np.int32),
                       stride0=50, stride1=60, tests=L.TEST_KS, method='ks')
    assert np.array_equal(g['ks_p'], r['ks_p'])
    # neither float32-exact nor on the grid: float32 images as keys, and the positions whose images tie are redone on the
    # float64 samples themselves (64-bit keys) — any float64 input the reference accepts is accepted, with ties between
    # doubles that differ below float32 resolution kept apart
    a64 = rng.normal(0, 1, off0[-1]); b64 = rng.normal(0.2, 1, off1[-1])
    a64[off0[3]:off0[3] + 2] = [0.1, 0.1 + 1e-12]; b64[off1[3]:off1[3] + 2] = [0.1 + 1e-12, 0.1 + 2e-12]   # equal as float32
    b64[off1[5]:off1[6]] = np.resize(a64[off0[5]:off0[6]], n1[5])                                         # exact cross ties
    for tests in (L.TEST_ALL, L.TEST_KS):
        got = nm.detect_host(a64, off0, b64, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
        exp = orc.detect_batch(a64[pad:], off0 - pad, b64[pad:], off1 - pad, rid, 2, 2.0, orc.METHOD_STOUFFER)
        if tests == L.TEST_ALL:
            H.compare_outputs(got, exp, True)
        else:
            H.assert_close_stat(got['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
            H.assert_close_p(got['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    r = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer').run(
        torch.as_tensor(a64, device='cuda:0'), torch.as_tensor(b64, device='cuda:0'), torch.as_tensor(rid, device='cuda:0'),
        off0=torch.as_tensor(off0, device='cuda:0'), off1=torch.as_tensor(off1, device='cuda:0'))
    torch.cuda.synchronize()
    assert np.array_equal(r['mwu_p'].cpu().numpy(), got0 := nm.detect_host(a64, off0, b64, off1, rid, nb=2, weights_dif=2.0, method='stouffer')['mwu_p'])


def test_float64_mixed_batch_per_position_keys(nm):
    """one batch whose positions are float32-exact, on the 0.001 grid, arbitrary doubles without ties, arbitrary doubles
    with exact ties and with float32-image ties, and arbitrary doubles beyond the wave-resident kernels: every position
    gets the reference's numbers (the decision is per position, not per batch), in both modes, CSR and fixed stride"""
    import nanomod_oracle as orc
    L = nm._lib
    rng = np.random.default_rng(77)
    npos = 120
    n0 = rng.integers(5, 260, npos); n1 = rng.integers(5, 260, npos)
    n0[7], n1[7] = 2600, 40                                            # a large position of arbitrary doubles (all tests: big pass)
    n0[8], n1[8] = 2300, 2500                                          # large in both modes
    off0 = np.zeros(npos + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(npos + 1, np.int64); off1[1:] = np.cumsum(n1)
    a = rng.normal(0, 1, off0[-1]); b = rng.normal(0.15, 1, off1[-1])
    for i in range(npos):
        sa, sb = slice(off0[i], off0[i + 1]), slice(off1[i], off1[i + 1])
        kind = i % 5
        if kind == 0:
            a[sa] = a[sa].astype(np.float32); b[sb] = b[sb].astype(np.float32)
        elif kind == 1:
            a[sa] = np.round(a[sa], 3); b[sb] = np.round(b[sb], 3)
        elif kind == 2 and i not in (7, 8):
            b[sb][: min(n0[i], n1[i]) // 2] = a[sa][: min(n0[i], n1[i]) // 2]          # exact ties across the groups
            a[sa][-1] = a[sa][0]                                                        # and inside one
        elif kind == 3:
            a[sa][1] = a[sa][0] * (1 + 2e-16); b[sb][0] = a[sa][0] * (1 - 2e-16)        # distinct doubles, equal float32 images
    rid = (np.arange(npos) // 17).astype(np.int32)
    exp = orc.detect_batch(a, off0, b, off1, rid, 2, 2.0, orc.METHOD_STOUFFER)
    got = nm.detect_host(a, off0, b, off1, rid, nb=2, weights_dif=2.0, method='stouffer')
    H.compare_outputs(got, exp, True)
    assert np.array_equal(got['status'], exp['status'])
    ks = nm.detect_host(a, off0, b, off1, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    H.assert_close_stat(ks['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    H.assert_close_p(ks['ks_p'], exp['ks_p'], 1e-9, 'ks_p')
    H.assert_close_p(ks['comb_p'], exp['comb_p'], 1e-9, 'comb_p')
    # fixed stride: 200 v 200 arbitrary doubles, a few positions with image ties
    P, n = 3000, 200
    sa = rng.normal(0, 1, P * n); sb = rng.normal(0.1, 1, P * n)
    for i in range(0, P, 97):
        sb[i * n + 3] = sa[i * n + 5] * (1 + 3e-16); sb[i * n + 4] = sa[i * n + 6]
    o = np.arange(0, (P + 1) * n, n, dtype=np.int64)
    exp = orc.detect_batch(sa, o, sb, o, np.zeros(P, np.int32), 2, 2.0, orc.METHOD_STOUFFER)
    got = nm.detect_host(sa, None, sb, None, np.zeros(P, np.int32), stride0=n, stride1=n, nb=2, weights_dif=2.0, method='stouffer')
    H.compare_outputs(got, exp, True)
    ks = nm.detect_host(sa, None, sb, None, np.zeros(P, np.int32), stride0=n, stride1=n, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    H.assert_close_stat(ks['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    H.assert_close_p(ks['comb_p'], exp['comb_p'], 1e-9, 'comb_p')


def test_rank_order_entry_point(nm):
    """nmod_rank_order against Python's stable tuple sort: heavy ties in every key, -0.0 / 0.0, the reversed 'st' form"""
    rng = np.random.default_rng(4)
    n = 20000
    k1 = np.round(rng.normal(0, 1, n), 1); k2 = np.round(rng.normal(0, 1, n), 0); k3 = rng.normal(0, 1, n)
    k1[::7] = 0.0; k1[3::7] = -0.0
    k3[::5] = 1e-300; k3[1::5] = 2.2250738585072014e-308
    recs = list(zip(k1.tolist(), k2.tolist(), k3.tolist(), range(n)))
    exp = [r[3] for r in sorted(recs, key=lambda r: (r[0], r[1], r[2]))]
    got = nm.engine.rank_order_host(k1, k2, k3)
    assert got.tolist() == exp
    assert nm.engine.rank_order_host(k1, k2, k3, descending=True).tolist() == exp[::-1]
    assert nm.engine.rank_order_host(k1[:1], k2[:1], k3[:1]).tolist() == [0]
    assert nm.engine.rank_order_host(k1[:0], k2[:0], k3[:0]).tolist() == []


@pytest.mark.parametrize('n', [1, 255, 2047, 2048, 2049, 4096 * 3 + 17, 1_300_000])
def test_rank_order_radix_sort_sizes_and_key_ranges(nm, n):
    """the hand-written radix sort behind nmod_rank_order (radix_sort.hpp) at its tile boundaries and over many tiles: keys over the
    whole float64 range (signs, infinities, denormals, NaN last), every byte of the image exercised, long runs of equal keys (stable)"""
    rng = np.random.default_rng(n)
    raw = rng.integers(0, 1 << 63, n, dtype=np.int64).view(np.float64)              # random bit patterns: every exponent, NaNs among them
    raw = np.where(rng.random(n) < 0.5, -raw, raw)
    k1 = np.where(rng.random(n) < 0.3, np.round(rng.normal(0, 1, n), 0), raw)       # a third: a handful of values, long equal runs
    k1[rng.random(n) < 0.01] = np.inf; k1[rng.random(n) < 0.01] = -np.inf
    k2 = rng.integers(0, 3, n).astype(np.float64) * 1e-310                          # denormals
    k3 = rng.normal(0, 1, n)
    img = np.where(np.isnan(k1), np.inf, k1)                                         # NaN sorts last, among themselves by the later keys
    nan = np.isnan(k1).astype(np.int8)
    exp = np.lexsort((k3, k2, img, nan))                                             # (lexsort: last key is primary; stable)
    got = nm.engine.rank_order_host(k1, k2, k3)
    assert np.array_equal(got, exp)


def test_argsort_keys_entry_point(nm):
    """nmod_argsort_keys (the grouping sort of nanomod_amd/simulate.py) against numpy's stable argsort: negative keys, long equal runs,
    host and device memory"""
    import torch
    rng = np.random.default_rng(8)
    for n in (1, 2049, 700_001):
        key = rng.integers(-5, 6, n).astype(np.int64) * (1 << 40) + rng.integers(0, 50, n)
        exp = np.argsort(key, kind='stable')
        got = nm.engine.argsort_device(torch.from_numpy(key).cuda())
        assert np.array_equal(got.cpu().numpy(), exp)
        out = np.empty(n, np.int32)
        L = nm._lib
        import ctypes as C
        prm = L.make_params(memspace=L.MEM_HOST)
        assert L.load().nmod_argsort_keys(C.byref(prm), n, key.ctypes.data, out.ctypes.data) == 0
        assert np.array_equal(out, exp)


@pytest.mark.parametrize('name', ['g50_stouffer', 'ragged_stouffer'])
def test_rank_order_matches_reference_ranking(nm, name):
    """the device ranking reproduces the reference's sorted_sign_test order on the golden numbers"""
    from nanomod_amd import cli
    exp, _ = H.load_expected(name)
    order = cli.rank_order(exp, 'stouffer', 'pv')
    assert np.array_equal(order, exp['sorted_index'])


@pytest.mark.parametrize('tag', ['w10_o0', 'w10_o1', 'w3_o1_A', 'w5_o0_ks', 'w4_o1_st'])
def test_region_rank_matches_reference(nm, tag):
    """RegionRankbyST=1 (myDetect.py:463-515) on the reference's own per-position numbers (golden): same ranked
    window centres in the same order (window keys and ranking on the device, nmod_region_rank)"""
    import nanomod_amd.detect as D
    z = np.load(os.path.join(H.GOLDEN, 'g50_regionrank_%s.npz' % tag))
    method = str(z['method'])
    exp, _ = H.load_expected('g50_' + method)
    recs = []
    for i in range(len(exp['pos'])):
        tests = [(exp['mwu_u'][i], exp['mwu_p'][i]), (exp['t_t'][i], exp['t_p'][i]), (exp['ks_d'][i], exp['ks_p'][i])]
        if method != 'ks':
            tests.append((exp['comb_st'][i], exp['comb_p'][i]))
        recs.append(((str(exp['chrom'][i]), str(exp['strand'][i]), int(exp['pos'][i]), str(exp['base'][i]),
                      int(exp['n0'][i]), int(exp['n1'][i])), tests))
    mo = {'sign_test': recs, 'window': int(z['window']), 'WindOvlp': int(z['WindOvlp']), 'percentile': float(z['percentile']),
          'NA': str(z['NA'])}
    ranked = D.region_rank(mo, 2 if method == 'ks' else 3, 1 if str(z['rankUse']) == 'pv' else 0)
    assert mo['window'] == int(z['window_after'])
    assert [r[0][2] for r in ranked] == list(z['pos'])
    assert [r[0][3] for r in ranked] == list(z['base'])


def test_full_size_properties(nm):
    """BASELINE.json's full size (4.6 M positions x 200 v 200) through size-independent properties: swapping the groups
    leaves D, U and every p-value unchanged and negates t; KS-only and all-tests mode agree on D to one rounding; the
    planted positions (and only a handful of others) reach the combined p-value the plant implies"""
    import torch
    L = nm._lib
    P, n = 4_600_000, 200
    dev = 'cuda:0'
    a = torch.empty(P * n, dtype=torch.float32, device=dev); b = torch.empty(P * n, dtype=torch.float32, device=dev)
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL)
    det.synth_fill(a, 7, 0, P, 0, n, 10000, 0.8); det.synth_fill(b, 7, 0, P, 1, n, 10000, 0.8)
    rid = torch.zeros(P, dtype=torch.int32, device=dev)
    r1 = {k: v.clone() for k, v in det.run(a, b, rid, stride0=n, stride1=n, npos=P).items()}
    r2 = det.run(b, a, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    # the rank statistics are integers: bit-equal under the swap.  The moments of a group are summed in a different
    # order when it is the sorted group (registers) and when it is the ranked one (ranking rounds): t and its p-value
    # agree to rounding (the same 1e-11 / 1e-9 gates as against the oracle), not bit for bit
    for k in ('ks_d', 'ks_p', 'mwu_u', 'mwu_p', 'comb_p', 'comb_st'):
        assert torch.equal(r1[k], r2[k]), k
    assert float(((r1['t_t'] + r2['t_t']).abs() / r1['t_t'].abs().clamp_min(1e-300)).max().item()) <= 1e-11
    assert float(((r1['t_p'] - r2['t_p']).abs() / r1['t_p'].clamp_min(1e-300)).max().item()) <= 1e-9
    assert int(r1['status'].max().item()) == 0
    ks = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS).run(a, b, rid, stride0=n, stride1=n, npos=P)
    torch.cuda.synchronize()
    assert float((ks['ks_d'] - r1['ks_d']).abs().max().item()) <= 0.0
    assert float(((ks['comb_p'] - r1['comb_p']).abs() / r1['comb_p']).max().item()) <= 1e-9
    planted = torch.arange(10000, P, 10000, device=dev)                 # (position 0 sits at the run edge: padded window, p = 1)
    assert float(r1['comb_p'][planted].max().item()) < 1e-10
    assert int((r1['comb_p'] < 1e-10).sum().item()) <= 5 * len(planted) + 10       # the window spreads a plant over its neighbours


@pytest.mark.parametrize('G', [2, 4, 8])
def test_logical_shards_equal_unsharded(nm, G):
    """the position partition + recomputed +-nb halo, G logical shards run one after the other through
    the HIP path on one device and reassembled, is bit-equal to the unsharded run (several runs cut by the shard edges)"""
    import torch
    from nanomod_amd import sharding
    L = nm._lib
    P, n, nb = 10007, 64, 3
    dev = 'cuda:0'
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    a = torch.empty(P * n, dtype=torch.float32, device=dev); b = torch.empty(P * n, dtype=torch.float32, device=dev)
    det.synth_fill(a, 3, 0, P, 0, n, 500, 0.8); det.synth_fill(b, 3, 0, P, 1, n, 500, 0.8)
    rid = torch.as_tensor((np.arange(P) // 1234).astype(np.int32), device=dev)
    full = {k: v.clone() for k, v in det.run(a, b, rid, stride0=n, stride1=n, npos=P).items()}
    got = {k: torch.empty_like(full[k]) for k in ('ks_p', 'comb_p', 'comb_st')}
    for r in range(G):
        lo, hi = sharding.shard_bounds(P, G, r)
        lo_h, hi_h = sharding.halo_bounds(lo, hi, nb, P)
        part = det.run(a[lo_h * n:hi_h * n], b[lo_h * n:hi_h * n], rid[lo_h:hi_h], stride0=n, stride1=n, npos=hi_h - lo_h)
        for k in got:
            got[k][lo:hi] = part[k][lo - lo_h:lo - lo_h + (hi - lo)]
    torch.cuda.synchronize()
    for k in got:
        assert torch.equal(got[k], full[k]), k


def test_mtest2_on_arbitrary_float64(nm):
    """the drop-in accepts signals that are neither float32-exact nor 3-dp rounded (the reference does): same numbers as
    the oracle through mtest2"""
    import nanomod_oracle as orc
    rng = np.random.default_rng(21)
    npos = 40
    ca = [rng.normal(0, 1, int(rng.integers(6, 90))) for _ in range(npos)]
    cb = [rng.normal(0.3, 1, int(rng.integers(6, 90))) for _ in range(npos)]
    with tempfile.TemporaryDirectory() as tmp:
        mo = {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
              'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 0, 'RegionRankbyST': 0,
              'outFolder': tmp, 'FileID': 'f64', 'MinCoverage': 5}
        for ds, ch in (('A', ca), ('B', cb)):
            mo[ds] = {'norm_mean': {('c', '+'): {10 + i: [np.float64(v) for v in ch[i]] for i in range(npos)}},
                      'base': {('c', '+'): {10 + i: 'A' for i in range(npos)}}, 'basedict': {}}
        nm.mfilter_coverage(mo)
        nm.mtest2(mo)
    for i, rec in enumerate(mo['sign_test']):
        e = orc.getKStest(ca[i], cb[i])
        assert rec[1][0][0] == e[0][0] and abs(rec[1][0][1] - e[0][1]) <= 1e-9 * e[0][1]
        assert abs(rec[1][1][1] - e[1][1]) <= 1e-9 * e[1][1] and abs(rec[1][2][0] - e[2][0]) <= 0.0
        assert abs(rec[1][2][1] - e[2][1]) <= 1e-9 * e[2][1]


def test_synth_fill_csr_matches_the_numpy_restatement(nm):
    """nmod_synth_fill_csr (bench.py --config ragged): sample `read` of position p is the fixed-stride generator's
    (p, read) value, written at off[p] + read; float32 and int16 output"""
    import torch
    L = nm._lib
    rng = np.random.default_rng(3)
    P, begin = 700, 9_990
    sizes = rng.integers(0, 300, P); sizes[5] = 0; sizes[17] = 1
    off = np.zeros(P + 1, np.int64); off[1:] = np.cumsum(sizes)
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
    d_off = torch.from_numpy(off).cuda()
    nmax = int(sizes.max())
    for tdt, name in ((torch.float32, 'f32'), (torch.int16, 'i16')):
        for g in (0, 1):
            out = torch.zeros(int(off[-1]), dtype=tdt, device='cuda:0')
            det.synth_fill_csr(out, 77, begin, d_off, g, 10000, 0.8)
            torch.cuda.synchronize()
            ref = H.synth_ref(77, begin, P, g, nmax, 10000, 0.8, dtype=name).reshape(P, nmax)
            exp = np.concatenate([ref[i, :sizes[i]] for i in range(P)])
            assert np.array_equal(out.cpu().numpy(), exp), (name, g)


@pytest.mark.parametrize('method', ['stouffer', 'ks'])
def test_cli_detect_on_read_folders_through_an_injected_reader(nm, method, capsys):
    """(f)2 as far as the image allows (h5py absent): `cli detect` on two FOLDERS of placeholder read files — walk,
    min_lr filter, strand-aware position mapping, CSR build, HIP kernels, table — through `--fast5Reader`, against
    (a) the table of the same reads handed over as .npz containers and (b) the oracle run on what the reference's
    own ReadAllFast5 built from these reads (tests/golden/fast5_expected_g*.npz)."""
    from nanomod_amd import cli, container, fast5_ingest
    import nanomod_amd.detect as D
    with tempfile.TemporaryDirectory() as tmp:
        dirs = [H.write_placeholder_reads(tmp, g) for g in (0, 1)]
        common = ['--FileID', 'x', '--testMethod', method, '--topN', '5', '--outLevel', '3', '--MinCoverage', '5']
        rc = cli.main(['detect', '--wrkBase1', dirs[0], '--wrkBase2', dirs[1], '--outFolder', os.path.join(tmp, 'a'),
                       '--fast5Reader', 'helpers:placeholder_reader'] + common)
        assert rc == 0
        got = open(os.path.join(tmp, 'a', 'x_sign_test.txt')).read()
        # (a) the container path on the same reads
        paths = []
        for g in (0, 1):
            c = fast5_ingest.ingest_folder(dirs[g], {'min_lr': 500, 'min_lr_nb': 0}, H.placeholder_reader, log=lambda *a: None)
            paths.append(os.path.join(tmp, 'g%d.npz' % g))
            container.save_group(paths[-1], c['chrom'], c['strand'], c['pos'], c['base'], c['off'], c['sig'])
        rc = cli.main(['detect', '--wrkBase1', paths[0], '--wrkBase2', paths[1], '--outFolder', os.path.join(tmp, 'b')] + common)
        assert rc == 0
        assert got == open(os.path.join(tmp, 'b', 'x_sign_test.txt')).read()
    assert got.count('\n') > 1000
    # (b) the oracle on the reference reader's view of the same reads
    import nanomod_oracle as orc
    e0, e1 = (dict(np.load(os.path.join(H.GOLDEN, 'fast5_expected_g%d.npz' % g))) for g in (0, 1))
    meta, sig0, off0, sig1, off1, rid = cli.select_positions(e0, e1, 5, 3, lambda *a: None)
    s0 = np.asarray(sig0, dtype=np.float64) * (1e-3 if np.asarray(sig0).dtype == np.int16 else 1.0)
    s1 = np.asarray(sig1, dtype=np.float64) * (1e-3 if np.asarray(sig1).dtype == np.int16 else 1.0)
    mcode = {'ks': orc.METHOD_KS, 'stouffer': orc.METHOD_STOUFFER}[method]
    out = orc.detect_batch(s0, off0, s1, off1, rid, 2, 2.0, mcode)
    lines = []
    for i in range(len(rid)):
        rec = [(out['mwu_u'][i], out['mwu_p'][i]), (out['t_t'][i], out['t_p'][i]), (out['ks_d'][i], out['ks_p'][i])]
        if method != 'ks':
            rec.append((out['comb_st'][i], out['comb_p'][i]))
        lines.append(orc.format_sign_test_line(str(meta['chrom'][i]), str(meta['strand'][i]), int(meta['pos'][i]), str(meta['base'][i]),
                                               int(meta['n0'][i]), int(meta['n1'][i]), rec, method != 'ks'))
    assert got == ''.join(lines)
    capsys.readouterr()


def test_ks_only_d_is_bit_exact_and_the_rational_flag_opts_out(nm):
    """KS-only mode (tests == NMOD_TEST_KS): by default D is ks_2samp's float form bit for bit, like the all-tests
    kernels; NMOD_FLAG_KS_RATIONAL_D reports the correctly rounded exact rational instead (<= 2 ulp away) and leaves the
    p-values within 1e-9.  Ragged sizes over every KS size class, continuous and tie-heavy data, device-resident."""
    import torch
    import oracle_c
    L = nm._lib
    rng = np.random.default_rng(77)
    P = 3000
    n0 = rng.integers(5, 900, P); n1 = rng.integers(5, 900, P)
    off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(n1)
    a = rng.normal(0, 1, off0[-1]).astype(np.float32); b = rng.normal(0.15, 1.1, off1[-1]).astype(np.float32)
    a[:off0[P // 2]] = np.round(a[:off0[P // 2]], 2); b[:off1[P // 2]] = np.round(b[:off1[P // 2]], 2)
    rid = np.zeros(P, np.int32)
    exp = oracle_c.detect_batch(a, off0, b, off1, rid, 2, 2.0, 'stouffer', tests=1, threads=0)
    t = lambda x: torch.from_numpy(x).cuda()
    da, db, o0, o1, r = t(a), t(b), t(off0), t(off1), t(rid)
    got = {}
    for name, flags in (('exact', 0), ('rational', L.FLAG_KS_RATIONAL_D)):
        det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, flags=flags)
        res = det.run(da, db, r, off0=o0, off1=o1, max_n0=900, max_n1=900)
        torch.cuda.synchronize()
        got[name] = {k: v.cpu().numpy() for k, v in res.items()}
    assert np.array_equal(got['exact']['ks_d'], exp['ks_d'])
    assert np.max(np.abs(got['rational']['ks_d'] - exp['ks_d'])) <= 4.5e-16
    assert np.any(got['rational']['ks_d'] != exp['ks_d'])                  # (the two forms do differ somewhere in 3000 positions)
    for name in got:
        H.assert_close_p(got[name]['ks_p'], exp['ks_p'], 1e-9, 'ks_p ' + name)
        H.assert_close_p(got[name]['comb_p'], exp['comb_p'], 1e-9, 'comb_p ' + name)
    # the flag is ignored as soon as another test is in the mask, and unknown flag bits are rejected
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL, flags=L.FLAG_KS_RATIONAL_D)
    res = det.run(da, db, r, off0=o0, off1=o1, max_n0=900, max_n1=900)
    assert np.array_equal(res['ks_d'].cpu().numpy(), exp['ks_d'])
    with pytest.raises(L.NanomodLibraryError):
        nm.DeviceDetector(0, tests=L.TEST_KS, flags=64).run(da, db, r, off0=o0, off1=o1, max_n0=900, max_n1=900)


@pytest.mark.parametrize('all_tests', [False, True])
def test_int16_packed_sort_edges(nm, all_tests):
    """The packed-int16 sorting network (16 keys per lane: smaller groups of 1..256 samples, both lane-group sizes) on the
    cases its pads and row loads can get wrong: every smaller-group size 1..260 (rows shorter than four samples, sizes
    that are not multiples of four, the class boundaries 64 / 128 / 256), samples EQUAL to the pad value 32767 and to
    -32768, constant groups, and tie-heavy data; fixed-stride and CSR; against the oracle (D and U exact)."""
    import oracle_c
    L = nm._lib
    rng = np.random.default_rng(2026)
    sizes = np.arange(1, 261)
    n0 = sizes.copy()
    n1 = rng.integers(1, 300, len(sizes)); n1[::7] = n0[::7]
    off0 = np.zeros(len(sizes) + 1, np.int64); off0[1:] = np.cumsum(n0)
    off1 = np.zeros(len(sizes) + 1, np.int64); off1[1:] = np.cumsum(n1)
    for mode in ('normal', 'extremes', 'coarse'):
        if mode == 'normal':
            a = np.rint(rng.normal(0, 1000, off0[-1])); b = np.rint(rng.normal(100, 1100, off1[-1]))
        elif mode == 'extremes':
            a = rng.choice([32767, 32766, -32768, 0, 5], off0[-1]).astype(np.float64)
            b = rng.choice([32767, -32768, -32767, 0, 7], off1[-1]).astype(np.float64)
        else:
            a = np.rint(rng.normal(0, 3, off0[-1])); b = np.rint(rng.normal(0.5, 3, off1[-1]))
        a = np.clip(a, -32768, 32767).astype(np.int16); b = np.clip(b, -32768, 32767).astype(np.int16)
        rid = np.zeros(len(sizes), np.int32)
        tests = 7 if all_tests else 1
        exp = oracle_c.detect_batch(a, off0, b, off1, rid, 2, 2.0, 'stouffer', tests=tests, threads=0)
        for swap in (False, True):
            args = (b, off1, a, off0) if swap else (a, off0, b, off1)
            got = nm.detect_host(*args, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL if all_tests else L.TEST_KS)
            assert np.array_equal(got['ks_d'], exp['ks_d']), (mode, swap)
            H.assert_close_p(got['ks_p'], exp['ks_p'], 1e-9, 'ks_p %s' % mode)
            if all_tests:
                ident = (exp['status'] & 1) != 0
                assert np.array_equal(got['mwu_u'][~ident], exp['mwu_u'][~ident]), (mode, swap)
                H.assert_close_p(got['mwu_p'][~ident], exp['mwu_p'][~ident], 1e-9, 'mwu_p %s' % mode)
                tt = exp['t_t'] * (-1.0 if swap else 1.0)
                H.assert_close_stat(got['t_t'], tt, 1e-11, 2e-14, 't_t %s' % mode)
    # fixed stride, sizes around the class boundaries and not multiples of four
    for n in (3, 63, 65, 127, 130, 199, 201, 255, 256):
        P = 64
        a = np.rint(rng.normal(0, 800, P * n)).astype(np.int16); b = np.rint(rng.normal(50, 900, P * n)).astype(np.int16)
        a[:n] = 32767; b[n:2 * n] = 32767
        o = np.arange(0, (P + 1) * n, n, dtype=np.int64)
        exp = oracle_c.detect_batch(a, o, b, o, np.zeros(P, np.int32), 2, 2.0, 'stouffer', tests=7 if all_tests else 1, threads=0)
        got = nm.detect_host(a, None, b, None, np.zeros(P, np.int32), nb=2, weights_dif=2.0, method='stouffer',
                             tests=L.TEST_ALL if all_tests else L.TEST_KS, stride0=n, stride1=n)
        assert np.array_equal(got['ks_d'], exp['ks_d']), n
        H.assert_close_p(got['comb_p'], exp['comb_p'], 1e-9, 'comb_p stride %d' % n)
        if all_tests:
            ident = (exp['status'] & 1) != 0
            assert np.array_equal(got['mwu_u'][~ident], exp['mwu_u'][~ident]), n
