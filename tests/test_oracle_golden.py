"""CPU: the oracle (Python restatement) against the golden vectors produced by the reference's own
glue code (oracle/gen_golden.py) and against the known-answer anchors of SURVEY.md §8c."""
import json
import math
import os
import tempfile

import numpy as np
import pytest

import helpers as H
import nanomod_oracle as orc

CASES = [('g50', 'g50_stouffer', 2, 2.0, 'stouffer'), ('g50', 'g50_fisher', 2, 2.0, 'fisher'),
         ('g50', 'g50_ks', 2, 2.0, 'ks'), ('ragged', 'ragged_stouffer', 2, 2.0, 'stouffer'),
         ('ragged', 'ragged_fisher', 2, 2.0, 'fisher'), ('ties', 'ties_stouffer', 2, 2.0, 'stouffer'),
         ('sweep', 'sweep_nb0_w2_stouffer', 0, 2.0, 'stouffer'), ('sweep', 'sweep_nb1_w1_stouffer', 1, 1.0, 'stouffer'),
         ('sweep', 'sweep_nb3_w3_stouffer', 3, 3.0, 'stouffer'), ('sweep', 'sweep_nb3_w2_fisher', 3, 2.0, 'fisher')]
# round 5: wide windows (the ABI's limit is 64) on a track of several runs; MinCoverage 3 / 20 (sixth field)
CASES += [('track600', 'track600_nb%d_%s' % (nb, m), nb, 2.0, m) for nb in (5, 16, 64) for m in ('stouffer', 'fisher')]
CASES += [('ragged', 'ragged_mc3', 2, 2.0, 'stouffer', 3), ('ragged', 'ragged_mc20', 2, 2.0, 'stouffer', 20)]
METHOD = {'ks': orc.METHOD_KS, 'stouffer': orc.METHOD_STOUFFER, 'fisher': orc.METHOD_FISHER}


def test_kat_anchors():
    with open(os.path.join(H.GOLDEN, 'kat.json')) as f:
        kat = json.load(f)
    # SURVEY.md §8c values, independent of the fixture file
    assert orc.getKStest([-1.2, -0.5, 0.0, 0.3, 0.3, 0.9, 1.4], [-0.1, 0.3, 0.8, 1.1, 1.5, 2.0]) == [
        (11.0, 0.08617969948071841), (-1.6840425969226844, 0.12040158126178833), (0.380952380952381, 0.6208327763555257)]
    for key in ('KAT-1', 'KAT-2'):
        got = orc.getKStest(kat[key]['a'], kat[key]['b'])
        for g, e in zip(got, kat[key]['out']):
            assert g[0] == e[0] and abs(g[1] - e[1]) <= 2e-16 * e[1] + 1e-300
    for method in ('stouffer', 'fisher'):
        k = kat['KAT-3-' + method]
        st, pv = orc.combine_track(np.zeros(7), np.array(k['ks_p']), np.zeros(7, np.int32), k['nb'], k['dif'], METHOD[method])
        for i, e in enumerate(k['out']):
            e = [float(v) for v in e]
            assert (st[i] == e[0] or abs(st[i] - e[0]) <= 1e-14 * abs(e[0])) and abs(pv[i] - e[1]) <= 1e-13 * e[1]
    z, p = orc.combine_stouffer(kat['KAT-4']['window'], kat['KAT-4']['weights'])
    assert abs(z - 4.1055560585756705) < 1e-14 and abs(p - 2.016717252680804e-05) < 1e-18
    x, p = orc.combine_fisher(kat['KAT-4']['window'])
    assert abs(x - 34.85485794188416) < 1e-13 and abs(p - 0.0001321894341105391) < 1e-17
    assert kat['EDGE-identical']['raises']
    with pytest.raises(orc.AllIdenticalError):
        orc.mannwhitneyu([0.25] * 6, [0.25] * 6)
    assert orc.ks_2samp([0.25] * 6, [0.25] * 6) == (0.0, 1.0)
    t, p = orc.ttest_welch([0.25] * 6, [0.25] * 6)
    assert math.isnan(t) and math.isnan(p)
    assert kat['constants']['DBL_MIN'] == orc.DBL_MIN and abs(kat['constants']['isf_DBL_MIN'] - 37.5193793471445) < 1e-12


@pytest.mark.parametrize('case', CASES, ids=[c[1] for c in CASES])
def test_oracle_reproduces_reference_tables(case):
    """position set/order (host logic), every number, and the formatted table"""
    import nanomod_amd.detect as D
    inp, name, nb, wdif, method = case[:5]
    fx = H.load_inputs(inp)
    exp, table = H.load_expected(name)
    mo = H.build_moptions(fx, tempfile.gettempdir(), name, nb, wdif, method, min_cov=case[5] if len(case) > 5 else 5)
    D.mfilter_coverage(mo)
    meta, sig0, off0, sig1, off1, rid = D.build_csr(mo)
    assert list(meta['chrom']) == list(exp['chrom']) and list(meta['pos']) == list(exp['pos'])
    assert list(meta['base']) == list(exp['base'])
    meta = list(zip(meta['chrom'], meta['strand'], meta['pos'], meta['base'], meta['n0'], meta['n1']))
    out = orc.detect_batch(sig0, off0, sig1, off1, rid, nb, wdif, METHOD[method])
    with_comb = method != 'ks'
    H.compare_outputs(out, exp, with_comb, p_rel=1e-12)
    assert np.array_equal(out['ks_d'], exp['ks_d'])
    lines = []
    for i, m in enumerate(meta):
        rec = [(out['mwu_u'][i], out['mwu_p'][i]), (out['t_t'][i], out['t_p'][i]), (out['ks_d'][i], out['ks_p'][i])]
        if with_comb:
            rec.append((out['comb_st'][i], out['comb_p'][i]))
        lines.append(orc.format_sign_test_line(m[0], m[1], m[2], m[3], m[4], m[5], rec, with_comb and nb > 0))
    assert ''.join(lines) == table


def test_ks_counts_is_the_exact_numerator():
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = np.round(rng.normal(0, 1, rng.integers(3, 90)), 2)
        b = np.round(rng.normal(0.3, 1, rng.integers(3, 90)), 2)
        d, _ = orc.ks_2samp(a, b)
        assert abs(d - orc.ks_counts(a, b) / (len(a) * len(b))) <= 2.3e-16


@pytest.mark.parametrize('inp', ['sweep', 'ragged'])
def test_oracle_mstd_against_the_reference(inp):
    """--mstd (myDetect.py:437-438): np.mean / np.std (ddof = 0) per group, as the reference's mtest2 recorded them"""
    import nanomod_amd.detect as D
    fx = H.load_inputs(inp)
    exp, _ = H.load_expected(inp + '_mstd')
    mo = H.build_moptions(fx, tempfile.gettempdir(), inp, 2, 2.0, 'stouffer')
    D.mfilter_coverage(mo)
    meta, sig0, off0, sig1, off1, rid = D.build_csr(mo)
    assert list(meta['pos']) == list(exp['pos'])
    for g, (sig, off) in enumerate(((sig0, off0), (sig1, off1))):
        x = np.asarray(sig, dtype=np.float64) * (1e-3 if sig.dtype == np.int16 else 1.0)
        mean = np.array([np.mean(x[off[i]:off[i + 1]]) for i in range(len(off) - 1)])
        std = np.array([np.std(x[off[i]:off[i + 1]]) for i in range(len(off) - 1)])
        assert np.all(np.abs(mean - exp['mean%d' % g]) <= 1e-13 * np.abs(exp['mean%d' % g]) + 1e-16)
        assert np.all(np.abs(std - exp['std%d' % g]) <= 1e-13 * exp['std%d' % g])
    # the file has one line per tested position, 0-based positions (the table's are 1-based)
    rows = open(os.path.join(H.GOLDEN, inp + '_mstd_meanstd.cvs')).read().strip().split('\n')
    assert len(rows) == len(exp['pos']) and [int(r.split(' ')[2]) for r in rows] == list(exp['pos'])
