"""Shared test helpers: fixture loading, the reference-shaped `moptions`, comparisons, and the
numpy restatement of the device synthetic generator (include/nanomod_hip.h: nmod_synth_fill)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_inputs(name):
    return dict(np.load(os.path.join(GOLDEN, name + '_inputs.npz')))


def load_expected(name):
    exp = dict(np.load(os.path.join(GOLDEN, name + '_expected.npz')))
    with open(os.path.join(GOLDEN, name + '_sign_test.txt')) as f:
        table = f.read()
    return exp, table


def build_moptions(fx, outdir, file_id, nb, wdif, method, min_cov=5, mstd=0):
    """The in-memory structure the reference's ReadAllFast5 builds (myDetect.py:562-572,124)."""
    mo = {'ds2': ['grpA', 'grpB'], 'outLevel': 3, 'mstd': mstd, 'coverages': [0, 0],
          'downsampling': 100, 'downsampling_quantile': 0.25, 'neighborPvalues': nb, 'WeightsDif': wdif,
          'testMethod': method, 'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
          'outFolder': outdir, 'FileID': file_id, 'MinCoverage': min_cov}
    for g, ds in enumerate(mo['ds2']):
        d = {'norm_mean': {}, 'base': {}, 'basedict': {}}
        sig = fx['sig%d' % g].astype(np.float64)
        off = fx['off%d' % g]
        for i in range(len(off) - 1):
            if off[i + 1] == off[i]:
                continue
            sk = (str(fx['chrom'][i]), str(fx['strand'][i]))
            pk = int(fx['pos'][i])
            d['norm_mean'].setdefault(sk, {})[pk] = [np.float64(v) for v in sig[off[i]:off[i + 1]]]
            d['base'].setdefault(sk, {})[pk] = str(fx['base%d' % g][i])
            d['basedict'].setdefault(sk, {})[pk] = {str(fx['base%d' % g][i]): 1}
        mo[ds] = d
    return mo


def assert_close_p(got, exp, rel=1e-9, name='p'):
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    assert got.shape == exp.shape, (name, got.shape, exp.shape)
    nan_g, nan_e = np.isnan(got), np.isnan(exp)
    assert np.array_equal(nan_g, nan_e), '%s: NaN pattern differs' % name
    inf_e = np.isinf(exp)
    assert np.array_equal(got[inf_e], exp[inf_e]), '%s: infinities differ' % name
    m = ~(nan_e | inf_e)
    err = np.abs(got[m] - exp[m])
    tol = rel * np.abs(exp[m]) + 1e-300
    bad = err > tol
    assert not bad.any(), '%s: %d of %d beyond rel %g; worst %g at %d (got %r exp %r)' % (
        name, bad.sum(), m.sum(), rel, (err / np.maximum(np.abs(exp[m]), 1e-300)).max(),
        int(np.argmax(err / np.maximum(np.abs(exp[m]), 1e-300))),
        got[m][np.argmax(err / np.maximum(np.abs(exp[m]), 1e-300))],
        exp[m][np.argmax(err / np.maximum(np.abs(exp[m]), 1e-300))])
    # the north-star bar: 1e-6 absolute
    assert (err <= 1e-6).all()


def assert_close_stat(got, exp, rel=1e-12, abs_=4e-16, name='stat'):
    got = np.asarray(got, dtype=np.float64)
    exp = np.asarray(exp, dtype=np.float64)
    nan_e = np.isnan(exp)
    assert np.array_equal(np.isnan(got), nan_e), '%s: NaN pattern differs' % name
    inf_e = np.isinf(exp)
    assert np.array_equal(got[inf_e], exp[inf_e]), '%s: infinities differ' % name
    m = ~(nan_e | inf_e)
    err = np.abs(got[m] - exp[m])
    tol = abs_[m] if isinstance(abs_, np.ndarray) else abs_
    assert (err <= rel * np.abs(exp[m]) + tol).all(), '%s: worst abs err %g' % (name, err.max())


def t_abs_gate(sig0, off0, sig1, off1):
    """Absolute tolerance of the Welch statistic per position: t = (mean0 - mean1) / se, and ANY order of summation leaves
    a mean within a few ulp of its value, so t is only defined to ~ulp(max |mean|) / se.  Rows that share a large level
    (event-like input: |mean| ~ 3 units, se ~ 0.01) move t by 1e-13 per ulp; rows around zero by 1e-16.  Gate: 2e-14 +
    6 ulp(max |mean|) / se (the relative part, 1e-11 |t|, is added by the caller)."""
    sc = 1e-3 if np.asarray(sig0).dtype == np.int16 else 1.0
    out = []
    for sig, off in ((sig0, off0), (sig1, off1)):
        x = np.asarray(sig, dtype=np.float64) * sc
        n = np.diff(off).astype(np.float64)
        idx = np.minimum(off[:-1], max(len(x) - 1, 0))
        s1 = np.add.reduceat(x, idx) if len(x) else np.zeros(len(n))
        s2 = np.add.reduceat(x * x, idx) if len(x) else np.zeros(len(n))
        s1 = np.where(n > 0, s1, 0.0); s2 = np.where(n > 0, s2, 0.0)
        with np.errstate(divide='ignore', invalid='ignore'):
            mean = s1 / n
            var = np.maximum(s2 - s1 * mean, 0.0) / np.maximum(n - 1.0, 1.0)
        out.append((np.abs(mean), var / n))
    with np.errstate(divide='ignore', invalid='ignore'):
        se = np.sqrt(out[0][1] + out[1][1])
        g = 2e-14 + 6.0 * np.spacing(np.maximum(out[0][0], out[1][0])) / se
    return np.where(np.isfinite(g), g, 2e-14)


def compare_outputs(got, exp, with_comb=True, p_rel=1e-9, t_abs=2e-14):
    assert np.array_equal(np.asarray(got['mwu_u']), np.asarray(exp['mwu_u'])), 'MWU U must be exact'
    assert_close_p(got['mwu_p'], exp['mwu_p'], p_rel, 'mwu_p')
    # t = (mean0 - mean1) / se: one ulp of a sample in a mean (1e-16 at |x| ~ 1) moves t by ~1e-15 / se absolute, and the
    # two sides sum in different orders — near t = 0 only the absolute error is meaningful (t_abs: see t_abs_gate)
    assert_close_stat(got['t_t'], exp['t_t'], 1e-11, t_abs, 't_t')
    assert_close_p(got['t_p'], exp['t_p'], p_rel, 't_p')
    assert_close_stat(got['ks_d'], exp['ks_d'], 0, 0.0, 'ks_d')
    assert_close_p(got['ks_p'], exp['ks_p'], p_rel, 'ks_p')
    if with_comb:
        assert_close_stat(got['comb_st'], exp['comb_st'], 1e-9, 1e-12, 'comb_st')
        assert_close_p(got['comb_p'], exp['comb_p'], p_rel, 'comb_p')


# ---- numpy restatement of the device synthetic generator
def synth_ref(seed, pos_begin, npos, group, n_per_pos, plant_period=0, plant_shift=0.0, dtype='f32'):
    with np.errstate(over='ignore'):
        pos = (np.arange(npos, dtype=np.int64) + pos_begin)[:, None]
        read = np.arange(n_per_pos, dtype=np.uint64)[None, :]
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (pos * 2 + group).astype(np.uint64)
        x = x ^ (read * np.uint64(0xD1B54A32D192ED03))
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    m = np.uint64(0xffff)
    s = ((x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))).astype(np.int64)
    v = (s - 131070).astype(np.float32) * np.float32(2.6428997e-05)
    if group == 1 and plant_period > 0:
        mm = (pos % plant_period)
        planted = ((mm == 0) | (mm == 1) | (mm == plant_period - 1))
        v = np.where(planted, v + np.float32(plant_shift), v).astype(np.float32)
    if dtype == 'i16':
        return np.rint(v * np.float32(1000.0)).astype(np.int16).reshape(-1)
    return v.reshape(-1)


def _mix64(seed, pos, group, read):
    with np.errstate(over='ignore'):
        x = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * (pos * 2 + group).astype(np.uint64)
        x = x ^ (read * np.uint64(0xD1B54A32D192ED03))
        x = x ^ (x >> np.uint64(30)); x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27)); x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def synth_events_ref(seed, pos_begin, npos, group, n_per_pos, plant_period=0, plant_shift_milli=0, spread_milli=200, dtype='f32', outlier_permille=0):
    """numpy restatement of nmod_synth_fill_events (include/nanomod_hip.h) as an [npos, n_per_pos] array: a level per position
    (both groups), reads spread around it, everything on the milli-unit grid"""
    pos = (np.arange(npos, dtype=np.int64) + pos_begin)[:, None]
    read = np.arange(n_per_pos, dtype=np.uint64)[None, :]
    lev = (_mix64(np.uint64(seed) ^ np.uint64(0xA5A5A5A5DEADBEEF), pos, 0, np.zeros((1, 1), np.uint64)) >> np.uint64(40)) % np.uint64(6001)
    lev = lev.astype(np.int64) - 3000
    x = _mix64(seed, pos, group, read)
    m = np.uint64(0xffff)
    z = ((x & m) + ((x >> np.uint64(16)) & m) + ((x >> np.uint64(32)) & m) + (x >> np.uint64(48))).astype(np.int64) - 131070
    k = lev + np.floor_divide(2 * z * int(spread_milli) + 37837, 75674)
    if group == 1 and plant_period > 0:
        mm = pos % plant_period
        k = k + np.where((mm == 0) | (mm == 1) | (mm == plant_period - 1), int(plant_shift_milli), 0)
    if outlier_permille > 0:       # a mis-segmented event: uniform over the +-5 unit clip range, whatever the level
        o = _mix64(np.uint64(seed) ^ np.uint64(0x0DDBA11C0FFEE123), pos, group, read)
        hit = ((o >> np.uint64(20)) % np.uint64(1000)).astype(np.int64) < int(outlier_permille)
        k = np.where(hit, ((o >> np.uint64(32)) % np.uint64(10001)).astype(np.int64) - 5000, k)
    k = np.clip(k, -32767, 32767)
    if dtype == 'i16':
        return k.astype(np.int16)
    return (k.astype(np.float64) / 1000.0).astype(np.float32)


# ---- placeholder read files for the FAST5 ingest ((f)2): one .npz per read with what the HDF5 reader would return
def write_placeholder_reads(root, group):
    """the reads of tests/golden/fast5_reads.npz (made by oracle/gen_golden.py; the reference's own ReadAllFast5 was run
    over the same reads through a stub h5py) as files under root/grp<group>/, in the fixture's folder layout"""
    z = np.load(os.path.join(GOLDEN, 'fast5_reads.npz'))
    for i in np.flatnonzero(z['group'] == group):
        path = os.path.join(root, 'grp%d' % group, str(z['rel'][i]))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        a, b = z['off'][i], z['off'][i + 1]
        with open(path, 'wb') as f:
            np.savez(f, chrom=z['chrom'][i], strand=z['strand'][i], start=z['start'][i], norm_mean=z['norm_mean'][a:b],
                     base=z['base'][a:b], has_align=z['has_align'][i])
    return os.path.join(root, 'grp%d' % group)


def placeholder_reader(path):
    """stands in for fast5_ingest.h5py_reader (h5py is not in this image): same return value, None without alignment"""
    r = np.load(path)
    if not bool(r['has_align']):
        return None
    return str(r['chrom']), int(r['start']), str(r['strand']), r['norm_mean'], r['base']
