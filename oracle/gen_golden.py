#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING THE REFERENCE'S OWN GLUE CODE here.

TEST INFRASTRUCTURE ONLY; runs only in the build container (needs
/root/reference).  Nothing produced here contains reference source: the
outputs are inputs + expected numbers + expected `_sign_test.txt` tables.

How the reference is executed:
  * fine-grained: lines 317-414 of /root/reference/bin/scripts/myDetect.py
    (m_min_float, m_max_float, getKStest, pos_check, combin_pvalues,
    get_combin_pvalue) are valid Python 3 and are exec'd verbatim from the
    reference checkout at generation time;
  * table-level: a throw-away lib2to3 conversion of myDetect.py in a temp
    dir, with stub modules for h5py / rpy2 / pkg_resources / myFast5, runs
    mfilter_coverage + mtest2 (+ save_test) on synthetic `moptions`.
The four scipy names the glue calls are bound as follows:
  ttest_ind, combine_pvalues -> the container's real scipy (unchanged
      semantics since the pinned 1.2.1);
  ks_2samp, mannwhitneyu     -> the scipy-1.2.1 restatements in
      oracle/nanomod_oracle.py (container scipy 1.15.3 changed their defaults),
      cross-checked below against scipy 1.15.3 where the semantics coincide.
"""
import io
import json
import os
import sys
import tempfile
import types
import contextlib

import numpy as np
import scipy.stats

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import nanomod_oracle as orc  # noqa: E402

REF = '/root/reference/bin/scripts'
OUT = os.path.join(HERE, '..', 'tests', 'golden')


# ---------------------------------------------------------------- scipy names
def _ks_2samp(a, b):
    return orc.ks_2samp(a, b)


def _mannwhitneyu(a, b):
    return orc.mannwhitneyu(a, b)


def crosscheck_restatements(rng):
    """ks_2samp D == scipy 1.15.3 D; mannwhitneyu == two-sided asymptotic / 2
    and min(U1, U2) (away from the 1.15.3 clip at p=1)."""
    for _ in range(300):
        n0, n1 = rng.integers(5, 300, size=2)
        a = np.round(rng.normal(0, 1, n0), 3)
        b = np.round(rng.normal(0.2, 1.1, n1), 3)
        d, _p = orc.ks_2samp(a, b)
        r = scipy.stats.ks_2samp(a, b, method='asymp')
        assert abs(d - r.statistic) < 1e-15, (d, r.statistic)
        # the 1.2.1 p-value expression, verbatim, with the container's kstwobign (unchanged since 1.2.1); the mpmath
        # series and the p < 1e-100 / D -> 0 cases are in tests/test_oracle_pins.py
        en = np.sqrt(n0 * n1 / float(n0 + n1))
        assert _p == float(scipy.stats.kstwobign.sf((en + 0.12 + 0.11 / en) * d)), (_p, d)
        u, p = orc.mannwhitneyu(a, b)
        r = scipy.stats.mannwhitneyu(a, b, use_continuity=True,
                                     alternative='two-sided', method='asymptotic')
        assert u == min(r.statistic, n0 * n1 - r.statistic)
        if r.pvalue < 1.0:
            assert abs(p - r.pvalue / 2) <= 1e-13 * p, (p, r.pvalue / 2)
        t, pt = orc.ttest_welch(a, b)
        r = scipy.stats.ttest_ind(a, b, equal_var=False)
        assert abs(t - r.statistic) <= 1e-14 * abs(t) and abs(pt - r.pvalue) <= 1e-13 * pt


# ------------------------------------------------------- fine-grained glue
def load_reference_glue():
    with open(os.path.join(REF, 'myDetect.py')) as f:
        lines = f.readlines()
    src = ''.join(lines[316:414])          # file lines 317..414
    ns = {'sys': sys, 'np': np,
          'mannwhitneyu': _mannwhitneyu, 'ttest_ind': scipy.stats.ttest_ind,
          'ks_2samp': _ks_2samp, 'combine_pvalues': scipy.stats.combine_pvalues}
    exec(compile(src, 'myDetect.py[317:414]', 'exec'), ns)
    return ns


# --------------------------------------------------------- table-level run
def load_converted_reference(tmpdir):
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    rt = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    for name in ('myDetect.py', 'myCom.py'):
        with open(os.path.join(REF, name)) as f:
            src = f.read()
        if not src.endswith('\n'):
            src += '\n'
        with open(os.path.join(tmpdir, name), 'w') as f:
            f.write(str(rt.refactor_string(src, name)))
    # stubs for modules absent in this image (never touched by mtest2)
    for modname in ('h5py', 'rpy2', 'rpy2.robjects', 'rpy2.robjects.packages',
                    'pkg_resources', 'myFast5'):
        m = types.ModuleType(modname)
        sys.modules[modname] = m
    sys.modules['rpy2'].robjects = sys.modules['rpy2.robjects']
    sys.modules['rpy2.robjects'].packages = sys.modules['rpy2.robjects.packages']
    sys.modules['rpy2.robjects.packages'].importr = lambda *a, **k: None
    sys.modules['pkg_resources'].resource_string = lambda *a, **k: ''
    sys.path.insert(0, tmpdir)
    import myDetect
    myDetect.mannwhitneyu = _mannwhitneyu
    myDetect.ks_2samp = _ks_2samp
    return myDetect


def build_moptions(fx, outdir, file_id, nb, wdif, method, min_cov=5, mstd=0):
    """The in-memory structure ReadAllFast5 builds (myDetect.py:562-572,124)."""
    mo = {'ds2': ['grpA', 'grpB'], 'outLevel': 3, 'mstd': mstd,
          'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
          'neighborPvalues': nb, 'WeightsDif': wdif, 'testMethod': method,
          'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
          'outFolder': outdir, 'FileID': file_id, 'MinCoverage': min_cov,
          'window': 10, 'WindOvlp': 0, 'percentile': 0.2}
    for g, ds in enumerate(mo['ds2']):
        d = {'norm_mean': {}, 'base': {}, 'basedict': {}}
        sig = fx['sig%d' % g].astype(np.float64)       # fp32-exact values, up-cast
        off = fx['off%d' % g]
        for i in range(len(off) - 1):
            sk = (str(fx['chrom'][i]), str(fx['strand'][i]))
            pk = int(fx['pos'][i])
            if off[i + 1] == off[i]:
                continue                                   # position absent in this group
            d['norm_mean'].setdefault(sk, {})[pk] = [np.float64(v) for v in sig[off[i]:off[i + 1]]]
            d['base'].setdefault(sk, {})[pk] = str(fx['base%d' % g][i])
            d['basedict'].setdefault(sk, {})[pk] = {str(fx['base%d' % g][i]): 1}
        mo[ds] = d
    return mo


# ------------------------------------------------------------ FAST5 ingest through the reference's reader
class _FakeH5File:
    """What the reference touches of an h5py.File (myDetect.py:42-72, myFast5.py:92-126), served from a
    placeholder file written by make_fast5_reads()."""
    def __init__(self, fn, mode='r'):
        z = np.load(fn, allow_pickle=False)
        self._has_align = bool(z['has_align'])
        ev = np.zeros(len(z['norm_mean']), dtype=[('norm_mean', '<f8'), ('norm_stdev', '<f8'), ('start', '<u4'),
                                                 ('length', '<u4'), ('base', 'U1')])
        ev['norm_mean'] = z['norm_mean']; ev['base'] = z['base']
        self._events = types.SimpleNamespace(value=ev)
        self._align = types.SimpleNamespace(attrs={'mapped_chrom': str(z['chrom']), 'mapped_start': int(z['start']),
                                                   'mapped_strand': str(z['strand'])})
    def __contains__(self, key):
        return ('Alignment' in key and self._has_align) or key.endswith('/Events')
    def __getitem__(self, key):
        return self._align if key.endswith('/Alignment') else self._events


def make_fast5_reads(rng):
    """Synthetic resquiggled reads of two groups over two chromosomes / both strands; some too short
    (min_lr filter), one without alignment, nested sub-folders and a 'mall' folder that must be skipped."""
    reads = []
    for g in (0, 1):
        for k in range(60):
            chrom = 'chrA' if k % 3 else 'chrB'
            strand = '+' if k % 2 else '-'
            start = int(rng.integers(0, 400))
            n = int(rng.integers(20, 120)) if k % 10 == 0 else int(rng.integers(500, 700))
            nm = np.round(rng.normal(0.3 * g if (k % 7 == 0) else 0.0, 1.0, n), 3)
            ref = np.array(list('ACGT'))[(np.arange(start, start + n) * 7 + (1 if chrom == 'chrB' else 0)) % 4]
            base = ref if strand == '+' else ref[::-1]
            sub = '' if k % 4 else ('sub%d/' % (k % 3)) + ('deep/' if k % 8 == 0 else '')
            if k == 13:
                sub = 'mall/'
            reads.append(dict(group=g, rel='%sread_%d_%d.fast5' % (sub, g, k), chrom=chrom, strand=strand, start=start,
                              norm_mean=nm, base=base, has_align=(k != 21)))
    return reads


def write_fast5_placeholders(reads, root):
    for r in reads:
        path = os.path.join(root, 'grp%d' % r['group'], r['rel'])
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'wb') as f:
            np.savez(f, chrom=r['chrom'], strand=r['strand'], start=r['start'], norm_mean=r['norm_mean'], base=r['base'],
                     has_align=r['has_align'])


def run_reference_ingest(myDetect, reads):
    sys.modules['h5py'].File = _FakeH5File
    myDetect.h5py = sys.modules['h5py']
    import myFast5 as _unused  # noqa: F401  (stub module; the paths below are what the reference defines)
    mf = sys.modules['myFast5']
    mf.rawAlignment_full = '/Analyses/NanomoCorrected_000/BaseCalled_template/Alignment'
    mf.ReadMapInfoInRef = lambda f5: [f5[mf.rawAlignment_full].attrs['mapped_chrom'], int(f5[mf.rawAlignment_full].attrs['mapped_start']),
                                      f5[mf.rawAlignment_full].attrs['mapped_strand']]
    mf.ReadNanoraw_events = lambda f5: f5['/Analyses/NanomoCorrected_000/BaseCalled_template/Events'].value
    out = {}
    with tempfile.TemporaryDirectory() as root:
        write_fast5_placeholders(reads, root)
        mo = {'wrkBase1': os.path.join(root, 'grp0'), 'wrkBase2': os.path.join(root, 'grp1'), '.fast5': '.fast5',
              'outLevel': 3, 'min_lr': 500, 'min_lr_nb': 0, 'window': 10}
        with contextlib.redirect_stdout(io.StringIO()):
            myDetect.ReadAllFast5(mo)
        for g, key in enumerate(mo['ds2']):
            ds = mo[key]
            chrom, strand, pos, base, chunks = [], [], [], [], []
            for sk in sorted(ds['norm_mean'].keys()):
                for pk in sorted(ds['norm_mean'][sk].keys()):
                    chrom.append(sk[0]); strand.append(sk[1]); pos.append(pk); base.append(ds['base'][sk][pk])
                    chunks.append(np.sort(np.asarray(ds['norm_mean'][sk][pk], dtype=np.float64)))
            off = np.zeros(len(pos) + 1, dtype=np.int64); off[1:] = np.cumsum([len(c) for c in chunks])
            out[g] = dict(chrom=np.array(chrom), strand=np.array(strand), pos=np.array(pos, dtype=np.int64),
                          base=np.array(base), off=off, sig=np.concatenate(chunks))
    return out


def run_reference_region_rank(myDetect, fx, method, window, wind_ovlp, percentile, na, rank_use='pv'):
    """mtest2 with RegionRankbyST=1 (myDetect.py:463-515): returns the ranked window centres."""
    with tempfile.TemporaryDirectory() as outdir:
        mo = build_moptions(fx, outdir, 'rr', 2, 2.0, method)
        mo.update({'RegionRankbyST': 1, 'window': window, 'WindOvlp': wind_ovlp, 'percentile': percentile, 'NA': na,
                   'rankUse': rank_use, 'SaveTest': 0})
        with contextlib.redirect_stdout(io.StringIO()):
            myDetect.mfilter_coverage(mo)
            myDetect.mtest2(mo)
    sst = mo['sorted_sign_test']
    return {'chrom': np.array([r[0][0] for r in sst]), 'strand': np.array([r[0][1] for r in sst]),
            'pos': np.array([r[0][2] for r in sst], dtype=np.int64), 'base': np.array([r[0][3] for r in sst]),
            'window_after': np.int64(mo['window'])}


def run_reference_table(myDetect, fx, nb, wdif, method, file_id, min_cov=5, rank_use='pv', mstd=0):
    meanstd = None
    with tempfile.TemporaryDirectory() as outdir:
        mo = build_moptions(fx, outdir, file_id, nb, wdif, method, min_cov, mstd)
        mo['rankUse'] = rank_use
        with contextlib.redirect_stdout(io.StringIO()):
            myDetect.mfilter_coverage(mo)
            myDetect.mtest2(mo)
        with open(os.path.join(outdir, file_id + '_sign_test.txt')) as f:
            table = f.read()
        if mstd:
            with open(os.path.join(outdir, file_id + '_meanstd.cvs')) as f:
                meanstd = f.read()
    st = mo['sign_test']
    exp = {
        'chrom': np.array([r[0][0] for r in st]), 'strand': np.array([r[0][1] for r in st]),
        'pos': np.array([r[0][2] for r in st], dtype=np.int64),
        'base': np.array([r[0][3] for r in st]),
        'n0': np.array([r[0][4] for r in st], dtype=np.int64),
        'n1': np.array([r[0][5] for r in st], dtype=np.int64),
        'mwu_u': np.array([r[1][0][0] for r in st]), 'mwu_p': np.array([r[1][0][1] for r in st]),
        't_t': np.array([r[1][1][0] for r in st]), 't_p': np.array([r[1][1][1] for r in st]),
        'ks_d': np.array([r[1][2][0] for r in st]), 'ks_p': np.array([r[1][2][1] for r in st]),
    }
    if len(st) and len(st[0][1]) > 3:
        exp['comb_st'] = np.array([r[1][3][0] for r in st])
        exp['comb_p'] = np.array([r[1][3][1] for r in st])
    # sorted order (myDetect.py:460) as indices into sign_test
    index_of = {id(r): i for i, r in enumerate(st)}
    exp['sorted_index'] = np.array([index_of[id(r)] for r in mo['sorted_sign_test']], dtype=np.int64)
    if mstd:
        # sign_test_mstd (myDetect.py:437-438): {(chrom, strand, pos): [[mean0, std0], [mean1, std1]]}, in sign_test order
        ms = [mo['sign_test_mstd'][(r[0][0], r[0][1], r[0][2])] for r in st]
        assert len(mo['sign_test_mstd']) == len(st)
        exp['mean0'] = np.array([m[0][0] for m in ms]); exp['std0'] = np.array([m[0][1] for m in ms])
        exp['mean1'] = np.array([m[1][0] for m in ms]); exp['std1'] = np.array([m[1][1] for m in ms])
        return exp, table, meanstd
    return exp, table


# ----------------------------------------------------------- fixture inputs
def _csr(chunks):
    off = np.zeros(len(chunks) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(c) for c in chunks])
    sig = np.concatenate(chunks).astype(np.float32) if len(chunks) else np.zeros(0, np.float32)
    return sig, off


def make_g50(rng):
    """2 000 positions, 50 v 50, N(0,1) rounded to 3 dp, +0.8 shift at 3 sites."""
    npos = 2000
    a = np.round(rng.normal(0, 1, (npos, 50)), 3)
    b = np.round(rng.normal(0, 1, (npos, 50)), 3)
    for s in (400, 401, 1500):
        b[s] = np.round(b[s] + 0.8, 3)
    sig0, off0 = _csr(list(a))
    sig1, off1 = _csr(list(b))
    bases = rng.choice(list('ACGT'), npos)
    return dict(sig0=sig0, off0=off0, sig1=sig1, off1=off1,
                chrom=np.array(['chr1'] * npos), strand=np.array(['+'] * npos),
                pos=np.arange(1000, 1000 + npos, dtype=np.int64), base0=bases, base1=bases)


def make_ragged(rng):
    """Ragged sizes 3..1000, two chromosomes x two strands, gaps in positions,
    some positions under MinCoverage in one group, one absent in group 2, one
    base mismatch (group-2 base is the one recorded, myDetect.py:436)."""
    recs = []
    for chrom, strand, start, count in (('chr2', '-', 50, 70), ('chr1', '+', 10, 90),
                                        ('chr1', '-', 5, 60), ('chr2', '+', 7, 80)):
        pos = start
        for _k in range(count):
            pos += 1 if rng.random() > 0.08 else int(rng.integers(2, 5))   # gaps break runs
            recs.append((chrom, strand, pos))
    npos = len(recs)
    ca, cb = [], []
    for i in range(npos):
        n0 = int(np.clip(np.round(rng.lognormal(np.log(60), 1.0)), 3, 1000))
        n1 = int(np.clip(np.round(rng.lognormal(np.log(40), 1.0)), 3, 1000))
        if i == 17:
            n1 = 0                                                  # absent in group 2
        shift = 0.9 if i % 37 == 5 else 0.0
        ca.append(np.round(rng.normal(0, 1, n0), 3))
        cb.append(np.round(rng.normal(shift, 1.2, n1), 3))
    sig0, off0 = _csr(ca)
    sig1, off1 = _csr(cb)
    base0 = rng.choice(list('ACGT'), npos)
    base1 = base0.copy()
    base1[33] = 'N' if base0[33] != 'N' else 'A'
    return dict(sig0=sig0, off0=off0, sig1=sig1, off1=off1,
                chrom=np.array([r[0] for r in recs]), strand=np.array([r[1] for r in recs]),
                pos=np.array([r[2] for r in recs], dtype=np.int64), base0=base0, base1=base1)


def make_ties(rng):
    """Tie-heavy data: low-variance signals on the 0.001 grid (int16 x1000
    representable), many within- and cross-group ties."""
    npos = 400
    ca, cb = [], []
    for i in range(npos):
        n0 = int(rng.integers(5, 260))
        n1 = int(rng.integers(5, 260))
        sd = (0.004, 0.02, 0.1, 1.0)[i % 4]
        ca.append(np.round(rng.normal(0.5, sd, n0), 3))
        cb.append(np.round(rng.normal(0.5 + (sd if i % 5 == 0 else 0), sd, n1), 3))
    sig0, off0 = _csr(ca)
    sig1, off1 = _csr(cb)
    bases = rng.choice(list('ACGT'), npos)
    return dict(sig0=sig0, off0=off0, sig1=sig1, off1=off1,
                chrom=np.array(['ecoli'] * npos), strand=np.array(['-'] * npos),
                pos=np.arange(3000, 3000 + npos, dtype=np.int64), base0=bases, base1=bases)


def make_sweep(rng):
    npos = 200
    a = rng.normal(0, 1, (npos, 30)).astype(np.float32)
    b = rng.normal(0, 1, (npos, 45)).astype(np.float32)
    b[60:63] += 1.0
    sig0, off0 = _csr(list(a))
    sig1, off1 = _csr(list(b))
    pos = np.arange(npos, dtype=np.int64)
    pos[120:] += 3                       # one gap -> two runs
    bases = rng.choice(list('ACGT'), npos)
    return dict(sig0=sig0, off0=off0, sig1=sig1, off1=off1,
                chrom=np.array(['chrS'] * npos), strand=np.array(['+'] * npos),
                pos=pos, base0=bases, base1=bases)


def make_track600(rng):
    """600 positions, 30 v 35 continuous reads, runs of 200, 50, 60, 70, 70 and 150 positions (gaps of 2-4 positions, a strand
    change, a chromosome change) — for windows up to neighborPvalues = 64 (129 positions wide: wider than most of the runs)"""
    recs = []
    for chrom, strand, start, count, gaps in (('chrT', '+', 100, 250, (200,)), ('chrT', '-', 40, 200, (60, 130)), ('chrU', '+', 7, 150, ())):
        pos = start
        for k in range(count):
            pos += int(rng.integers(2, 5)) if k in gaps else 1
            recs.append((chrom, strand, pos))
    npos = len(recs)
    a = rng.normal(0, 1, (npos, 30)).astype(np.float32)
    b = rng.normal(0, 1, (npos, 35)).astype(np.float32)
    for s in (70, 71, 72, 300, 520, 521):
        b[s] += 0.9
    sig0, off0 = _csr(list(a))
    sig1, off1 = _csr(list(b))
    bases = rng.choice(list('ACGT'), npos)
    return dict(sig0=sig0, off0=off0, sig1=sig1, off1=off1,
                chrom=np.array([r[0] for r in recs]), strand=np.array([r[1] for r in recs]),
                pos=np.array([r[2] for r in recs], dtype=np.int64), base0=bases, base1=bases)


def save_fixture(name, fx):
    np.savez_compressed(os.path.join(OUT, name + '_inputs.npz'), **fx)


def save_expected(name, exp, table):
    np.savez_compressed(os.path.join(OUT, name + '_expected.npz'), **exp)
    with open(os.path.join(OUT, name + '_sign_test.txt'), 'w') as f:
        f.write(table)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(1)
    crosscheck_restatements(np.random.default_rng(7))
    glue = load_reference_glue()
    mopt = {'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25}

    # ---- known-answer anchors through the reference's getKStest / get_combin_pvalue
    kat = {}
    a1 = [-1.2, -0.5, 0.0, 0.3, 0.3, 0.9, 1.4]
    b1 = [-0.1, 0.3, 0.8, 1.1, 1.5, 2.0]
    kat['KAT-1'] = {'a': a1, 'b': b1, 'out': glue['getKStest'](mopt, a1, b1, '+')}
    a2 = [(i - 25) / 10 for i in range(50)]
    b2 = [x + 0.75 for x in a2]
    kat['KAT-2'] = {'a': a2, 'b': b2, 'out': glue['getKStest'](mopt, a2, b2, '-')}
    pks = [0.5, 0.04, 1e-12, 3e-3, 0.7, 0.2, 0.9]
    for method in ('stouffer', 'fisher'):
        mo = {'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': method,
              'sign_test': [(('c', '+', 10 + i, 'A', 5, 5), [(0, 1), (0, 1), (0.1 * i, p)])
                            for i, p in enumerate(pks)]}
        kat['KAT-3-' + method] = {'ks_p': pks, 'nb': 2, 'dif': 2.0,
                                  'out': [glue['get_combin_pvalue'](mo, i) for i in range(len(pks))]}
    win = [0.9, 0.3, 1e-5, 0.02, 0.5]
    kat['KAT-4'] = {'window': win, 'weights': [25, 50, 100, 50, 25],
                    'stouffer': [float(v) for v in scipy.stats.combine_pvalues(win, method='stouffer', weights=[25, 50, 100, 50, 25])],
                    'fisher': [float(v) for v in scipy.stats.combine_pvalues(win)]}
    same = [0.25] * 6
    try:
        glue['getKStest'](mopt, same, same, '+')
        kat['EDGE-identical'] = {'raises': False}
    except ValueError as e:
        kat['EDGE-identical'] = {'raises': True, 'message': str(e),
                                 'ks': list(_ks_2samp(same, same)),
                                 'welch': [float(v) for v in scipy.stats.ttest_ind(same, same, equal_var=False)]}
    kat['constants'] = {'DBL_MIN': sys.float_info.min, 'DBL_MAX': sys.float_info.max,
                        'isf_DBL_MIN': float(scipy.stats.norm.isf(sys.float_info.min))}

    def enc(o):
        if isinstance(o, (np.floating, float)):
            f = float(o)
            return f if np.isfinite(f) else repr(f)
        if isinstance(o, (tuple, list)):
            return [enc(v) for v in o]
        if isinstance(o, dict):
            return {k: enc(v) for k, v in o.items()}
        return o
    with open(os.path.join(OUT, 'kat.json'), 'w') as f:
        json.dump(enc(kat), f, indent=1)

    # ---- table-level fixtures through the converted mfilter_coverage + mtest2 + save_test
    with tempfile.TemporaryDirectory() as tmp:
        myDetect = load_converted_reference(tmp)
        fx = make_g50(rng)
        save_fixture('g50', fx)
        for method in ('stouffer', 'fisher', 'ks'):
            exp, table = run_reference_table(myDetect, fx, 2, 2.0, method, 'g50_' + method)
            save_expected('g50_' + method, exp, table)
        # region ranking (RegionRankbyST=1) on the same inputs
        for tag, method, window, ovlp, pct, na, ru in (('w10_o0', 'stouffer', 10, 0, 0.1, '', 'pv'), ('w10_o1', 'stouffer', 10, 1, 0.1, '', 'pv'),
                                                       ('w3_o1_A', 'fisher', 3, 1, 0.3, 'A', 'pv'), ('w5_o0_ks', 'ks', 5, 0, 0.0, '', 'pv'),
                                                       ('w4_o1_st', 'stouffer', 4, 1, 0.5, '', 'st')):
            rr = run_reference_region_rank(myDetect, fx, method, window, ovlp, pct, na, ru)
            np.savez_compressed(os.path.join(OUT, 'g50_regionrank_%s.npz' % tag), method=method, window=window,
                                WindOvlp=ovlp, percentile=pct, NA=na, rankUse=ru, **rr)
        # FAST5 ingest: the reference's ReadAllFast5 over placeholder reads (stub h5py)
        reads = make_fast5_reads(np.random.default_rng(77))
        np.savez_compressed(os.path.join(OUT, 'fast5_reads.npz'),
                            group=np.array([r['group'] for r in reads]), rel=np.array([r['rel'] for r in reads]),
                            chrom=np.array([r['chrom'] for r in reads]), strand=np.array([r['strand'] for r in reads]),
                            start=np.array([r['start'] for r in reads]), has_align=np.array([r['has_align'] for r in reads]),
                            off=np.cumsum([0] + [len(r['norm_mean']) for r in reads]),
                            norm_mean=np.concatenate([r['norm_mean'] for r in reads]),
                            base=np.concatenate([r['base'] for r in reads]))
        ing = run_reference_ingest(myDetect, reads)
        for g in (0, 1):
            np.savez_compressed(os.path.join(OUT, 'fast5_expected_g%d.npz' % g), **ing[g])
        fx = make_ragged(rng)
        save_fixture('ragged', fx)
        for method in ('stouffer', 'fisher'):
            exp, table = run_reference_table(myDetect, fx, 2, 2.0, method, 'ragged_' + method)
            save_expected('ragged_' + method, exp, table)
        fx = make_ties(rng)
        save_fixture('ties', fx)
        exp, table = run_reference_table(myDetect, fx, 2, 2.0, 'stouffer', 'ties_stouffer')
        save_expected('ties_stouffer', exp, table)
        fx = make_sweep(rng)
        save_fixture('sweep', fx)
        for nb in (0, 1, 2, 3):
            for wdif in (1.0, 2.0, 3.0):
                for method in ('stouffer', 'fisher'):
                    if method == 'fisher' and wdif != 2.0:
                        continue
                    tag = 'sweep_nb%d_w%g_%s' % (nb, wdif, method)
                    exp, table = run_reference_table(myDetect, fx, nb, wdif, method, tag)
                    save_expected(tag, exp, table)
        # ---- round 5: corners no fixture pinned before (their own generator state: everything above reproduces unchanged)
        # --mstd (myDetect.py:425,437-438,541-544): the record dict and the `_meanstd.cvs` file (0-based positions), on
        # continuous rows (no mean sits on a decimal rounding boundary: the file is comparable byte for byte) and on the
        # 3-decimal ragged rows (means of n grid values do hit '%.3f' boundaries: compared numerically)
        for inp in ('sweep', 'ragged'):
            fx = dict(np.load(os.path.join(OUT, inp + '_inputs.npz')))
            exp, table, meanstd = run_reference_table(myDetect, fx, 2, 2.0, 'stouffer', inp + '_mstd', mstd=1)
            save_expected(inp + '_mstd', exp, table)
            with open(os.path.join(OUT, inp + '_mstd_meanstd.cvs'), 'w') as f:
                f.write(meanstd)
        # MinCoverage 3 and 20 (myDetect.py:301-314) on the ragged rows: positions enter / leave the tested set, runs re-form
        fx = dict(np.load(os.path.join(OUT, 'ragged_inputs.npz')))
        for mc in (3, 20):
            exp, table = run_reference_table(myDetect, fx, 2, 2.0, 'stouffer', 'ragged_mc%d' % mc, min_cov=mc)
            save_expected('ragged_mc%d' % mc, exp, table)
        # rankUse = 'st' (myDetect.py:447-462): the global order by statistics, reversed
        fx = dict(np.load(os.path.join(OUT, 'g50_inputs.npz')))
        for method in ('stouffer', 'ks'):
            exp, table = run_reference_table(myDetect, fx, 2, 2.0, method, 'g50_rankst_' + method, rank_use='st')
            save_expected('g50_rankst_' + method, exp, table)
        # neighborPvalues 5, 16 and 64 (the largest the ABI takes) on a 600-position track of several runs
        fx = make_track600(np.random.default_rng(505))
        save_fixture('track600', fx)
        for nb in (5, 16, 64):
            for method in ('stouffer', 'fisher'):
                tag = 'track600_nb%d_%s' % (nb, method)
                exp, table = run_reference_table(myDetect, fx, nb, 2.0, method, tag)
                save_expected(tag, exp, table)
    print('golden fixtures written to', os.path.abspath(OUT))


if __name__ == '__main__':
    main()
