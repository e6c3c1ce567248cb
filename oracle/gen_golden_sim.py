#!/usr/bin/env python3
"""Golden vectors for the simulation repeat loops: the reference's own
getGenomeEvents (mySimulate.py:124-139) + mfilter_coverage + mtest2 (myDetect.py) + getTopRank
(mySimulate.py:287-328), run in THIS container on synthetic read pools with explicit read selections
(the random draws of mySimulat2.py:136-141 / myDownSampling0.py:66-78 are inputs here, so the run is
deterministic).  TEST INFRASTRUCTURE: only fixtures (inputs + expected numbers) are committed to tests/golden/;
the reference is read through a throw-away lib2to3 conversion in a temp directory, exactly like gen_golden.py.

    python oracle/gen_golden_sim.py     # needs /root/reference
"""
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G                      # noqa: E402  (REF, the restated scipy-1.2.1 functions, the converter)

OUT = os.path.join(HERE, '..', 'tests', 'golden')


def load_converted_simulate(tmpdir):
    from lib2to3.refactor import RefactoringTool, get_fixers_from_package
    myDetect = G.load_converted_reference(tmpdir)
    rt = RefactoringTool(get_fixers_from_package('lib2to3.fixes'))
    with open(os.path.join(G.REF, 'mySimulate.py')) as f:
        src = f.read()
    src = src.expandtabs(8)                      # (the file mixes tabs and spaces; Python 2 reads a tab as the next multiple of 8)
    if not src.endswith('\n'):
        src += '\n'
    with open(os.path.join(tmpdir, 'mySimulate.py'), 'w') as f:
        f.write(str(rt.refactor_string(src, 'mySimulate.py')))
    import mySimulate
    return myDetect, mySimulate


class Py2Str(str):
    """a base letter that orders above every integer, as any str does in Python 2 (getTopRank's `rank_of_interest[-1]>-1`
    relies on it: mySimulate.py:327)"""
    def __gt__(self, other):
        return True if isinstance(other, int) else str.__gt__(self, other)


class Events(object):
    """what ReadNanoraw_events returns as far as getGenomeEvents uses it: len() and ['norm_mean'][i] / ['base'][i]"""
    def __init__(self, norm_mean, base):
        self.cols = {'norm_mean': norm_mean, 'base': base}

    def __len__(self):
        return len(self.cols['norm_mean'])

    def __getitem__(self, k):
        return self.cols[k]


def make_pools(rng, n_case=260, n_control=420, genome=(('spel', '-', 2900, 3300), ('spel', '+', 2900, 3300), ('chrB', '+', 100, 400))):
    """Read pools in the structure readEvents builds (mySimulate.py:101-122): key -> (events, (chrom, start, strand)).
    Case reads carry a +0.9 shift at the reference's hard-coded site spel:-:3072 and its two neighbours."""
    pools = {}
    for label, n, shifted in (('case', n_case, True), ('control', n_control, False)):
        reads = {}
        for r in range(n):
            chrom, strand, lo, hi = genome[rng.integers(0, len(genome) if r % 5 == 0 else 1)]
            length = int(rng.integers(60, 300))
            start = int(rng.integers(lo, hi - 40))
            length = min(length, hi - start)
            nm = np.round(rng.normal(0, 1, length), 3)
            base = []
            for i in range(length):
                pos = start + i if strand == '+' else start + length - 1 - i
                base.append(Py2Str('ACGT'[pos % 4]))
                if shifted and chrom == 'spel' and strand == '-' and abs(pos - 3072) <= 1:
                    nm[i] = np.round(nm[i] + 0.9, 3)
            reads['%s_read_%04d.fast5' % (label, r)] = (Events([np.float64(v) for v in nm], base), (chrom, start, strand))
        pools[label] = reads
    return pools


def run_reference_repeat(myDetect, mySimulate, pools, sel, labels, opts):
    """one trip of the loops at mySimulat2.py:135-167 / myDownSampling0.py:62-117 with the random draws given"""
    from collections import defaultdict
    mo = dict(opts)
    mo['ds2'] = ['simulate_case', 'folder_control']
    mo['sign_test'] = []; mo['sorted_sign_test'] = []
    for ds in mo['ds2']:
        mo[ds] = {'base': defaultdict(lambda: defaultdict(str)), 'norm_mean': defaultdict(lambda: defaultdict(list))}
    dicts = []
    for pool_name, idx in sel:
        keys = sorted(pools[pool_name].keys())
        dicts.append({keys[i]: pools[pool_name][keys[i]] for i in idx})
    mySimulate.getGenomeEvents(dicts, labels, mo)
    myDetect.mfilter_coverage(mo)
    myDetect.mtest2(mo)
    rank = mySimulate.getTopRank(mo)
    st = mo['sign_test']
    return rank, dict(chrom=np.array([r[0][0] for r in st]), strand=np.array([r[0][1] for r in st]),
                      pos=np.array([r[0][2] for r in st], dtype=np.int64), n0=np.array([r[0][4] for r in st]),
                      n1=np.array([r[0][5] for r in st]), comb_p=np.array([r[1][3][1] for r in st]),
                      ks_p=np.array([r[1][2][1] for r in st]), mwu_u=np.array([r[1][0][0] for r in st]))


def main():
    rng = np.random.default_rng(20240917)
    pools = make_pools(rng)
    opts = {'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
            'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 0,
            'RegionRankbyST': 0, 'MinCoverage': 5, 'window': 2, 'topN': 30, 'outFolder': tempfile.gettempdir(), 'FileID': 'sim'}
    # flat arrays of the pools (what the build's loader holds): reads sorted by key
    flat = {}
    for name in ('case', 'control'):
        keys = sorted(pools[name].keys())
        flat[name + '_chrom'] = np.array([pools[name][k][1][0] for k in keys])
        flat[name + '_strand'] = np.array([pools[name][k][1][2] for k in keys])
        flat[name + '_start'] = np.array([pools[name][k][1][1] for k in keys], dtype=np.int64)
        flat[name + '_off'] = np.cumsum([0] + [len(pools[name][k][0]) for k in keys]).astype(np.int64)
        flat[name + '_norm_mean'] = np.concatenate([np.asarray(pools[name][k][0]['norm_mean'], dtype=np.float64) for k in keys])
        flat[name + '_base'] = np.array([str(b) for k in keys for b in pools[name][k][0]['base']], dtype='U1')
    repeats = []
    with tempfile.TemporaryDirectory() as tmp:
        myDetect, mySimulate = load_converted_simulate(tmp)
        ncase, ncon = len(pools['case']), len(pools['control'])
        # (a) mySimulat2.py: CaseSize modified reads + unmodified ones at `Percentage`, against a control sample
        for rt, (case_size, perc) in enumerate(((60, 0.5), (90, 0.6), (40, 0.4), (120, 0.75))):
            n_un1 = int(case_size * (1 - perc) / perc); n_un2 = int(case_size / perc)
            c = rng.choice(ncase, case_size, replace=False)
            k = rng.choice(ncon, n_un1 + n_un2, replace=False)
            sel = [('case', c), ('control', k[:n_un1]), ('control', k[n_un1:])]
            labels = ['simulate_case', 'simulate_case', 'folder_control']
            rank, tab = run_reference_repeat(myDetect, mySimulate, pools, sel, labels, opts)
            repeats.append(('simulat2_%d' % rt, sel, labels, rank, tab))
        # (b) myDownSampling0.py: both groups down-sampled to CaseSize reads
        for rt, case_size in enumerate((80, 150, 220)):
            c = rng.choice(ncase, case_size, replace=False)
            k = rng.choice(ncon, case_size, replace=False)
            sel = [('case', c), ('control', k)]
            labels = ['simulate_case', 'folder_control']
            rank, tab = run_reference_repeat(myDetect, mySimulate, pools, sel, labels, opts)
            repeats.append(('downsampling_%d' % rt, sel, labels, rank, tab))
        # (c) RegionRankbyST = 0 with a larger window and the KS method (closesize / window completeness branches)
        o2 = dict(opts); o2['testMethod'] = 'fisher'; o2['window'] = 5
        c = rng.choice(ncase, 100, replace=False); k = rng.choice(ncon, 100, replace=False)
        sel = [('case', c), ('control', k)]
        rank, tab = run_reference_repeat(myDetect, mySimulate, pools, sel, ['simulate_case', 'folder_control'], o2)
        repeats.append(('fisher_w5', sel, ['simulate_case', 'folder_control'], rank, tab))
    out = dict(flat)
    out['names'] = np.array([r[0] for r in repeats])
    for name, sel, labels, rank, tab in repeats:
        out[name + '_rank'] = np.int64(rank)
        out[name + '_nsel'] = np.int64(len(sel))
        for j, (pool_name, idx) in enumerate(sel):
            out['%s_sel%d_pool' % (name, j)] = np.array(pool_name)
            out['%s_sel%d_idx' % (name, j)] = np.asarray(idx, dtype=np.int64)
            out['%s_sel%d_label' % (name, j)] = np.array(labels[j])
        for k2, v in tab.items():
            out['%s_%s' % (name, k2)] = v
    np.savez_compressed(os.path.join(OUT, 'simulate_repeats.npz'), **out)
    print('ranks:', [(r[0], r[3]) for r in repeats])
    print('written', os.path.abspath(os.path.join(OUT, 'simulate_repeats.npz')))


if __name__ == '__main__':
    main()
