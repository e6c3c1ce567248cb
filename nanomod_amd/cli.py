"""`detect`-compatible command line: the reference's `NanoMod.py detect` flags
(NanoMod.py:344-393) over neutral `.npz` containers, array-native end to end: coverage filter and
position intersection (myDetect.py:301-314,421-431) with numpy, the tests + combine on the GPU through
the C ABI, `_sign_test.txt` through nmod_write_sign_test, ranking as myDetect.py:447-462.

    python -m nanomod_amd.cli detect --wrkBase1 groupA.npz --wrkBase2 groupB.npz --FileID run1 --outFolder out/
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

from . import _lib as L
from . import container, detect, engine


def build_parser():
    p = argparse.ArgumentParser(prog='nanomod_amd', description='MI355X implementation of NanoMod detect (hot path only)')
    sub = p.add_subparsers(dest='cmd')
    d = sub.add_parser('detect', help='per-base KS / MWU / Welch-t tests + window combine')
    d.add_argument('--outLevel', type=int, default=2, choices=[0, 1, 2, 3])           # NanoMod.py:348
    d.add_argument('--wrkBase1', required=True, help='read group 1: a .npz container, or a FAST5 folder (needs h5py)')
    d.add_argument('--wrkBase2', required=True, help='read group 2: a .npz container, or a FAST5 folder (needs h5py)')
    d.add_argument('--min_lr', type=int, default=500)                                  # NanoMod.py:387
    d.add_argument('--min_lr_nb', type=int, default=0)
    d.add_argument('--FileID', default='mod')                                          # NanoMod.py:349
    d.add_argument('--outFolder', default='mRes')                                      # NanoMod.py:350
    d.add_argument('--MinCoverage', type=int, default=5)                               # NanoMod.py:354
    d.add_argument('--topN', type=int, default=30)                                     # NanoMod.py:355
    d.add_argument('--neighborPvalues', type=int, default=2)                           # NanoMod.py:357
    d.add_argument('--WeightsDif', type=float, default=2.0)                            # NanoMod.py:358
    d.add_argument('--testMethod', default='stouffer', choices=['fisher', 'stouffer', 'ks'])   # NanoMod.py:359
    d.add_argument('--rankUse', default='pv', choices=['st', 'pv'])                    # NanoMod.py:361
    d.add_argument('--SaveTest', type=int, default=1, choices=[0, 1])                  # NanoMod.py:362
    d.add_argument('--mstd', type=int, default=0)                                      # NanoMod.py:378
    d.add_argument('--window', type=int, default=21)                                   # NanoMod.py:351
    d.add_argument('--RegionRankbyST', type=int, default=0, choices=[0, 1])            # NanoMod.py:363
    d.add_argument('--percentile', type=float, default=0.1)                            # NanoMod.py:364
    d.add_argument('--WindOvlp', type=int, default=0, choices=[0, 1])                  # NanoMod.py:365
    d.add_argument('--NA', type=str, default='', choices=['', 'A', 'C', 'G', 'T'])     # NanoMod.py:366
    d.add_argument('--Pos', default='', help="region of interest chr:pos[:pos2] (1-based)")        # NanoMod.py:377
    d.add_argument('--plotType', default='Density', choices=['Violin', 'Density'],
                   help='accepted for compatibility: the R plots are outside this build')  # NanoMod.py:385
    d.add_argument('--downsampling_quantile', type=float, default=0.25)                # NanoMod.py:389
    d.add_argument('--downsampling', type=int, default=100)                            # NanoMod.py:390
    d.add_argument('--coverages', type=str, default='0-0')                             # NanoMod.py:392
    d.add_argument('--seed', type=int, default=0, help='seed of the down-sampling draws (the reference is unseeded)')
    d.add_argument('--device', type=int, default=0)
    d.add_argument('--fast5Reader', default='', help="module:function used to read one resquiggled read file, path -> "
                   "(mapped_chrom, mapped_start, mapped_strand, norm_mean[], base[]) | None; default: the h5py reader of "
                   "nanomod_amd.fast5_ingest (Events table + Alignment attributes, myFast5.py:92-126)")
    return p


def validate(a):
    """NanoMod.py:40-97 (mCommonParam): the checks that concern this path."""
    errs = []
    if a.MinCoverage < 3:
        errs.append('Error: --MinCoverage should be not less than 3')                 # NanoMod.py:65-67
    if a.topN < 1:
        errs.append('Error: --topN should be larger than 0')
    if a.neighborPvalues < 0:
        errs.append('Error: --neighborPvalues should not be negative')
    if a.neighborPvalues > L.MAX_NB:
        errs.append('Error: --neighborPvalues larger than %d is not supported' % L.MAX_NB)
    if a.WeightsDif < 1.0:                                                             # NanoMod.py:76-78: floor at 1.0
        a.WeightsDif = 1.0
    if (a.window - 1) // 2 < 1:                                                        # NanoMod.py:51-53
        errs.append('Window size (%d) is too small' % a.window)
    a.percentile = 0.0 if a.percentile < 0 else (0.99 if a.percentile >= 1 else a.percentile)   # NanoMod.py:91-92
    a.roi = {}                                                                         # NanoMod.py:117-129
    if a.Pos != '':
        mpos = a.Pos.split(':')
        a.roi['Chr'] = mpos[0]
        if len(mpos) > 1:
            a.roi['Pos'] = int(mpos[1]) - 1
            if a.roi['Pos'] < 0:
                errs.append('The position (%d) of interest should not be less than 0' % a.roi['Pos'])
        if len(mpos) > 2:
            a.roi['Pos2'] = int(mpos[2]) - 1
            if a.roi['Pos2'] < 0:
                errs.append('The position (%d) of interest should not be less than 0' % a.roi['Pos2'])
            if a.roi['Pos2'] - a.roi['Pos'] < 1:
                errs.append('The end position (%d) is not larger than the start position (%d)' % (a.roi['Pos2'], a.roi['Pos']))
        if 'Pos' in a.roi and 'Pos2' not in a.roi:                                     # myDetect.py:550-558
            neighbors = (a.window - 1) // 2
            a.roi['start_pos'] = max(a.roi['Pos'] - neighbors, 0)
            a.roi['end_pos'] = a.roi['Pos'] + neighbors
    for f in (a.wrkBase1, a.wrkBase2):
        if not (os.path.isfile(f) or os.path.isdir(f)):
            errs.append('Error: input %s does not exist' % f)
    return errs


def _chrom_codes(*chrom_arrays):
    """sorted chromosome names over all inputs and, per input, the index of every entry's name.  Chromosome columns are
    long runs of one name: only the first entry of every run is looked up (np.unique over 9 M strings was half of
    select_positions)."""
    runs = []
    for c in chrom_arrays:
        c = np.asarray(c)
        if len(c) == 0:
            runs.append((c[:0], np.zeros(0, np.int64)))
            continue
        heads = np.flatnonzero(np.r_[True, c[1:] != c[:-1]])
        runs.append((c[heads], np.diff(np.r_[heads, len(c)])))
    heads_all = np.concatenate([r[0] for r in runs]) if runs else np.zeros(0, dtype=str)
    names = np.unique(heads_all)
    out = [np.repeat(np.searchsorted(names, vals), lens).astype(np.int64) for vals, lens in runs]
    return [str(n) for n in names.tolist()], out


def _keys(g, cid):
    sid = (g['strand'] == '-').astype(np.int64)                                       # '+' sorts before '-'
    return (cid << 41) | (sid << 40) | g['pos'].astype(np.int64)


def select_positions(g0, g1, min_coverage, out_level=detect.OUTPUT_ERROR, log=print):
    """Coverage filter + intersection + ordering: the tested-position set of mtest2 as CSR arrays."""
    # mfilter_coverage (myDetect.py:301-314): per group
    keep0 = np.nonzero(np.diff(g0['off']) >= min_coverage)[0]
    keep1 = np.nonzero(np.diff(g1['off']) >= min_coverage)[0]
    names, (cid0, cid1) = _chrom_codes(g0['chrom'], g1['chrom'])
    k0, k1 = _keys(g0, cid0)[keep0], _keys(g1, cid1)[keep1]
    # positions present in both groups, in sorted (chrom, strand, pos) order (myDetect.py:421,427-431)
    if len(k0) == len(k1) and (len(k0) < 2 or bool(np.all(k0[1:] > k0[:-1]))) and np.array_equal(k0, k1):
        common, rows0, rows1 = k0, keep0, keep1          # the same sorted positions in both groups: nothing to intersect
    else:
        common, i0, i1 = np.intersect1d(k0, k1, assume_unique=True, return_indices=True)
        rows0, rows1 = keep0[i0], keep1[i1]
    sig0, off0 = container.gather_rows(g0['sig'], g0['off'], rows0)
    sig1, off1 = container.gather_rows(g1['sig'], g1['off'], rows1)
    npos = len(common)
    chrom = g1['chrom'][rows1]; strand = g1['strand'][rows1]; pos = g1['pos'][rows1]; base = g1['base'][rows1]
    mism = np.nonzero(g0['base'][rows0] != base)[0]
    if out_level <= detect.OUTPUT_ERROR:
        for i in mism[:20]:
            log('Error not equal', (chrom[i], strand[i]), int(pos[i]), base[i], g0['base'][rows0][i])
    if npos:
        sig0, sig1 = detect.encode_pair(np.asarray(sig0), np.asarray(sig1))
    else:
        sig0 = sig1 = np.zeros(0, np.float32)
    rid = detect.run_ids(chrom, strand, pos)
    n0 = np.diff(off0).astype(np.int32); n1 = np.diff(off1).astype(np.int32)
    meta = dict(chrom=chrom, strand=strand, pos=pos, base=base, n0=n0, n1=n1, names=names,
                chrom_id=cid1[rows1].astype(np.int32))
    return meta, sig0, off0, sig1, off1, rid


def load_input(path, a, log=print):
    """A `.npz` container, or a folder of resquiggled FAST5 files read like ReadAllFast5 (myDetect.py:547-633)."""
    if os.path.isdir(path):
        from . import fast5_ingest
        opts = {'min_lr': a.min_lr, 'min_lr_nb': a.min_lr_nb}
        opts.update(getattr(a, 'roi', {}))                                              # read- and event-level filters
        reader = None
        if getattr(a, 'fast5Reader', ''):
            import importlib
            mod, _, fn = a.fast5Reader.partition(':')
            reader = getattr(importlib.import_module(mod), fn)
        return fast5_ingest.ingest_folder(path, opts, reader=reader, log=log)
    g = container.load_group(path)
    roi = getattr(a, 'roi', {})
    if roi:
        # a container holds aggregated positions: only the position-level part of the region filter applies
        # (myDetect.py:72,112-114); the read-level parts need the reads (FAST5 folders)
        keep = g['chrom'] == roi['Chr']
        if 'start_pos' in roi:
            keep &= (g['pos'] >= roi['start_pos']) & (g['pos'] <= roi['end_pos'])
        rows = np.flatnonzero(keep)
        sig, off = container.gather_rows(g['sig'], g['off'], rows)
        g = dict(chrom=g['chrom'][rows], strand=g['strand'][rows], pos=g['pos'][rows], base=g['base'][rows], off=off, sig=sig)
    return g


def run_detect(a, log=print):
    engine.warm_up(a.device)                                    # HIP start-up beside the loading of the inputs
    g0, g1 = load_input(a.wrkBase1, a, log), load_input(a.wrkBase2, a, log)
    t0 = time.time()
    meta, sig0, off0, sig1, off1, rid = select_positions(g0, g1, a.MinCoverage, a.outLevel, log)
    npos = len(rid)
    chrom, strand, pos, base = meta['chrom'], meta['strand'], meta['pos'], meta['base']
    method, nb = a.testMethod, a.neighborPvalues
    dev_method = method if (method in ('stouffer', 'fisher') and nb > 0) else 'ks'
    res = engine.detect_host(sig0, off0, sig1, off1, rid, nb=nb, weights_dif=a.WeightsDif, method=dev_method,
                             want_mstd=a.mstd != 0, device=a.device)
    if npos and np.any(res['status'] & L.STATUS_MWU_ALL_IDENTICAL):
        raise ValueError('All numbers are identical in mannwhitneyu')                  # scipy 1.2.1, uncaught in the reference
    cov = [int(x) for x in a.coverages.split('-')]                                     # NanoMod.py:174-176
    if npos:
        detect.downsample_update(res, sig0, off0, sig1, off1, rid, strand, cov * 2 if len(cov) == 1 else cov,
                                 iters=a.downsampling, quantile=a.downsampling_quantile, seed=a.seed, nb=nb,
                                 weights_dif=a.WeightsDif, method=dev_method, device=a.device)
    if a.outLevel <= detect.OUTPUT_INFO:
        log('Producing pvalues: consuming time %d' % (time.time() - t0))
    if method != 'ks' and nb == 0:                                                     # myDetect.py:413
        res['comb_st'], res['comb_p'] = res['ks_d'].copy(), res['ks_p'].copy()
    os.makedirs(a.outFolder, exist_ok=True)
    if a.SaveTest:
        with_comb = nb > 0 and method != 'ks'                                          # myDetect.py:533
        path = os.path.join(a.outFolder, a.FileID + '_sign_test.txt')
        write_sign_test(path, meta, res, with_comb)
        if a.outLevel <= detect.OUTPUT_ERROR:
            log('Test data is saved in', path)
        if a.mstd != 0:
            with open(os.path.join(a.outFolder, a.FileID + '_meanstd.cvs'), 'w') as mw:   # myDetect.py:541-544
                for i in range(npos):
                    mw.write('%s %s %d %s %.3f %.3f %.3f %.3f\n' % (chrom[i], strand[i], pos[i], base[i], res['mean0'][i],
                                                                    res['std0'][i], res['mean1'][i], res['std1'][i]))
    if a.RegionRankbyST == 0:
        order = rank_order(res, method, a.rankUse, a.device)
    else:                                                                              # myDetect.py:463-515
        recs = []
        for i in range(npos):
            t = [(res['mwu_u'][i], res['mwu_p'][i]), (res['t_t'][i], res['t_p'][i]), (res['ks_d'][i], res['ks_p'][i])]
            if method != 'ks':
                t.append((res['comb_st'][i], res['comb_p'][i]))
            recs.append(((str(chrom[i]), str(strand[i]), int(pos[i]), str(base[i]), int(meta['n0'][i]), int(meta['n1'][i]), i), t))
        mo = {'sign_test': recs, 'window': (a.window - 1) // 2, 'WindOvlp': a.WindOvlp, 'percentile': a.percentile, 'NA': a.NA}
        ranked = detect.region_rank(mo, 2 if method == 'ks' else 3, 1 if a.rankUse == 'pv' else 0)
        order = np.array([r[0][6] for r in ranked], dtype=np.int64)
    return meta, res, order


def rank_order(res, method, rank_use, device=0):
    """myDetect.py:447-462: stable ascending sort by (combined, KS, MWU) p-value (or statistic, reversed)."""
    pind = 'p' if rank_use == 'pv' else 'st'
    first = ('comb_p' if pind == 'p' else 'comb_st') if method != 'ks' else ('ks_p' if pind == 'p' else 'ks_d')
    return engine.rank_order_host(res[first], res['ks_p' if pind == 'p' else 'ks_d'], res['mwu_p' if pind == 'p' else 'mwu_u'],
                                  descending=(rank_use == 'st'), device=device)


def write_sign_test(path, meta, res, with_comb):
    engine.write_sign_test_host(path, meta, res, with_comb)


def main(argv=None):
    parser = build_parser()
    a = parser.parse_args(argv)
    if a.cmd != 'detect':
        parser.print_help()
        return 1
    errs = validate(a)
    if errs:
        print('\n'.join(errs))
        parser.parse_args(['detect', '-h']) if False else None
        return 1
    meta, res, order = run_detect(a)
    top = order[:a.topN]
    first = ('comb_p' if a.testMethod != 'ks' else 'ks_p')
    for r, i in enumerate(top):
        print('%d %s %s %d %s %.3E' % (r + 1, meta['chrom'][i], meta['strand'][i], meta['pos'][i] + 1, meta['base'][i],
                                       res[first][i]))
    return 0


if __name__ == '__main__':
    sys.exit(main())
