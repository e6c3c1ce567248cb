"""End-to-end cost of the drop-in mtest2 (host glue + PCIe staging + kernels + table + ranking) for the three input
shapes nanomod_amd.detect.build_csr accepts:

  lists      the reference's own structure: dict -> dict -> Python list of numpy.float64 (myDetect.py:124,569-572)
  arrays     the same dicts with one numpy array per position
  container  flat CSR arrays attached as moptions[ds]['nmod_container'] (what an array-native loader hands over)

usage: python tools/profile_mtest2.py [positions=4600000] [reads_per_group=20] [shapes=container,arrays,lists]
(the `lists` shape holds positions x reads x 2 Python float objects: 4.6 M x 20 is ~8 GB of host memory)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import nanomod_amd as nm                       # noqa: E402
from nanomod_amd import container              # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shapes = (sys.argv[3] if len(sys.argv) > 3 else 'container,arrays,lists').split(',')
rng = np.random.default_rng(1)
vals = {ds: np.round(rng.normal(shift, 1, (P, n)), 3) for ds, shift in (('A', 0.0), ('B', 0.1))}     # 3-dp Events, float64


def base_options(out):
    return {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
            'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
            'outFolder': out, 'FileID': 'prof', 'MinCoverage': 5}


def timed(label, fn):
    t0 = time.perf_counter()
    r = fn()
    print('  %-34s %8.2f s' % (label, time.perf_counter() - t0), flush=True)
    return r


print('mtest2 end to end: %d positions x %d v %d reads (float64 on the 0.001 grid), stouffer window 5, SaveTest 1' % (P, n, n))
for shape in shapes:
    out = tempfile.mkdtemp()
    mo = base_options(out)
    t_build = time.perf_counter()
    if shape == 'container':
        for ds in ('A', 'B'):
            mo[ds] = {'nmod_container': dict(chrom=np.full(P, 'chr'), strand=np.full(P, '+'), pos=np.arange(P, dtype=np.int64),
                                             base=np.full(P, 'A'), off=np.arange(0, (P + 1) * n, n, dtype=np.int64),
                                             sig=vals[ds].reshape(-1))}
    else:
        for ds in ('A', 'B'):
            v = vals[ds]
            rows = {i: v[i] for i in range(P)} if shape == 'arrays' else {i: [np.float64(x) for x in v[i]] for i in range(P)}
            mo[ds] = {'norm_mean': {('chr', '+'): rows}, 'base': {('chr', '+'): {i: 'A' for i in range(P)}}, 'basedict': {}}
    print('[%s] input structure built in %.1f s (not part of mtest2)' % (shape, time.perf_counter() - t_build), flush=True)
    if shape != 'container':
        timed('mfilter_coverage', lambda: nm.mfilter_coverage(mo))
    # the stages of mtest2, timed one by one through the same functions it calls
    from nanomod_amd import detect as D, engine as E
    meta, sig0, off0, sig1, off1, rid = timed('build_csr (order, CSR, dtype)', lambda: D.build_csr(mo))
    res = timed('detect_host (PCIe + kernels)', lambda: E.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method='stouffer'))
    timed('rank_order (device radix sort)', lambda: E.rank_order_host(res['comb_p'], res['ks_p'], res['mwu_p']))
    timed('write _sign_test.txt (C writer)', lambda: E.write_sign_test_host(os.path.join(out, 'x.txt'), meta, res, True))
    t0 = time.perf_counter()
    nm.mtest2(mo)
    dt = time.perf_counter() - t0
    print('  mtest2 total                       %8.2f s  = %.3g positions/s; first record %r' % (dt, P / dt, mo['sorted_sign_test'][0][0]))
    t0 = time.perf_counter()
    k = sum(1 for _ in zip(range(100000), mo['sign_test']))
    print('  materialising %d records lazily  %8.2f s' % (k, time.perf_counter() - t0), flush=True)
    del mo
