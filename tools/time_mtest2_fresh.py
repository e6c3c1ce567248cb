"""mtest2 in a fresh process (HIP start-up included): python tools/time_mtest2_fresh.py [positions] [reads]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
t_imp = time.perf_counter()
import nanomod_amd as nm
t_imp = time.perf_counter() - t_imp
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(1)
mo = {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
      'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
      'outFolder': tempfile.mkdtemp(), 'FileID': 'fresh', 'MinCoverage': 5}
for ds, sh in (('A', 0.0), ('B', 0.1)):
    mo[ds] = {'nmod_container': dict(chrom=np.full(P, 'chr'), strand=np.full(P, '+'), pos=np.arange(P, dtype=np.int64),
                                     base=np.full(P, 'A'), off=np.arange(0, (P + 1) * n, n, dtype=np.int64),
                                     sig=np.round(rng.normal(sh, 1, P * n), 3))}
t0 = time.perf_counter()
nm.mtest2(mo)
dt = time.perf_counter() - t0
print('import nanomod_amd %.2f s; first mtest2 of the process: %.2f s = %.3g positions/s (%d positions x %d v %d, container input)'
      % (t_imp, dt, P / dt, P, n, n))
