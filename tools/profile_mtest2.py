"""Where the drop-in mtest2 spends its time on a reference-shaped moptions dict (host glue vs device)."""
import cProfile, os, pstats, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import nanomod_amd as nm
P, n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 30
rng = np.random.default_rng(1)
mo = {'ds2': ['A', 'B'], 'outLevel': 3, 'mstd': 0, 'coverages': [0, 0], 'downsampling': 100, 'downsampling_quantile': 0.25,
      'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'SaveTest': 1, 'RegionRankbyST': 0,
      'outFolder': tempfile.mkdtemp(), 'FileID': 'prof', 'MinCoverage': 5}
t0 = time.time()
for ds, shift in (('A', 0.0), ('B', 0.1)):
    vals = np.round(rng.normal(shift, 1, (P, n)), 3)
    mo[ds] = {'norm_mean': {('chr', '+'): {i: [np.float64(v) for v in vals[i]] for i in range(P)}},
              'base': {('chr', '+'): {i: 'A' for i in range(P)}}, 'basedict': {}}
print('built reference-shaped dicts in %.1f s' % (time.time() - t0), flush=True)
nm.mfilter_coverage(mo)
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
nm.mtest2(mo)
dt = time.time() - t0; pr.disable()
print('mtest2: %.2f s for %d positions (%.0f positions/s)' % (dt, P, P / dt))
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
