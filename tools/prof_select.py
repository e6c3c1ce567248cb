"""cProfile of cli.select_positions on a synthetic container: python tools/prof_select.py [positions] [reads]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from nanomod_amd import cli
P = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(1)
def grp(shift):
    return dict(chrom=np.full(P, 'chr'), strand=np.full(P, '+'), pos=np.arange(P, dtype=np.int64), base=np.full(P, 'A'),
                off=np.arange(0, (P + 1) * n, n, dtype=np.int64), sig=np.round(rng.normal(shift, 1, P * n), 3))
g0, g1 = grp(0.0), grp(0.1)
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
cli.select_positions(g0, g1, 5, 3)
pr.disable(); print('select_positions %.2f s' % (time.perf_counter() - t0))
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
