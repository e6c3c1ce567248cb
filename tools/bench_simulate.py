"""Throughput of the simulation repeat loops on one GPU: synthetic read pools of the shape the
reference's studies use (reads of a few hundred events over a ~5 kb amplicon, a planted shift at the target site),
mySimulat2-style repeats.  python tools/bench_simulate.py [case_reads] [control_reads] [case_size] [repeats]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch                                    # noqa: E402
from nanomod_amd import simulate                # noqa: E402

n_case = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
n_con = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
case_size = int(sys.argv[3]) if len(sys.argv) > 3 else 400
repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 50
rng = np.random.default_rng(7)


def pool(n, shifted):
    start = rng.integers(1000, 5000, n); lens = rng.integers(200, 900, n)
    off = np.zeros(n + 1, np.int64); off[1:] = np.cumsum(lens)
    nm = np.round(rng.normal(0, 1, off[-1]), 3)
    strand = np.where(rng.random(n) < 0.5, '-', '+')
    if shifted:
        for r in range(n):
            if strand[r] == '-':
                i = start[r] + lens[r] - 1 - simulate.TARGET_POS
                if 0 <= i < lens[r]:
                    nm[off[r] + i] += 0.9
    base = np.frombuffer(rng.integers(0, 4, off[-1]).astype(np.uint8).tobytes(), dtype=np.uint8)
    base = np.array(list('ACGT'))[base]
    return simulate.ReadPool(np.full(n, simulate.TARGET_CHR), strand, start, off, nm, base, device='cuda:0')


case, control = pool(n_case, True), pool(n_con, False)
opts = {'MinCoverage': 5, 'neighborPvalues': 2, 'WeightsDif': 2.0, 'testMethod': 'stouffer', 'rankUse': 'pv', 'window': 2, 'RegionRankbyST': 0}
simulate.simulate_case_size(case, control, case_size, 0.5, 3, opts, seed=1)          # warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
ranks = simulate.simulate_case_size(case, control, case_size, 0.5, repeats, opts, seed=2)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print('pools %d / %d reads, %d modified reads at 50 %% against %d control reads per repeat: %d repeats in %.2f s = %.1f repeats/s; '
      'median rank of the planted site %d' % (n_case, n_con, case_size, int(case_size / 0.5), repeats, dt, repeats / dt, int(np.median(ranks))))
pr = cProfile.Profile(); pr.enable()
simulate.simulate_case_size(case, control, case_size, 0.5, 10, opts, seed=3)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
