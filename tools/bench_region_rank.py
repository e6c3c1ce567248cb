"""nmod_region_rank and nmod_rank_order at genome scale (4.6 M records, window 21 + 1, overlapping windows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import nanomod_amd as nm
from nanomod_amd import engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
rng = np.random.default_rng(1)
half = n // 2
lo = np.r_[np.zeros(half, np.int32), np.full(n - half, half, np.int32)]
hi = np.r_[np.full(half, half - 1, np.int32), np.full(n - half, n - 1, np.int32)]
pos = np.r_[np.arange(half), np.arange(n - half)].astype(np.int64)
base = rng.choice(list(b'ACGT'), n).astype(np.uint8).tobytes()
p = rng.random(n) ** 3
engine.rank_order_host(p[:1000], p[:1000], p[:1000])          # load + warm up
for ovlp in (1, 0):
    t0 = time.time()
    idx = engine.region_rank_host(lo, hi, pos, base, p, 22, 1 if ovlp else 22, '', 0.1, ovlp)
    print('region rank WindOvlp=%d: %d windows ranked from %d positions in %.2f s' % (ovlp, len(idx), n, time.time() - t0), flush=True)
t0 = time.time()
o = engine.rank_order_host(p, np.round(p, 2), np.round(p, 1))
print('3-key rank order of %d records: %.3f s (numpy lexsort: ' % (n, time.time() - t0), end='')
t0 = time.time(); o2 = np.lexsort((np.round(p, 1), np.round(p, 2), p)); print('%.3f s)' % (time.time() - t0))
assert np.array_equal(o, o2)
