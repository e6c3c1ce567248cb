import time, numpy as np, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import nanomod_amd as nm
rng = np.random.default_rng(1)
for n in (460_000, 4_600_000, 10_000_000):
    k1 = rng.random(n); k2 = rng.random(n); k3 = rng.random(n)
    nm.engine.rank_order_host(k1, k2, k3)
    t0 = time.perf_counter(); o = nm.engine.rank_order_host(k1, k2, k3); t = time.perf_counter() - t0
    print('n %d  rank_order_host %.1f ms (host arrays in, order out)' % (n, t * 1e3), bool(np.all(np.diff(k1[o]) >= 0)))
