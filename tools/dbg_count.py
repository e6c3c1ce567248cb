import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'oracle'))
import numpy as np, helpers as H, oracle_c
import nanomod_amd as nm
P = 400
for dt in ('i16', 'f32'):
    for n in (200, 64 * 3, 130):
        a = H.synth_events_ref(11, 4990, P, 0, n, 5000, 800, 200, dt); b = H.synth_events_ref(11, 4990, P, 1, n, 5000, 800, 200, dt)
        off = np.arange(0, (P + 1) * n, n, dtype=np.int64)
        rid = np.zeros(P, np.int32)
        exp = oracle_c.detect_batch(a.reshape(-1), off, b.reshape(-1), off, rid, 2, 2.0, 'fisher', tests=7)
        got = nm.detect_host(a.reshape(-1), None, b.reshape(-1), None, rid, nb=2, weights_dif=2.0, method='fisher', stride0=n, stride1=n)
        for k in ('mwu_u', 'mwu_p', 't_t', 'ks_d', 'ks_p'):
            e, g = exp[k], got[k]
            bad = ~((e == g) | (np.abs(e - g) <= 1e-9 * np.abs(e) + 1e-13))
            print(dt, n, k, 'bad', int(bad.sum()), 'first', np.flatnonzero(bad)[:5], g[bad][:3], e[bad][:3])
