#!/usr/bin/env python3
"""K1 time of one all-tests pass over a CSR batch of event-like int16 / float32 rows whose group sizes are uniform in [LO, HI]
(what build_csr hands over for real coverage): python3 tools/csr_shape_ab.py LO HI [positions] [dtype] [spread]; run it under
NMOD_NO_COUNTING=1 / NMOD_NO_COUNT_WIDE=1 for the A/B.  Checked against the oracle on the first 3 000 positions."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.environ.get('GRAFT_REPO_ROOT') or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import nanomod_amd as nm  # noqa: E402
L = nm._lib

lo, hi = int(sys.argv[1]), int(sys.argv[2])
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
dt = sys.argv[4] if len(sys.argv) > 4 else 'i16'
spread = int(sys.argv[5]) if len(sys.argv) > 5 else 200
dev = torch.device('cuda:0')
rng = np.random.default_rng(5)
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='fisher', tests=L.TEST_ALL)
sig, hoff, doff = [None, None], [None, None], [None, None]
for g in (0, 1):
    sz = rng.integers(lo, hi + 1, P)
    hoff[g] = np.zeros(P + 1, np.int64); np.cumsum(sz, out=hoff[g][1:])
    doff[g] = torch.from_numpy(hoff[g]).to(dev)
    sig[g] = torch.empty(int(hoff[g][-1]), dtype=torch.int16 if dt == 'i16' else torch.float32, device=dev)
    if spread:
        det.synth_fill_events(sig[g], 7, 0, P, g, n_per_pos=0, off=doff[g], plant_period=10000, plant_shift_milli=800, spread_milli=spread)
    else:
        det.synth_fill_csr(sig[g], 7, 0, doff[g], g, 10000, 0.8)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
outs = det.alloc_outputs(P)


def run_once():
    det.run(sig[0], sig[1], rid, off0=doff[0], off1=doff[1], max_n0=hi, max_n1=hi, out=outs)


run_once(); torch.cuda.synchronize()
import oracle_c  # noqa: E402
import helpers as H  # noqa: E402
vn = 3000
a = sig[0][:int(hoff[0][vn])].cpu().numpy(); b = sig[1][:int(hoff[1][vn])].cpu().numpy()
exp = oracle_c.detect_batch(a, hoff[0][:vn + 1], b, hoff[1][:vn + 1], np.zeros(vn, np.int32), 2, 2.0, 'fisher', tests=7)
got = {k: outs[k][:vn - 2].cpu().numpy() for k in ('mwu_u', 'ks_d', 'ks_p', 'mwu_p', 't_p')}
ok = all(np.array_equal(got[k], exp[k][:vn - 2], equal_nan=True) for k in ('mwu_u', 'ks_d')) and \
    all(np.allclose(got[k], exp[k][:vn - 2], rtol=1e-9, atol=1e-300, equal_nan=True) for k in ('ks_p', 'mwu_p', 't_p'))
tm = nm.EventTimer(64); det.timer = tm
steps = 5
t0 = time.perf_counter()
for _ in range(steps):
    run_once()
torch.cuda.synchronize()
el = time.perf_counter() - t0
k1, _ = tm.read(L.KERNEL_RANK_STATS)
print('sizes [%d..%d] %s spread %d counting=%s wide=%s  %.4g pos/s  K1 %.3f ms  verify %s' % (
    lo, hi, dt, spread, 'off' if os.environ.get('NMOD_NO_COUNTING') else 'on', 'off' if os.environ.get('NMOD_NO_COUNT_WIDE') else 'on', P * steps / el, k1 / steps, ok))
