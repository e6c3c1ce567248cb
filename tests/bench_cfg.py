#!/usr/bin/env python3
"""Side measurements of the other BASELINE.json configs on one GPU (not the headline bench):
cfg4-like (500 v 500, fixed stride) and cfg5 (ragged ~1000 v ~50, lognormal sizes, CSR + size-class binning)."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import nanomod_amd as nm
L = nm._lib

def run(name, det, sig0, sig1, rid, steps=5, **kw):
    for _ in range(2): det.run(sig0, sig1, rid, **kw)
    det.timer = nm.EventTimer(64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): det.run(sig0, sig1, rid, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    k1, n = det.timer.read(L.KERNEL_RANK_STATS); det.timer = None
    npos = rid.numel()
    nbytes = sig0.numel() * sig0.element_size() + sig1.numel() * sig1.element_size()
    print(json.dumps({'config': name, 'positions': npos, 'positions_per_s': npos / dt, 'ms_per_step': dt * 1e3,
                      'k1_ms': k1 / n, 'input_GBps_over_k1': nbytes / (k1 / n * 1e-3) / 1e9}))

ap = argparse.ArgumentParser(); ap.add_argument('--all-tests', action='store_true')
ap.add_argument('--cap0', type=int, default=4000, help='clip of the cfg5 group-1 sizes'); a = ap.parse_args()
tests = L.TEST_ALL if a.all_tests else L.TEST_KS
dev = 'cuda:0'
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
# cfg4-like: 500 v 500, 2 M positions (8 GB)
P, n = 2_000_000, 500
s0 = torch.empty(P * n, dtype=torch.float32, device=dev); s1 = torch.empty(P * n, dtype=torch.float32, device=dev)
det.synth_fill(s0, 1, 0, P, 0, n, 10000, 0.8); det.synth_fill(s1, 1, 0, P, 1, n, 10000, 0.8)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
run('cfg4-like 500v500 stride', det, s0, s1, rid, stride0=n, stride1=n, npos=P)
del s0, s1
# cfg5: ragged lognormal sizes, 1 M positions
P = 1_000_000
rng = np.random.default_rng(5)
cap0 = a.cap0                                 # clipped [5, 4000]; groups beyond 2048 take big_rank_kernel in all-tests mode
n0 = np.clip(np.round(rng.lognormal(np.log(1000), 0.5, P)), 5, cap0).astype(np.int64)
n1 = np.clip(np.round(rng.lognormal(np.log(50), 0.5, P)), 5, 400).astype(np.int64)
off0 = np.zeros(P + 1, np.int64); off0[1:] = np.cumsum(n0)
off1 = np.zeros(P + 1, np.int64); off1[1:] = np.cumsum(n1)
g = torch.Generator(device=dev); g.manual_seed(1)
s0 = torch.randn(int(off0[-1]), dtype=torch.float32, device=dev, generator=g)
s1 = torch.randn(int(off1[-1]), dtype=torch.float32, device=dev, generator=g)
o0 = torch.from_numpy(off0).to(dev); o1 = torch.from_numpy(off1).to(dev)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
run('cfg5 ragged ~1000v~50 CSR', det, s0, s1, rid, off0=o0, off1=o1, max_n0=cap0, max_n1=400)
# spot-check a few positions against the oracle
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import nanomod_oracle as orc
res = det.run(s0, s1, rid, off0=o0, off1=o1, max_n0=cap0, max_n1=400); torch.cuda.synchronize()
h0 = s0.cpu().numpy(); h1 = s1.cpu().numpy(); ksp = res['ks_p'].cpu().numpy(); ksd = res['ks_d'].cpu().numpy()
for i in list(range(0, P, P // 50)):
    d, p = orc.ks_2samp(h0[off0[i]:off0[i + 1]], h1[off1[i]:off1[i + 1]])
    assert abs(ksd[i] - d) <= 0.0 and abs(ksp[i] - max(p, orc.DBL_MIN)) <= 1e-9 * p, (i, ksd[i], d)
print('cfg5 spot-check vs oracle ok')
