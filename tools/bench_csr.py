import sys, time, torch, numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import nanomod_amd as nm
L = nm._lib
P, N = 4_600_000, 200
dev = 'cuda:0'
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
sig0 = torch.empty(P * N, dtype=torch.float32, device=dev); sig1 = torch.empty(P * N, dtype=torch.float32, device=dev)
det.synth_fill(sig0, 1, 0, P, 0, N, 10000, 0.8); det.synth_fill(sig1, 1, 0, P, 1, N, 10000, 0.8)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
off = torch.arange(P + 1, dtype=torch.int64, device=dev) * N
out = det.alloc_outputs(P)
for mode in ('stride', 'csr'):
    det.timer = nm.EventTimer(64)
    kw = dict(stride0=N, stride1=N, npos=P) if mode == 'stride' else dict(off0=off, off1=off, max_n0=N, max_n1=N)
    for _ in range(5): det.run(sig0, sig1, rid, out=out, **kw)
    det.timer.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): det.run(sig0, sig1, rid, out=out, **kw)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    k1, n = det.timer.read(L.KERNEL_RANK_STATS)
    print(mode, 'ms/step %.3f' % (dt * 1e3), 'K1 ms %.3f' % (k1 / n), 'Mpos/s %.1f' % (P / dt / 1e6))
