"""KS + Stouffer on the bench workload with int16 milli-unit input (what NanoMod's 3-dp Events become) against float32."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import nanomod_amd as nm
L = nm._lib
P, N = 4_600_000, 200
dev = 'cuda:0'
for tests, label in ((L.TEST_KS, 'KS-only'), (L.TEST_ALL, 'all tests')):
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
    for dt in (torch.float32, torch.int16):
        sig0 = torch.empty(P * N, dtype=dt, device=dev); sig1 = torch.empty(P * N, dtype=dt, device=dev)
        det.synth_fill(sig0, 1, 0, P, 0, N, 10000, 0.8); det.synth_fill(sig1, 1, 0, P, 1, N, 10000, 0.8)
        rid = torch.zeros(P, dtype=torch.int32, device=dev)
        out = det.alloc_outputs(P)
        det.timer = nm.EventTimer(64)
        for _ in range(8): det.run(sig0, sig1, rid, out=out, stride0=N, stride1=N, npos=P)
        det.timer.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): det.run(sig0, sig1, rid, out=out, stride0=N, stride1=N, npos=P)
        torch.cuda.synchronize(); dtm = (time.perf_counter() - t0) / 20
        k1, n = det.timer.read(L.KERNEL_RANK_STATS)
        print(label, str(dt).split('.')[-1], 'ms/step %.3f' % (dtm * 1e3), 'K1 ms %.3f' % (k1 / n), 'Mpos/s %.1f' % (P / dtm / 1e6), flush=True)
        del sig0, sig1
