import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import nanomod_amd as nm
L = nm._lib
P, N = 200000, 200
dev = 'cuda:0'
g = torch.Generator(device=dev); g.manual_seed(1)
a = torch.randn(P * N, dtype=torch.float64, device=dev, generator=g); b = torch.randn(P * N, dtype=torch.float64, device=dev, generator=g)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
for tests, label in ((L.TEST_KS, 'KS-only'), (L.TEST_ALL, 'all tests')):
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
    det.run(a, b, rid, stride0=N, stride1=N, npos=P); torch.cuda.synchronize()
    t0 = time.perf_counter(); det.run(a, b, rid, stride0=N, stride1=N, npos=P); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('fp64 keys', label, '%.1f ms for %d positions: %.3g positions/s' % (dt * 1e3, P, P / dt))
