"""float64 input (NMOD_DTYPE_F64): arbitrary doubles, 200 v 200, device-resident; KS-only and all tests.
Also a mixed batch (half the positions on the 0.001 grid) to show the per-position decision."""
import os, sys, time, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import nanomod_amd as nm
L = nm._lib
P, N = 1000000, 200
dev = 'cuda:0'
g = torch.Generator(device=dev); g.manual_seed(1)
a = torch.randn(P * N, dtype=torch.float64, device=dev, generator=g); b = torch.randn(P * N, dtype=torch.float64, device=dev, generator=g)
rid = torch.zeros(P, dtype=torch.int32, device=dev)
a2 = a.clone(); b2 = b.clone()
# (tensor / tensor is a true division; tensor / 1000 multiplies by 0.001, which is NOT the grid value k / 1000.0 of
# round(x, 3) and would make these positions arbitrary doubles full of exact ties: the 64-bit-key fallback)
thousand = torch.full((), 1000.0, dtype=torch.float64, device=dev)
a2[:P * N // 2] = torch.round(a2[:P * N // 2] * 1000) / thousand; b2[:P * N // 2] = torch.round(b2[:P * N // 2] * 1000) / thousand
for name, x, y in (('arbitrary doubles', a, b), ('half the positions on the 0.001 grid', a2, b2)):
    for tests, label in ((L.TEST_KS, 'KS-only'), (L.TEST_ALL, 'all tests')):
        det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=tests)
        det.run(x, y, rid, stride0=N, stride1=N, npos=P); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): det.run(x, y, rid, stride0=N, stride1=N, npos=P)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print('float64 input, %s, %s: %.1f ms for %d positions: %.3g positions/s' % (name, label, dt * 1e3, P, P / dt))
