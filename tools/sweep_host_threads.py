import ctypes, os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import nanomod_amd as nm
L = nm._lib; lib = L.load()
npos = 4_600_000; n0 = n1 = 200
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
rid = np.zeros(npos, np.int32)
for dtype in (torch.float32, torch.int16):
    a = torch.empty(npos * n0, dtype=dtype, device='cuda:0'); b = torch.empty(npos * n1, dtype=dtype, device='cuda:0')
    det.synth_fill(a, 1, 0, npos, 0, n0, 10000, 0.8); det.synth_fill(b, 1, 0, npos, 1, n1, 10000, 0.8)
    a = a.cpu().numpy(); b = b.cpu().numpy()
    out = None
    for threads in (4, 6, 8, 12, 4):
        lib.nmod_host_pipeline_config(0, 0, threads, 0)
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            out = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, stride0=n0, stride1=n1, out=out)
            best = min(best, time.perf_counter() - t0)
        print(str(dtype), 'threads', threads, '%.1f ms  %.1f GB/s' % (best * 1e3, (a.nbytes + b.nbytes) / best / 1e9), flush=True)
