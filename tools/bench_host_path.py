#!/usr/bin/env python3
"""Sweep of the NMOD_MEM_HOST pipeline's tunables on BASELINE configs[1] rows held in host memory:
chunk size x ring depth x copy threads, pageable and page-locked input, float32 and int16.
usage: python tools/bench_host_path.py [positions]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import nanomod_amd as nm

L = nm._lib
lib = L.load()
npos = int(sys.argv[1]) if len(sys.argv) > 1 else 4_600_000
n0 = n1 = 200
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)


def rows(dtype):
    a = torch.empty(npos * n0, dtype=dtype, device='cuda:0'); b = torch.empty(npos * n1, dtype=dtype, device='cuda:0')
    det.synth_fill(a, 20240601, 0, npos, 0, n0, 10000, 0.8); det.synth_fill(b, 20240601, 0, npos, 1, n1, 10000, 0.8)
    return a.cpu().numpy(), b.cpu().numpy()


pin = torch.empty(1 << 28, dtype=torch.float32).pin_memory()
dst = torch.empty(1 << 28, dtype=torch.float32, device='cuda:0')
dst.copy_(pin, non_blocking=True); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    dst.copy_(pin, non_blocking=True)
e1.record(); torch.cuda.synchronize()
peak = 3 * pin.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
print('pinned H2D %.2f GB/s' % peak)
del pin, dst
rid = np.zeros(npos, np.int32)
for dtype in (torch.float32, torch.int16):
    a, b = rows(dtype)
    out = None
    for label, tests, method in (('KS+Stouffer', L.TEST_KS, 'stouffer'), ('all+Fisher', L.TEST_ALL, 'fisher')):
        out = None
        for pinned in (False, True):
            if pinned:
                rt = torch.cuda.cudart()
                assert all(int(rt.cudaHostRegister(x.ctypes.data, x.nbytes, 0)) == 0 for x in (a, b))
            for chunk_mb, slots, threads in ((32, 3, 4), (64, 3, 4), (16, 3, 4), (32, 4, 4), (32, 3, 2), (32, 3, 8), (64, 4, 8), (128, 3, 4)):
                if pinned and threads != 4:
                    continue
                assert lib.nmod_host_pipeline_config(chunk_mb << 20, slots, threads, 0) == 0
                best = 1e9
                for rep in range(3):
                    t0 = time.perf_counter()
                    out = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method=method, tests=tests, stride0=n0, stride1=n1, out=out)
                    best = min(best, time.perf_counter() - t0)
                st = L.NmodHostStats(); lib.nmod_last_host_stats(ctypes.byref(st))
                print('%-8s %-12s %-8s chunk %3d MiB slots %d threads %d: %.1f ms  %.2fe7 pos/s  H2D %.1f GB/s = %.3f of pinned  (dev %d MB, pinned ring %d MB)'
                      % (str(dtype).split('.')[1], label, 'pinned' if pinned else 'pageable', chunk_mb, slots, st.copy_threads, best * 1e3, npos / best / 1e7,
                         st.h2d_bytes / best / 1e9, st.h2d_bytes / best / 1e9 / peak, st.device_bytes >> 20, st.pinned_bytes >> 20), flush=True)
            if pinned:
                for x in (a, b):
                    rt.cudaHostUnregister(x.ctypes.data)
    # fresh result arrays every call (what a one-shot caller pays)
    lib.nmod_host_pipeline_config(0, 0, 0, 0)
    t0 = time.perf_counter()
    nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, stride0=n0, stride1=n1)
    print('%s default config, fresh result arrays: %.1f ms' % (str(dtype).split('.')[1], (time.perf_counter() - t0) * 1e3))
