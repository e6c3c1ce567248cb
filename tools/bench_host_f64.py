#!/usr/bin/env python3
"""The host-resident entry on float64 rows (what the reference holds, myDetect.py:124): BASELINE configs[1]'s values as float64 —
3-decimal (NanoMod's events) and continuous — against the pinned copy rate.  usage: python tools/bench_host_f64.py [positions]"""
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
q = [torch.empty(npos * n, dtype=torch.int16, device='cuda:0') for n in (n0, n1)]
f = [torch.empty(npos * n, dtype=torch.float32, device='cuda:0') for n in (n0, n1)]
for g in (0, 1):
    det.synth_fill(q[g], 20240601, 0, npos, g, (n0, n1)[g], 10000, 0.8)
    det.synth_fill(f[g], 20240601, 0, npos, g, (n0, n1)[g], 10000, 0.8)
rows = {'3-decimal float64 (k / 1000.0)': [x.cpu().numpy().astype(np.float64) / 1000.0 for x in q],
        'continuous float64 (float32-exact)': [x.cpu().numpy().astype(np.float64) for x in f]}
rows['continuous float64 (not float32-exact)'] = [x * (1.0 + 2.0 ** -40) for x in rows['continuous float64 (float32-exact)']]
del q, f
for label, (a, b) in rows.items():
    for tests, method, tl in ((L.TEST_KS, 'stouffer', 'KS+Stouffer'), (L.TEST_ALL, 'fisher', 'all+Fisher')):
        out = None
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            out = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method=method, tests=tests, stride0=n0, stride1=n1, out=out)
            best = min(best, time.perf_counter() - t0)
        st = L.NmodHostStats(); lib.nmod_last_host_stats(ctypes.byref(st))
        print('%-40s %-12s %.1f ms  %.2fe7 pos/s  H2D %.1f GB/s = %.3f of pinned  chunks %d' % (label, tl, best * 1e3, npos / best / 1e7, st.h2d_bytes / best / 1e9,
                                                                                         st.h2d_bytes / best / 1e9 / peak, st.chunks), flush=True)
