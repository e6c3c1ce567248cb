#!/usr/bin/env python3
"""The host-resident entry on float64 rows on the milli-unit grid (narrowed to int16 by the copy threads): positions/s against the
number of copy threads, for the library NMOD_HIP_LIB names.  usage: python tools/sweep_narrow_threads.py [positions] [threads ...]"""
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
npos = int(sys.argv[1]) if len(sys.argv) > 1 else 1_150_000
threads = [int(t) for t in sys.argv[2:]] or [4, 8, 12, 16, 8]
n0 = n1 = 200
det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS)
q = [torch.empty(npos * n, dtype=torch.int16, device='cuda:0') for n in (n0, n1)]
for g in (0, 1):
    det.synth_fill(q[g], 20240601, 0, npos, g, (n0, n1)[g], 10000, 0.8)
a, b = [x.cpu().numpy().astype(np.float64) / 1000.0 for x in q]
del q
rid = np.zeros(npos, np.int32)
# the conversion alone, one thread
out16 = np.empty(1 << 24, np.int16)
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    ok = lib.nmod_narrow_probe(a.ctypes.data_as(ctypes.c_void_p), 1 << 24, out16.ctypes.data_as(ctypes.c_void_p))
    best = min(best, time.perf_counter() - t0)
print('%s: conversion alone, one thread: %.2f G samples/s (ok %d)' % (os.path.basename(L.LIB_PATH), (1 << 24) / best / 1e9, ok), flush=True)
for tests, method, tl in ((L.TEST_KS, 'stouffer', 'KS+Stouffer'), (L.TEST_ALL, 'fisher', 'all+Fisher')):
    for t in threads:
        lib.nmod_host_pipeline_config(0, 0, t, 0)
        out = None
        best = 1e9
        for rep in range(4):
            t0 = time.perf_counter()
            out = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method=method, tests=tests, stride0=n0, stride1=n1, out=out)
            best = min(best, time.perf_counter() - t0)
        st = L.NmodHostStats(); lib.nmod_last_host_stats(ctypes.byref(st))
        print('%-12s threads %2d (used %d)  %.1f ms  %.2fe7 pos/s  %.1f G samples/s  narrowed %d of %d chunks' % (
            tl, t, st.copy_threads, best * 1e3, npos / best / 1e7, npos * (n0 + n1) / best / 1e9, st.narrowed_chunks, st.chunks), flush=True)
