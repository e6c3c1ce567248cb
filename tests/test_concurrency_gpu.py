"""-m gpu: what include/nanomod_hip.h promises about re-entrancy and error paths.  Two host threads x two streams on one
device run nmod_detect_batch at the same time (device-resident batches that need the library-owned scratch pool and the
float64 front end, and the host-resident pipeline, whose cached ring only one call at a time can take) and get the serial
results; the error returns (workspace too small, a group beyond what the caller promised, a group beyond NMOD_MAX_RANKED)
leave nothing allocated behind."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _batch(rng, sizes0, sizes1, dtype):
    off0 = np.zeros(len(sizes0) + 1, np.int64); off1 = np.zeros(len(sizes1) + 1, np.int64)
    np.cumsum(sizes0, out=off0[1:]); np.cumsum(sizes1, out=off1[1:])
    a = np.round(rng.normal(0, 1, off0[-1]), 3); b = np.round(rng.normal(0.1, 1, off1[-1]), 3)
    if dtype == np.float64:                          # off the grid and not float32-exact: class 3 keys, ties -> the 64-bit redo
        a = a + 1e-11; b = b + 1e-11
    return a.astype(dtype), off0, b.astype(dtype), off1


def _cases():
    rng = np.random.default_rng(5)
    big0 = np.array([40, 3000, 64, 5000, 200, 2500, 90, 30] * 8); big1 = np.array([50, 2800, 64, 100, 210, 2600, 9000, 35] * 8)
    small0 = rng.integers(5, 300, 4000); small1 = rng.integers(5, 300, 4000)
    return {
        'large_positions_f32': (_batch(rng, big0, big1, np.float32), 7, 'stouffer'),          # big_rank_kernel: scratch from the pool
        'float64': (_batch(rng, small0[:1500], small1[:1500], np.float64), 7, 'fisher'),      # f64 front end: pool + redo round trip
        'ragged_f32_ks': (_batch(rng, small0, small1, np.float32), 1, 'stouffer'),
        'ragged_i16': (tuple(x if i % 2 else np.rint(x * 1000).astype(np.int16) for i, x in enumerate(_batch(rng, small0, small1, np.float32))), 7, 'stouffer'),
    }


def _run_device(case, stream=None):
    import torch
    import nanomod_amd as nm
    (a, off0, b, off1), tests, method = case
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method=method, tests=tests)
    rid = torch.zeros(len(off0) - 1, dtype=torch.int32, device='cuda:0')
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        t = [torch.from_numpy(x).cuda() for x in (a, off0, b, off1)]
        r = det.run(t[0], t[2], rid, off0=t[1], off1=t[3])
        out = {k: v.cpu().numpy() for k, v in r.items()}
    return out


def _equal(x, y):
    return all(np.array_equal(x[k], y[k], equal_nan=True) for k in x)


def test_two_threads_two_streams_device_resident():
    import torch
    cases = _cases()
    serial = {k: _run_device(v) for k, v in cases.items()}
    assert not serial['large_positions_f32']['status'].any()
    errors = []

    def worker(names, stream):
        try:
            for it in range(6):
                for n in names:
                    got = _run_device(cases[n], stream)
                    if not _equal(got, serial[n]):
                        errors.append((n, it))
        except Exception as e:                       # noqa: BLE001 — reported below
            errors.append(repr(e))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    names = list(cases)
    th = [threading.Thread(target=worker, args=(names, s1)), threading.Thread(target=worker, args=(names[::-1], s2))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors


def test_two_threads_host_pipeline_and_device_path_together():
    import nanomod_amd as nm
    L = nm._lib
    lib = L.load()
    cases = _cases()
    assert lib.nmod_host_pipeline_config(64 << 10, 3, 2, 0) == 0          # many chunks per call
    try:
        serial = {}
        for n, ((a, off0, b, off1), tests, method) in cases.items():
            serial[n] = nm.detect_host(a, off0, b, off1, np.zeros(len(off0) - 1, np.int32), nb=2, weights_dif=2.0, method=method, tests=tests)
        errors = []

        def host_worker(names):
            try:
                for it in range(4):
                    for n in names:
                        (a, off0, b, off1), tests, method = cases[n]
                        got = nm.detect_host(a, off0, b, off1, np.zeros(len(off0) - 1, np.int32), nb=2, weights_dif=2.0, method=method, tests=tests)
                        if not _equal(got, serial[n]):
                            errors.append(('host', n, it))
            except Exception as e:                   # noqa: BLE001
                errors.append(repr(e))

        def dev_worker():
            import torch
            try:
                s = torch.cuda.Stream()
                for it in range(4):
                    for n in cases:
                        got = _run_device(cases[n], s)
                        ref = serial[n]
                        if not all(np.array_equal(got[k], ref[k], equal_nan=True) for k in ref):
                            errors.append(('device', n, it))
            except Exception as e:                   # noqa: BLE001
                errors.append(repr(e))
        names = list(cases)
        th = [threading.Thread(target=host_worker, args=(names,)), threading.Thread(target=host_worker, args=(names[::-1],)),
              threading.Thread(target=dev_worker)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errors, errors
    finally:
        lib.nmod_host_pipeline_config(0, 0, 0, 0)


def test_error_returns_leak_nothing():
    import torch
    import nanomod_amd as nm
    L = nm._lib
    lib = L.load()
    rng = np.random.default_rng(2)
    (a, off0, b, off1) = _batch(rng, np.array([30, 3000, 50, 70000, 20]), np.array([30, 40, 2500, 60, 20]), np.float32)
    t = [torch.from_numpy(x).cuda() for x in (a, off0, b, off1)]
    rid = torch.zeros(5, dtype=torch.int32, device='cuda:0')
    det = nm.DeviceDetector(0, nb=2, weights_dif=2.0, method='stouffer', tests=7)
    res = det.alloc_outputs(5)
    o = L.NmodOut()
    for name in L.OUT_FIELDS:
        if name in res:
            setattr(o, name, res[name].data_ptr())
    o.status = res['status'].data_ptr()
    torch.cuda.synchronize()
    assert lib.nmod_trim_scratch(0) == 0
    free0 = torch.cuda.mem_get_info(0)[0]

    def call(prm, ws, ws_bytes, off0_t=t[1], npos=5):
        return lib.nmod_detect_batch(C.byref(prm), npos, t[0].data_ptr(), off0_t.data_ptr(), t[2].data_ptr(), t[3].data_ptr(), rid.data_ptr(),
                                     ws.data_ptr() if ws is not None else None, ws_bytes, C.byref(o))
    prm = det._params(L.DTYPE_F32, 0, 0, 0, 0)
    need = lib.nmod_workspace_bytes(C.byref(prm), 5)
    ws = torch.empty(need, dtype=torch.uint8, device='cuda:0')
    free0 = torch.cuda.mem_get_info(0)[0]
    # workspace missing / one byte short
    assert call(prm, None, 0) == -4 and call(prm, ws, need - 1) == -4
    assert b'workspace' in lib.nmod_strerror(-4)
    # a group of 70 000 samples: beyond NMOD_MAX_RANKED, found by the library's own reduction of the offsets (max_n unknown) —
    # since round 5 that position is skipped and flagged, the rest of the batch is computed
    assert call(prm, ws, need) == 0
    torch.cuda.synchronize()
    st = res['status'].cpu().numpy()
    assert (st[3] & L.STATUS_TOO_LARGE) and not (st[[0, 1, 2, 4]] & L.STATUS_TOO_LARGE).any()
    # the caller promises max_n0 = 4096 and breaks the promise mid-batch: the position is skipped and flagged, the rest is computed
    prm2 = det._params(L.DTYPE_F32, 0, 0, 4096, 4096)
    assert call(prm2, ws, need) == 0
    torch.cuda.synchronize()
    st = res['status'].cpu().numpy()
    assert (st[3] & L.STATUS_TOO_LARGE) and not st[[0, 1, 2, 4]].any()          # (+ T_NAN: the skipped position's t-test is NaN)
    assert np.isnan(res['ks_p'].cpu().numpy()[3]) and np.isfinite(res['ks_p'].cpu().numpy()[[0, 1, 2, 4]]).all()
    # the batch used the large-position scratch (positions 1 and 2): the pool holds it until the trim
    assert lib.nmod_trim_scratch(0) == 0
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info(0)[0] <= (2 << 20)
    # invalid arguments: rejected before any device work
    bad = det._params(L.DTYPE_F32, 0, 0, 0, 0); bad.nb = 65
    assert call(bad, ws, need) == -1
    bad = det._params(L.DTYPE_F32, 0, 0, 0, 0); bad.flags = 64
    assert call(bad, ws, need) == -1
    bad = det._params(L.DTYPE_F32, 0, 0, 0, 0); bad.device = 99
    assert call(bad, ws, need) == -5
    assert free0 - torch.cuda.mem_get_info(0)[0] <= (2 << 20)
