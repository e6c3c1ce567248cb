"""-m gpu: the host-resident entry (NMOD_MEM_HOST, host_pipeline.hpp) — what a drop-in mtest2 takes, everything on this path
is host memory in the reference (myDetect.py:416-445).  The chunked, overlapped path must return what the device-resident
path returns on the same rows, bit for bit, wherever the chunk cuts fall: inside runs, at run edges, one position per chunk;
from pageable and from page-locked arrays; with a device footprint bounded by the chunk size."""
import ctypes as C

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def _lib():
    import nanomod_amd._lib as L
    return L, L.load()


def _stats():
    L, lib = _lib()
    st = L.NmodHostStats()
    assert lib.nmod_last_host_stats(C.byref(st)) == 0
    return st


@pytest.fixture(autouse=True)
def _default_config():
    L, lib = _lib()
    yield
    assert lib.nmod_host_pipeline_config(0, 0, 0, 0) == 0


def _device_reference(sig0, off0, sig1, off1, rid, *, nb, method, tests, stride0=0, stride1=0, want_mstd=False):
    """the same rows through the device-resident entry (one launch over the whole batch, K3 over the whole track)"""
    import torch
    import nanomod_amd as nm
    det = nm.DeviceDetector(0, nb=nb, weights_dif=2.0, method=method, tests=tests, want_mstd=want_mstd)
    t = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a)).cuda()
    npos = len(rid)
    r = det.run(t(sig0), t(sig1), t(rid), off0=t(off0), off1=t(off1), stride0=stride0, stride1=stride1, npos=npos)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in r.items()}


def _ragged(rng, npos, lo, hi, dtype):
    n0 = rng.integers(lo, hi, npos); n1 = rng.integers(lo, hi, npos)
    off0 = np.zeros(npos + 1, np.int64); off1 = np.zeros(npos + 1, np.int64)
    np.cumsum(n0, out=off0[1:]); np.cumsum(n1, out=off1[1:])
    a = np.round(rng.normal(0, 1, off0[-1]), 2); b = np.round(rng.normal(0.1, 1, off1[-1]), 2)     # 0.01 grid: ties
    if dtype == np.int16:
        return np.rint(a * 1000).astype(np.int16), off0, np.rint(b * 1000).astype(np.int16), off1
    return a.astype(dtype), off0, b.astype(dtype), off1


def _runs(rng, npos):
    """run ids with edges everywhere: runs of 1..40 positions"""
    rid = np.zeros(npos, np.int32)
    i, r = 0, 0
    while i < npos:
        ln = int(rng.integers(1, 40))
        rid[i:i + ln] = r
        i += ln; r += 1
    return rid


def _same(got, ref, names):
    for k in names:
        assert np.array_equal(got[k], ref[k], equal_nan=True), (k, int(np.sum(got[k] != ref[k])))


@pytest.mark.parametrize('dtype', [np.float32, np.int16, np.float64])
@pytest.mark.parametrize('tests_mask,method', [(7, 'stouffer'), (1, 'fisher'), (7, 'ks')])
def test_chunked_equals_device_resident_csr(dtype, tests_mask, method):
    import nanomod_amd as nm
    L, lib = _lib()
    rng = np.random.default_rng(11 + tests_mask)
    npos = 3000
    sig0, off0, sig1, off1 = _ragged(rng, npos, 5, 120, dtype)
    rid = _runs(rng, npos)
    ref = _device_reference(sig0, off0, sig1, off1, rid, nb=2, method=method, tests=tests_mask)
    names = [k for k in ref if k != 'status'] + ['status']
    flags = L.FLAG_NO_HOST_NARROW if dtype == np.float64 else 0
    row_bytes = (off0[-1] + off1[-1]) * sig0.itemsize / npos
    # chunk cuts: ~7 positions per chunk (cuts inside runs and on run edges), ~300 per chunk, and one position per chunk
    for chunk_bytes, slots in ((int(7 * row_bytes), 2), (int(300 * row_bytes), 3), (1, 4)):
        assert lib.nmod_host_pipeline_config(chunk_bytes, slots, 2, 0) == 0
        # (float64 rows on the 0.01 grid: sent as float64 here, like the device-resident reference; narrowed to int16 below)
        got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method=method, tests=tests_mask, flags=flags)
        st = _stats()
        assert st.chunks >= npos // 400 and st.slots == slots and st.pinned_input == 0 and st.narrowed_chunks == 0
        if chunk_bytes == 1:
            assert st.chunks == npos and st.chunk_positions == 1
        _same(got, ref, [k for k in names if k in got])
        if dtype == np.float64:
            # the bounce fill narrows every chunk to int16 milli-units: a quarter of the bytes over the bus, the rank statistics and
            # their p-values bit for bit, the Welch pair from exact integer sums (within the gates every path is held to)
            nar = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method=method, tests=tests_mask)
            st2 = _stats()
            assert st2.narrowed_chunks == st2.chunks == st.chunks and st.h2d_bytes - st2.h2d_bytes == 6 * int(off0[-1] + off1[-1])       # 2 bytes per sample instead of 8
            _same(nar, ref, [k for k in names if k in nar and k not in ('t_t', 't_p')])
            if 't_t' in nar:
                H.assert_close_stat(nar['t_t'], ref['t_t'], 1e-11, H.t_abs_gate(sig0, off0, sig1, off1), 't_t')
                H.assert_close_p(nar['t_p'], ref['t_p'], 1e-9, 't_p')
    # the default configuration: a small batch is cut into ~32 chunks of >= 1 MiB (here: one or two chunks)
    assert lib.nmod_host_pipeline_config(0, 0, 0, 0) == 0
    got = nm.detect_host(sig0, off0, sig1, off1, rid, nb=2, weights_dif=2.0, method=method, tests=tests_mask, flags=flags)
    _same(got, ref, [k for k in names if k in got])


@pytest.mark.parametrize('dtype', [np.float32, np.int16])
def test_chunked_equals_device_resident_stride_and_oracle(dtype):
    """fixed stride 200 v 200 (BASELINE configs[1] shape), windows across every chunk cut, against the device path AND the oracle"""
    import nanomod_amd as nm
    import oracle_c
    L, lib = _lib()
    npos, n0, n1 = 6000, 200, 200
    a = H.synth_ref(5, 0, npos, 0, n0, 1000, 0.8, 'i16' if dtype == np.int16 else 'f32')
    b = H.synth_ref(5, 0, npos, 1, n1, 1000, 0.8, 'i16' if dtype == np.int16 else 'f32')
    rid = np.zeros(npos, np.int32)
    rid[2500:] = 1; rid[2501:] = 2                  # a one-position run between two long ones
    ref = _device_reference(a, None, b, None, rid, nb=2, method='stouffer', tests=7, stride0=n0, stride1=n1, want_mstd=True)
    assert lib.nmod_host_pipeline_config(101 * (n0 + n1) * a.itemsize, 3, 3, 0) == 0     # 101 positions per chunk: cuts at 101, 202, ...
    got = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n0, stride1=n1, want_mstd=True)
    st = _stats()
    assert st.chunks == (npos + 100) // 101 and st.chunk_positions == 101
    _same(got, ref, list(got))
    off0 = np.arange(0, (npos + 1) * n0, n0, dtype=np.int64); off1 = np.arange(0, (npos + 1) * n1, n1, dtype=np.int64)
    exp = oracle_c.detect_batch(a, off0, b, off1, rid, 2, 2.0, 'stouffer', tests=7)
    H.compare_outputs(got, exp)


def test_float64_chunks_narrow_only_where_every_sample_allows():
    """float64 rows as the reference holds them (myDetect.py:124), a batch of four kinds of positions — 3-decimal events, continuous
    doubles, 3-decimal values beyond the int16 range (|k| up to 40 000), 3-decimal events with a NaN — cut into chunks that are
    pure or mixed: a chunk is narrowed to int16 on the host only when EVERY sample of it is k / 1000.0 with |k| <= 32 767, and the
    results are those of the all-float64 path (NMOD_FLAG_NO_HOST_NARROW): rank statistics, p-values and status bits bit for bit,
    the Welch pair within its gate; all against the oracle"""
    import nanomod_amd as nm
    import nanomod_oracle as orc
    L, lib = _lib()
    rng = np.random.default_rng(23)
    per, n = 400, 60                                              # positions per kind, samples per group
    kinds = []
    kinds.append(np.round(rng.normal(0, 1, (per, 2, n)) * 0.3 + rng.uniform(-3, 3, (per, 1, 1)), 3))                  # events
    kinds.append(rng.normal(0, 1, (per, 2, n)))                                                                       # continuous
    kinds.append(np.round(rng.normal(0, 1, (per, 2, n)) * 12.0, 3))                                                   # |k| up to ~40 000
    ev = np.round(rng.normal(0, 1, (per, 2, n)) * 0.3, 3); ev[per // 2, 1, 7] = np.nan; kinds.append(ev)              # a NaN in one position
    x = np.concatenate(kinds)                                     # [4 per, 2, n]
    npos = x.shape[0]
    a = np.ascontiguousarray(x[:, 0, :]).reshape(-1); b = np.ascontiguousarray(x[:, 1, :]).reshape(-1)
    rid = np.zeros(npos, np.int32)
    # 100 positions per chunk: 4 pure chunks per kind (the NaN spoils one of the last kind's)
    assert lib.nmod_host_pipeline_config(100 * 2 * n * 8, 3, 2, 0) == 0
    got = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n, stride1=n, flags=L.FLAG_CHECK_FINITE)
    st = _stats()
    assert st.chunks == 16 and st.narrowed_chunks == 4 + 3, (st.chunks, st.narrowed_chunks)
    ref = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n, stride1=n, flags=L.FLAG_CHECK_FINITE | L.FLAG_NO_HOST_NARROW)
    st0 = _stats()
    assert st0.narrowed_chunks == 0 and st.h2d_bytes == st0.h2d_bytes - 7 * 100 * 2 * n * 6
    nanpos = 3 * per + per // 2
    assert (got['status'][nanpos] & L.STATUS_NONFINITE) and np.array_equal(got['status'], ref['status'])
    keep = np.ones(npos, bool); keep[nanpos - 2:nanpos + 3] = False                 # (the NaN position's own numbers are unspecified, its window's combined pair with them)
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p'):
        assert np.array_equal(got[k][keep], ref[k][keep]), k
    H.assert_close_p(got['t_p'][keep], ref['t_p'][keep], 1e-9, 't_p')
    # mixed chunks (150 positions per chunk: every second chunk holds two kinds) narrow only where both kinds allow
    assert lib.nmod_host_pipeline_config(150 * 2 * n * 8, 3, 2, 0) == 0
    mix = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n, stride1=n, flags=L.FLAG_CHECK_FINITE)
    assert 1 <= _stats().narrowed_chunks <= 4
    for k in ('mwu_u', 'mwu_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p'):
        assert np.array_equal(mix[k][keep], ref[k][keep]), k
    off = np.arange(0, (npos + 1) * n, n, dtype=np.int64)
    fin = np.arange(npos) < 3 * per
    exp = orc.detect_batch(a[:3 * per * n], off[:3 * per + 1], b[:3 * per * n], off[:3 * per + 1], rid[:3 * per], 2, 2.0, orc.METHOD_STOUFFER)
    inner = slice(0, 3 * per - 2)
    H.compare_outputs({k: v[fin][inner] for k, v in got.items()}, {k: v[inner] for k, v in exp.items()}, True,
                      t_abs=H.t_abs_gate(a[:3 * per * n], off[:3 * per + 1], b[:3 * per * n], off[:3 * per + 1])[inner])


def test_pinned_input_is_copied_from_where_it_is():
    import torch
    import nanomod_amd as nm
    L, lib = _lib()
    rng = np.random.default_rng(3)
    npos, n0, n1 = 20000, 60, 50
    pa = torch.empty(npos * n0, dtype=torch.float32).pin_memory()
    pb = torch.empty(npos * n1, dtype=torch.float32).pin_memory()
    a = pa.numpy(); b = pb.numpy()
    a[:] = rng.normal(0, 1, a.size).astype(np.float32); b[:] = rng.normal(0.2, 1, b.size).astype(np.float32)
    rid = _runs(rng, npos)
    assert lib.nmod_host_pipeline_config(256 << 10, 3, 0, 0) == 0
    got = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n0, stride1=n1)
    st = _stats()
    assert st.pinned_input == 1 and st.copy_threads == 1 and st.chunks > 10
    assert st.h2d_bytes == a.nbytes + b.nbytes                          # every input byte crosses the bus once
    # the same arrays as pageable copies (bounce slots), and forced through the bounce slots although pinned
    got2 = nm.detect_host(a.copy(), None, b.copy(), None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n0, stride1=n1)
    assert _stats().pinned_input == 0
    assert lib.nmod_host_pipeline_config(256 << 10, 3, 0, 2) == 0
    got3 = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', stride0=n0, stride1=n1)
    assert _stats().pinned_input == 0
    ref = _device_reference(a, None, b, None, rid, nb=2, method='stouffer', tests=7, stride0=n0, stride1=n1)
    for g in (got, got2, got3):
        _same(g, ref, list(g))


def test_device_footprint_is_bounded_by_the_chunk_not_the_batch():
    import torch
    import nanomod_amd as nm
    L, lib = _lib()
    npos, n0, n1 = 400_000, 200, 200                                    # 640 MB of samples
    a = H.synth_ref(9, 0, npos, 0, n0); b = H.synth_ref(9, 0, npos, 1, n1)
    rid = np.zeros(npos, np.int32)
    chunk = 8 << 20
    assert lib.nmod_host_pipeline_config(chunk, 3, 0, 0) == 0
    lib.nmod_trim_scratch(0)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    got = nm.detect_host(a, None, b, None, rid, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_KS, stride0=n0, stride1=n1)
    st = _stats()
    pos_per_chunk = chunk // ((n0 + n1) * 4)
    assert st.chunk_positions == pos_per_chunk and st.chunks == -(-npos // pos_per_chunk)
    # 3 slots x (8 MiB of rows + 86 B of workspace and 17 B of results per position) + 36 B per position of the batch
    assert st.device_bytes <= 3 * (chunk + 2 * 512 + pos_per_chunk * 128 + 4096) + 36 * npos + 8192
    assert st.device_bytes < (a.nbytes + b.nbytes) // 8
    held = free0 - torch.cuda.mem_get_info(0)[0]                        # what the library's pool still caches
    assert held <= st.device_bytes + (64 << 20)
    assert lib.nmod_trim_scratch(0) == 0
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info(0)[0] <= (8 << 20)           # ring, streams and pool are back
    ref = _device_reference(a, None, b, None, rid, nb=2, method='stouffer', tests=L.TEST_KS, stride0=n0, stride1=n1)
    _same(got, ref, list(got))


def test_large_positions_and_float64_redo_inside_chunks():
    """chunks whose device call has to wait for the device (large-position scratch, float64 redo list) keep the pipeline correct"""
    import nanomod_amd as nm
    import nanomod_oracle as orc
    L, lib = _lib()
    rng = np.random.default_rng(17)
    sizes0 = np.array([30, 5000, 40, 2500, 64, 9000, 12, 700, 3000, 20] * 6)
    sizes1 = np.array([35, 4000, 2100, 60, 64, 50, 8000, 650, 3100, 25] * 6)
    npos = len(sizes0)
    off0 = np.zeros(npos + 1, np.int64); off1 = np.zeros(npos + 1, np.int64)
    np.cumsum(sizes0, out=off0[1:]); np.cumsum(sizes1, out=off1[1:])
    a = rng.normal(0, 1, off0[-1]) + 1e-9 * rng.normal(0, 1, off0[-1])      # doubles that are neither float32-exact nor on the grid
    b = rng.normal(0, 1, off1[-1])
    a[off0[3]:off0[3] + 7] = a[off0[3]]                                      # exact float64 ties -> the 64-bit-key redo
    rid = np.zeros(npos, np.int32)
    for dtype in (np.float64, np.float32):
        x = a.astype(dtype); y = b.astype(dtype)
        exp = orc.detect_batch(x.astype(np.float64), off0, y.astype(np.float64), off1, rid, 2, 2.0, orc.METHOD_STOUFFER)
        for chunk_bytes in (64 << 10, 1 << 30):
            assert lib.nmod_host_pipeline_config(chunk_bytes, 3, 2, 0) == 0
            got = nm.detect_host(x, off0, y, off1, rid, nb=2, weights_dif=2.0, method='stouffer')
            H.compare_outputs(got, exp)
            assert not got['status'].any()


def test_host_entry_errors_leave_no_memory_behind():
    import torch
    import nanomod_amd as nm
    L, lib = _lib()
    a = np.zeros(70000, np.float32)
    lib.nmod_trim_scratch(0)
    free0 = torch.cuda.mem_get_info(0)[0]
    # (a group beyond NMOD_MAX_RANKED is no error since round 5: the position is flagged, the rest computed)
    r = nm.detect_host(a, np.array([0, 10, 70000]), a, np.array([0, 10, 70000]), np.zeros(2, np.int32))
    assert (r['status'][1] & L.STATUS_TOO_LARGE) and not (r['status'][0] & L.STATUS_TOO_LARGE)
    with pytest.raises(L.NanomodLibraryError, match='invalid'):
        nm.detect_host(a, np.array([0, 10, 5]), a, np.array([0, 10, 20]), np.zeros(2, np.int32))     # decreasing offsets
    prm = L.make_params(memspace=L.MEM_HOST, method=L.METHOD_STOUFFER, weights_dif=0.0)
    out = L.NmodOut()
    buf = np.zeros(8); out.comb_st = buf.ctypes.data; out.comb_p = buf.ctypes.data
    off = np.array([0, 5, 10], np.int64)
    rid = np.zeros(2, np.int32)
    rc = lib.nmod_detect_batch(C.byref(prm), 2, a.ctypes.data, off.ctypes.data, a.ctypes.data, off.ctypes.data, rid.ctypes.data, None, 0, C.byref(out))
    assert rc == -1                                                       # WeightsDif <= 0 with Stouffer: rejected before any copy
    lib.nmod_trim_scratch(0)
    assert free0 - torch.cuda.mem_get_info(0)[0] <= (4 << 20)
