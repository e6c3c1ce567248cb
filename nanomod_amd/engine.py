"""Batch entry points over the C ABI.

`detect_host`   — numpy arrays in host memory (the drop-in `mtest2` path uses it).
`detect_device` — torch tensors already resident in HBM (bench / sharded path);
                  torch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _check_csr(off, npos, name):
    if off is None:
        return
    if off.dtype != np.int64 or off.ndim != 1 or off.shape[0] != npos + 1:
        raise ValueError('%s must be int64[npos+1]' % name)


_warm = {}


def warm_up(device=0):
    """Start the HIP runtime, the device context and the code object of the library on a background thread (about a
    second in a fresh process) while the caller prepares its inputs on the host; detect_host joins it.  Idempotent."""
    import threading
    if device in _warm:
        return _warm[device]

    def run():
        try:
            L.load().nmod_selftest(int(device))          # a 64-thread kernel: forces context creation and module load
        except Exception:                                # (errors surface in the real call, with their message)
            pass
    t = threading.Thread(target=run, name='nanomod-warm-up', daemon=True)
    if not _warm:
        # a caller that fails before detect_host joins the thread (build_csr raising, say) must not let the interpreter
        # shut down while the HIP runtime is still initialising on it
        import atexit
        atexit.register(lambda: [_join_warm_up(d) for d in list(_warm)])
    _warm[device] = t
    t.start()
    return t


def _join_warm_up(device):
    t = _warm.get(device)
    if t is not None and t.is_alive():
        t.join()


def detect_host(sig0, off0, sig1, off1, run_id, *, nb=2, weights_dif=2.0, method='stouffer',
                tests=L.TEST_ALL, want_mstd=False, device=0, stride0=0, stride1=0, flags=0, out=None):
    """Run the hot path on host-resident CSR inputs; returns a dict of numpy arrays.

    sig0/sig1: float32 (canonical), int16 (milli-units) or float64 1-D arrays; off0/off1:
    int64[npos+1] (or None with a fixed stride); run_id: int32[npos]; flags: L.FLAG_* (include/nanomod_hip.h).
    out: the dict a previous call of the same shape returned — its arrays are written again instead of allocating
    (and first-touching) new ones: at 4.6 M positions the page faults of fresh result arrays cost as much as 15 % of
    the PCIe-bound call."""
    lib = L.load()
    _join_warm_up(device)
    sig0 = np.ascontiguousarray(sig0)
    sig1 = np.ascontiguousarray(sig1)
    if sig0.dtype != sig1.dtype or sig0.dtype not in (np.float32, np.int16, np.float64):
        raise ValueError('sig0/sig1 must both be float32, both int16 (milli-units) or both float64')
    # float64 (what the reference holds): the library re-encodes it on the device, NMOD_DTYPE_F64
    dtype = {np.dtype(np.float32): L.DTYPE_F32, np.dtype(np.int16): L.DTYPE_I16_MILLI, np.dtype(np.float64): L.DTYPE_F64}[sig0.dtype]
    if off0 is not None:
        npos = len(off0) - 1
    elif off1 is not None:
        npos = len(off1) - 1
    else:
        npos = sig0.shape[0] // stride0
    off0 = None if off0 is None else np.ascontiguousarray(off0, dtype=np.int64)
    off1 = None if off1 is None else np.ascontiguousarray(off1, dtype=np.int64)
    _check_csr(off0, npos, 'off0')
    _check_csr(off1, npos, 'off1')
    method_id = L.METHOD_BY_NAME[method] if isinstance(method, str) else method
    run = None
    if run_id is not None:
        run = np.ascontiguousarray(run_id, dtype=np.int32)
        if run.shape[0] != npos:
            raise ValueError('run_id must be int32[npos]')
    prm = L.make_params(device=device, memspace=L.MEM_HOST, dtype=dtype, tests=tests, method=method_id,
                        nb=nb, weights_dif=weights_dif, want_mstd=int(bool(want_mstd)),
                        stride0=stride0 if off0 is None else 0, stride1=stride1 if off1 is None else 0, flags=flags)
    res = {}
    wanted = []
    if tests & L.TEST_MWU:
        wanted += ['mwu_u', 'mwu_p']
    if tests & L.TEST_WELCH:
        wanted += ['t_t', 't_p']
    if (tests & L.TEST_KS) or method_id != L.METHOD_KS:
        wanted += ['ks_d', 'ks_p']
    if method_id != L.METHOD_KS:
        wanted += ['comb_st', 'comb_p']
    if want_mstd:
        wanted += ['mean0', 'std0', 'mean1', 'std1']
    reuse = out
    out = L.NmodOut()
    for name in wanted:
        if reuse is not None:
            a = reuse[name]
            if a.dtype != np.float64 or a.shape != (npos,) or not a.flags.c_contiguous:
                raise ValueError('out[%r] must be a contiguous float64[npos] array' % name)
            res[name] = a
        else:
            res[name] = np.empty(npos, dtype=np.float64)      # (every element is written: finalize_kernel stores all positions)
        setattr(out, name, _np_ptr(res[name]))
    if reuse is not None:
        res['status'] = reuse['status']
        if res['status'].dtype != np.uint8 or res['status'].shape != (npos,):
            raise ValueError("out['status'] must be a uint8[npos] array")
    else:
        res['status'] = np.empty(npos, dtype=np.uint8)
    out.status = _np_ptr(res['status'])
    rc = lib.nmod_detect_batch(C.byref(prm), npos, _np_ptr(sig0), _np_ptr(off0), _np_ptr(sig1), _np_ptr(off1),
                               _np_ptr(run), None, 0, C.byref(out))
    L.check(rc, 'nmod_detect_batch')
    return res


def combine_host(ks_d, ks_p, run_id, *, nb=2, weights_dif=2.0, method='stouffer', device=0):
    lib = L.load()
    ks_p = np.ascontiguousarray(ks_p, dtype=np.float64)
    ks_d = np.ascontiguousarray(ks_d, dtype=np.float64)
    run = np.ascontiguousarray(run_id, dtype=np.int32)
    npos = ks_p.shape[0]
    method_id = L.METHOD_BY_NAME[method] if isinstance(method, str) else method
    prm = L.make_params(device=device, memspace=L.MEM_HOST, method=method_id, nb=nb, weights_dif=weights_dif)
    st = np.empty(npos, dtype=np.float64)
    pv = np.empty(npos, dtype=np.float64)
    rc = lib.nmod_combine_track(C.byref(prm), npos, _np_ptr(ks_d), _np_ptr(ks_p), _np_ptr(run), _np_ptr(st), _np_ptr(pv))
    L.check(rc, 'nmod_combine_track')
    return st, pv


def rank_order_host(key_primary, key_second, key_third, descending=False, device=0):
    """myDetect.py:447-462 on the device: the order of Python's stable sorted() by the tuple
    (key_primary, key_second, key_third), reversed as a whole when `descending` (rankUse == 'st')."""
    lib = L.load()
    ks = [np.ascontiguousarray(k, dtype=np.float64) for k in (key_primary, key_second, key_third)]
    n = ks[0].shape[0]
    order = np.empty(n, dtype=np.int32)
    prm = L.make_params(device=device, memspace=L.MEM_HOST)
    rc = lib.nmod_rank_order(C.byref(prm), n, _np_ptr(ks[0]), _np_ptr(ks[1]), _np_ptr(ks[2]), 1 if descending else 0, _np_ptr(order))
    L.check(rc, 'nmod_rank_order')
    return order


def argsort_device(key):
    """stable ascending order of an int64 CUDA tensor (nmod_argsort_keys: the library's radix sort), as an int64 CUDA tensor"""
    import torch
    assert key.is_cuda and key.dtype == torch.int64 and key.is_contiguous()
    n = key.numel()
    order = torch.empty(n, dtype=torch.int32, device=key.device)
    prm = L.make_params(device=key.device.index or 0, memspace=L.MEM_DEVICE, stream=torch.cuda.current_stream(key.device).cuda_stream)
    L.check(L.load().nmod_argsort_keys(C.byref(prm), n, key.data_ptr(), order.data_ptr()), 'nmod_argsort_keys')
    return order.to(torch.int64)


def region_rank_host(strand_lo, strand_hi, pos, base, value, w, movesize, na, percentile, wind_ovlp, device=0):
    """myDetect.py:463-515 on array-shaped records (see nmod_region_rank): indices of the ranked window centres."""
    lib = L.load()
    n = len(pos)
    lo = np.ascontiguousarray(strand_lo, dtype=np.int32); hi = np.ascontiguousarray(strand_hi, dtype=np.int32)
    pos = np.ascontiguousarray(pos, dtype=np.int64); value = np.ascontiguousarray(value, dtype=np.float64)
    base = bytes(base)
    assert len(base) == n
    out = np.empty(n, dtype=np.int32)
    cnt = C.c_int64(0)
    prm = L.make_params(device=device, memspace=L.MEM_HOST)
    rc = lib.nmod_region_rank(C.byref(prm), n, _np_ptr(lo), _np_ptr(hi), _np_ptr(pos), base, _np_ptr(value), int(w), int(movesize),
                              (na or '\0').encode()[:1], float(percentile), int(wind_ovlp), _np_ptr(out), C.byref(cnt))
    L.check(rc, 'nmod_region_rank')
    return out[:cnt.value]


def _first_chars(a):
    """the first character of every element (a blank for an empty one) as one bytes object, without a Python loop"""
    u = np.asarray(a)
    if u.dtype.kind != 'U':
        u = u.astype('U1') if u.dtype.kind in 'SO' or u.size == 0 else np.array([str(x) for x in u.tolist()], dtype='U1')
    elif u.dtype.itemsize != 4:
        u = u.astype('U1')
    codes = np.ascontiguousarray(u).view(np.uint32)
    if codes.size and int(codes.max()) > 127:                      # non-ASCII: the slow, general way
        return ''.join(str(x)[:1] or ' ' for x in np.asarray(a).tolist()).encode()
    c = codes.astype(np.uint8)
    c[c == 0] = 32
    return c.tobytes()


def write_sign_test_host(path, meta, res, with_comb):
    """save_test's table (myDetect.py:522-538) through nmod_write_sign_test.  meta: arrays chrom_id (int32), pos
    (int64, 0-based), strand / base (one character each), n0 / n1 (int32) + the list `names` that chrom_id indexes."""
    lib = L.load()
    npos = len(meta['pos'])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    cid = np.ascontiguousarray(meta['chrom_id'], dtype=np.int32)
    names = b''.join(str(n).encode() + b'\0' for n in meta['names'])
    strand = _first_chars(meta['strand'])
    base = _first_chars(meta['base'])
    pos = np.ascontiguousarray(meta['pos'], dtype=np.int64)
    n0 = np.ascontiguousarray(meta['n0'], dtype=np.int32); n1 = np.ascontiguousarray(meta['n1'], dtype=np.int32)
    cols = [np.ascontiguousarray(res[k], dtype=np.float64) for k in ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p')]
    comb = [np.ascontiguousarray(res[k], dtype=np.float64) for k in ('comb_st', 'comb_p')] if with_comb else [None, None]
    rc = lib.nmod_write_sign_test(str(path).encode(), npos, p(cid), names, len(meta['names']), strand, p(pos), base,
                                  p(n0), p(n1), *[p(c) for c in cols],
                                  *(p(c) if c is not None else None for c in comb), 1 if with_comb else 0)
    L.check(rc, 'nmod_write_sign_test')


class EventTimer:
    """HIP-event timer handle (nmod_evtimer_*): per-kernel elapsed ms measured on the launch stream."""

    def __init__(self, capacity=4096):
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.nmod_evtimer_create(capacity, C.byref(self._h)), 'nmod_evtimer_create')

    @property
    def handle(self):
        return self._h

    def reset(self):
        L.check(self._lib.nmod_evtimer_reset(self._h), 'nmod_evtimer_reset')

    def read(self, kernel):
        ms = C.c_double()
        n = C.c_int32()
        L.check(self._lib.nmod_evtimer_read(self._h, kernel, C.byref(ms), C.byref(n)), 'nmod_evtimer_read')
        return ms.value, n.value

    def close(self):
        if self._h:
            self._lib.nmod_evtimer_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceDetector:
    """Device-resident form: inputs and outputs are torch CUDA tensors; nothing is copied
    and nothing synchronises unless max_n0/max_n1 are unknown for CSR inputs or allow groups beyond MAX_GROUP
    (one round trip sizes the scratch of the large-position pass)."""

    def __init__(self, device=0, nb=2, weights_dif=2.0, method='stouffer', tests=L.TEST_ALL, want_mstd=False, flags=0):
        import torch
        self.torch = torch
        self.lib = L.load()
        self.device = device
        self.nb = nb
        self.weights_dif = weights_dif
        self.method = L.METHOD_BY_NAME[method] if isinstance(method, str) else method
        self.tests = tests
        self.want_mstd = bool(want_mstd)
        self.flags = int(flags)          # L.FLAG_* (include/nanomod_hip.h: NMOD_FLAG_*)
        self._ws = None
        self.timer = None

    def _params(self, dtype, stride0, stride1, max_n0, max_n1):
        stream = self.torch.cuda.current_stream(self.device).cuda_stream
        return L.make_params(device=self.device, stream=stream, memspace=L.MEM_DEVICE, dtype=dtype,
                             tests=self.tests, method=self.method, nb=self.nb, weights_dif=self.weights_dif,
                             want_mstd=int(self.want_mstd), stride0=stride0, stride1=stride1,
                             max_n0=max_n0, max_n1=max_n1,
                             timer=self.timer.handle if self.timer is not None else None, flags=self.flags)

    def _dtype_of(self, t):
        torch = self.torch
        if t.dtype == torch.float32:
            return L.DTYPE_F32
        if t.dtype == torch.int16:
            return L.DTYPE_I16_MILLI
        if t.dtype == torch.float64:
            return L.DTYPE_F64
        raise ValueError('signals must be float32, int16 (milli-units) or float64')

    def workspace(self, prm, npos):
        need = self.lib.nmod_workspace_bytes(C.byref(prm), npos)
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.torch.empty(need, dtype=self.torch.uint8, device='cuda:%d' % self.device)
        return self._ws, need

    def alloc_outputs(self, npos):
        torch = self.torch
        dev = 'cuda:%d' % self.device
        names = []
        if self.tests & L.TEST_MWU:
            names += ['mwu_u', 'mwu_p']
        if self.tests & L.TEST_WELCH:
            names += ['t_t', 't_p']
        if (self.tests & L.TEST_KS) or self.method != L.METHOD_KS:
            names += ['ks_d', 'ks_p']
        if self.method != L.METHOD_KS:
            names += ['comb_st', 'comb_p']
        if self.want_mstd:
            names += ['mean0', 'std0', 'mean1', 'std1']
        res = {n: torch.empty(npos, dtype=torch.float64, device=dev) for n in names}
        res['status'] = torch.empty(npos, dtype=torch.uint8, device=dev)
        return res

    def dispatch_stats(self):
        """which K1 form computed the positions of the last run() (nmod_last_dispatch_stats; synchronises the stream)"""
        return L.last_dispatch_stats()

    def run(self, sig0, sig1, run_id, *, off0=None, off1=None, stride0=0, stride1=0, npos=None,
            max_n0=0, max_n1=0, out=None):
        """Enqueue the hot path on the current stream.  Either CSR offsets (int64 CUDA tensors)
        or fixed strides describe the rows.  Returns the dict of output tensors."""
        dtype = self._dtype_of(sig0)
        if self._dtype_of(sig1) != dtype:
            raise ValueError('sig0 and sig1 must share a dtype')
        if npos is None:
            npos = (off0.numel() - 1) if off0 is not None else sig0.numel() // stride0
        prm = self._params(dtype, stride0 if off0 is None else 0, stride1 if off1 is None else 0, max_n0, max_n1)
        ws, need = self.workspace(prm, npos)
        res = out if out is not None else self.alloc_outputs(npos)
        o = L.NmodOut()
        for name in L.OUT_FIELDS:
            if name in res:
                setattr(o, name, res[name].data_ptr())
        o.status = res['status'].data_ptr()
        ptr = lambda t: (t.data_ptr() if t is not None else None)
        rc = self.lib.nmod_detect_batch(C.byref(prm), npos, ptr(sig0), ptr(off0), ptr(sig1), ptr(off1),
                                        ptr(run_id), ws.data_ptr(), need, C.byref(o))
        L.check(rc, 'nmod_detect_batch')
        return res

    def synth_fill(self, out, seed, pos_begin, npos, group, n_per_pos, plant_period=0, plant_shift=0.0):
        prm = self._params(self._dtype_of(out), 0, 0, 0, 0)
        rc = self.lib.nmod_synth_fill(C.byref(prm), seed, pos_begin, npos, group, n_per_pos,
                                      plant_period, plant_shift, out.data_ptr())
        L.check(rc, 'nmod_synth_fill')
        return out


    def synth_fill_csr(self, out, seed, pos_begin, off, group, plant_period=0, plant_shift=0.0):
        """ragged rows: `off` = int64 CUDA tensor of npos + 1 element offsets into `out`"""
        prm = self._params(self._dtype_of(out), 0, 0, 0, 0)
        rc = self.lib.nmod_synth_fill_csr(C.byref(prm), seed, pos_begin, off.numel() - 1, group, off.data_ptr(),
                                          plant_period, plant_shift, out.data_ptr())
        L.check(rc, 'nmod_synth_fill_csr')
        return out

    def synth_fill_events(self, out, seed, pos_begin, npos, group, n_per_pos=0, off=None, plant_period=0, plant_shift_milli=0,
                          spread_milli=200, outlier_permille=0):
        """event-like rows (nmod_synth_fill_events): a level per position, reads spread `spread_milli` around it, on the
        3-decimal grid; fixed stride (n_per_pos > 0) or ragged (`off` = int64 CUDA tensor of npos + 1 offsets); `outlier_permille`
        of 1 000 reads are replaced by a uniform draw over +-5 units (mis-segmented events)"""
        prm = self._params(self._dtype_of(out), 0, 0, 0, 0)
        rc = self.lib.nmod_synth_fill_events(C.byref(prm), seed, pos_begin, npos, group, n_per_pos,
                                             off.data_ptr() if off is not None else None, plant_period,
                                             int(plant_shift_milli), int(spread_milli), int(outlier_permille), out.data_ptr())
        L.check(rc, 'nmod_synth_fill_events')
        return out


def downsample_ks(sig0, off0, sig1, off1, positions, cov, *, iters=100, quantile=0.25, seed=0, device=0):
    """The down-sampling branch of getKStest (myDetect.py:345-361) for the positions `positions` (indices into
    the CSR arrays) through nmod_downsample_ks: `iters` times, a group with more than cov[i] samples is resampled WITH
    replacement to cov[i] samples (np.random.choice semantics), KS is run on each resample, and the (D, p) pair at index
    int(iters * quantile) of the p-sorted resamples is reported.  The reference draws from an unseeded
    global RNG, so its numbers are not reproducible; here the draws come from a seeded counter-based device generator —
    statistically equivalent, not bit-comparable (SURVEY.md §8a row A3', §8f row 4).  The resampled rows are
    materialised chunk by chunk in HBM and go through the same KS kernel as everything else; resampling, KS and the
    quantile selection all run in the library (round 3 assembled the rows with torch indexing).
    Returns (ks_d, ks_p) numpy arrays aligned with `positions`."""
    lib = L.load()
    _join_warm_up(device)
    sig0 = np.ascontiguousarray(sig0); sig1 = np.ascontiguousarray(sig1)
    if sig0.dtype != sig1.dtype or sig0.dtype not in (np.float32, np.int16, np.float64):
        raise ValueError('sig0/sig1 must both be float32, both int16 (milli-units) or both float64')
    dtype = {np.dtype(np.float32): L.DTYPE_F32, np.dtype(np.int16): L.DTYPE_I16_MILLI, np.dtype(np.float64): L.DTYPE_F64}[sig0.dtype]
    positions = np.ascontiguousarray(positions, dtype=np.int64)
    cov = np.ascontiguousarray(cov, dtype=np.int64)
    off0 = np.ascontiguousarray(off0, dtype=np.int64); off1 = np.ascontiguousarray(off1, dtype=np.int64)
    if positions.shape != cov.shape or (len(positions) and (positions.min() < 0 or positions.max() >= len(off0) - 1)):
        raise ValueError('positions / cov must be aligned and index the CSR rows')
    out_d = np.empty(len(positions)); out_p = np.empty(len(positions))
    prm = L.make_params(device=device, memspace=L.MEM_HOST, dtype=dtype, tests=L.TEST_KS, method=L.METHOD_KS)
    rc = lib.nmod_downsample_ks(C.byref(prm), len(positions), _np_ptr(sig0), _np_ptr(off0), _np_ptr(sig1), _np_ptr(off1),
                                _np_ptr(positions), _np_ptr(cov), int(iters), float(quantile), int(seed) & 0xFFFFFFFFFFFFFFFF,
                                _np_ptr(out_d), _np_ptr(out_p))
    L.check(rc, 'nmod_downsample_ks')
    return out_d, out_p
