"""ctypes wrapper of oracle/libnanomod_oracle.so (the C restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, 'libnanomod_oracle.so')
if not os.path.exists(_PATH):
    raise ImportError('oracle/libnanomod_oracle.so not built (make -C oracle)')
_lib = C.CDLL(_PATH)
_lib.nmod_oracle_detect.restype = C.c_int
_lib.nmod_oracle_detect.argtypes = [C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_double, C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int]
_lib.nmod_oracle_max_threads.restype = C.c_int
METHODS = {'ks': 0, 'stouffer': 1, 'fisher': 2}


def max_threads():
    return _lib.nmod_oracle_max_threads()


def detect_batch(sig0, off0, sig1, off1, run_id, nb=2, weights_dif=2.0, method='stouffer', tests=7, threads=0):
    sig0 = np.ascontiguousarray(sig0)
    sig1 = np.ascontiguousarray(sig1)
    assert sig0.dtype == sig1.dtype and sig0.dtype in (np.float32, np.int16)
    dtype = 0 if sig0.dtype == np.float32 else 1
    off0 = np.ascontiguousarray(off0, dtype=np.int64)
    off1 = np.ascontiguousarray(off1, dtype=np.int64)
    run_id = np.ascontiguousarray(run_id, dtype=np.int32)
    npos = len(off0) - 1
    m = METHODS[method] if isinstance(method, str) else method
    names = ['mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p']
    out = {k: np.full(npos, np.nan) for k in names}
    out['status'] = np.zeros(npos, np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib.nmod_oracle_detect(npos, dtype, p(sig0), p(off0), p(sig1), p(off1), p(run_id), nb, weights_dif, m, tests,
                                 *[p(out[k]) for k in names], p(out['status']), threads)
    assert rc == 0
    if m == 0:
        del out['comb_st'], out['comb_p']
    return out
