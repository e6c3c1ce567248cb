"""ctypes binding of libnanomod_hip.so (include/nanomod_hip.h).

The product path has no CPU fallback: if the HIP library is missing, loading
fails loudly with instructions to build it.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (NMOD_HIP_LIB: another build of the same library, for A/B measurements of kernel variants on one box)
LIB_PATH = os.environ.get('NMOD_HIP_LIB') or os.path.join(_HERE, 'libnanomod_hip.so')

NMOD_ABI_VERSION = 4
DTYPE_F32, DTYPE_I16_MILLI, DTYPE_F64 = 0, 1, 2
MEM_HOST, MEM_DEVICE = 0, 1
METHOD_KS, METHOD_STOUFFER, METHOD_FISHER = 0, 1, 2
TEST_KS, TEST_MWU, TEST_WELCH, TEST_ALL = 1, 2, 4, 7
FLAG_KS_RATIONAL_D, FLAG_CHECK_FINITE, FLAG_NO_COUNTING, FLAG_NO_COUNT_WIDE, FLAG_NO_HOST_NARROW = 1, 2, 4, 8, 16
STATUS_MWU_ALL_IDENTICAL, STATUS_T_NAN, STATUS_EMPTY, STATUS_TOO_LARGE, STATUS_NONFINITE = 1, 2, 4, 8, 16
KERNEL_RANK_STATS, KERNEL_FINALIZE, KERNEL_COMBINE, KERNEL_SYNTH = 0, 1, 2, 3
MAX_GROUP = 2048          # largest group of the wave-resident kernels; larger ones (<= MAX_RANKED) take big_rank_kernel
MAX_RANKED = 65535
MAX_NB = 64
COMM_ID_BYTES = 128
ERR_NO_RCCL, ERR_RCCL = -6, -7

METHOD_BY_NAME = {'ks': METHOD_KS, 'stouffer': METHOD_STOUFFER, 'fisher': METHOD_FISHER}

OUT_FIELDS = ('mwu_u', 'mwu_p', 't_t', 't_p', 'ks_d', 'ks_p', 'comb_st', 'comb_p',
              'mean0', 'std0', 'mean1', 'std1')


class NmodParams(C.Structure):
    _fields_ = [('struct_size', C.c_int32), ('device', C.c_int32), ('stream', C.c_void_p),
                ('memspace', C.c_int32), ('dtype', C.c_int32), ('tests', C.c_int32),
                ('method', C.c_int32), ('nb', C.c_int32), ('want_mstd', C.c_int32),
                ('weights_dif', C.c_double), ('stride0', C.c_int64), ('stride1', C.c_int64),
                ('max_n0', C.c_int32), ('max_n1', C.c_int32), ('timer', C.c_void_p), ('flags', C.c_int32), ('reserved', C.c_int32)]


class NmodOut(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in OUT_FIELDS] + [('status', C.c_void_p)]


class NmodHostStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('chunks', 'slots', 'copy_threads', 'pinned_input', 'chunk_positions',
                                         'device_bytes', 'pinned_bytes', 'h2d_bytes', 'd2h_bytes', 'narrowed_chunks')]


class NmodDispatchStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ('positions', 'ks_rank', 'rank_hist', 'rank_hist_wide', 'rank_pair', 'rank_count', 'rank_count_wide',
                                         'big', 'skipped', 'count_tried', 'count_rejected', 'f64_redo')] + [('reserved', C.c_int64 * 4)]


def last_dispatch_stats():
    """nmod_last_dispatch_stats as a dict: which K1 form computed the positions of this thread's last nmod_detect_batch"""
    st = NmodDispatchStats()
    check(load().nmod_last_dispatch_stats(C.byref(st)), 'nmod_last_dispatch_stats')
    return {n: int(getattr(st, n)) for n, _ in NmodDispatchStats._fields_ if n != 'reserved'}


class NanomodLibraryError(RuntimeError):
    pass


_lib = None

# every symbol include/nanomod_hip.h declares: (restype, argtypes)
_SIGNATURES = {
    'nmod_abi_version': (C.c_int, []),
    'nmod_device_count': (C.c_int, []),
    'nmod_strerror': (C.c_char_p, [C.c_int]),
    'nmod_workspace_bytes': (C.c_int64, [C.POINTER(NmodParams), C.c_int64]),
    'nmod_detect_batch': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(NmodOut)]),
    'nmod_combine_track': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p]),
    'nmod_synth_fill': (C.c_int, [C.POINTER(NmodParams), C.c_uint64, C.c_int64, C.c_int64, C.c_int32,
                                  C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    'nmod_synth_fill_csr': (C.c_int, [C.POINTER(NmodParams), C.c_uint64, C.c_int64, C.c_int64, C.c_int32,
                                      C.c_void_p, C.c_int64, C.c_float, C.c_void_p]),
    'nmod_synth_fill_events': (C.c_int, [C.POINTER(NmodParams), C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    'nmod_evtimer_create': (C.c_int, [C.c_int32, C.POINTER(C.c_void_p)]),
    'nmod_evtimer_reset': (C.c_int, [C.c_void_p]),
    'nmod_evtimer_read': (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    'nmod_evtimer_destroy': (C.c_int, [C.c_void_p]),
    'nmod_selftest': (C.c_int, [C.c_int32]),
    'nmod_comm_unique_id': (C.c_int, [C.c_void_p]),
    'nmod_comm_init_rank': (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    'nmod_allgather_tracks': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'nmod_comm_destroy': (C.c_int, [C.c_void_p]),
    'nmod_narrow_probe': (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    'nmod_format_probe': (C.c_int, [C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_char_p, C.c_int64]),
    'nmod_trim_scratch': (C.c_int, [C.c_int32]),
    'nmod_host_pipeline_config': (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    'nmod_last_host_stats': (C.c_int, [C.POINTER(NmodHostStats)]),
    'nmod_last_dispatch_stats': (C.c_int, [C.POINTER(NmodDispatchStats)]),
    'nmod_build_info': (C.c_char_p, []),
    'nmod_downsample_ks': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int32, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p]),
    'nmod_argsort_keys': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p]),
    'nmod_describe_dispatch': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_int64, C.c_char_p, C.c_int32]),
    'nmod_write_sign_test': (C.c_int, [C.c_char_p, C.c_int64, C.c_void_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_void_p,
                                       C.c_char_p] + [C.c_void_p] * 10 + [C.c_int32]),
    'nmod_rank_order': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    'nmod_region_rank': (C.c_int, [C.POINTER(NmodParams), C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_char, C.c_double, C.c_int32, C.c_void_p, C.POINTER(C.c_int64)]),
}


def load():
    """Load the HIP library (once).  Raises NanomodLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NanomodLibraryError(
            'libnanomod_hip.so not found at %s. Build it with `python -c "import __graft_entry__ as g; '
            'g.build()"` or `make -C nanomod_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and the package
    # uses torch for device memory and streams, so let torch load its copy first; the loader then
    # binds our DT_NEEDED libamdhip64.so.7 to that already-loaded object (same SONAME).  Loading in
    # the other order leaves two runtimes in the process and torch then sees no GPU.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI is incomplete
        fn.restype = res
        fn.argtypes = args
    if lib.nmod_abi_version() != NMOD_ABI_VERSION:
        raise NanomodLibraryError('ABI version mismatch: library %d, binding %d'
                                  % (lib.nmod_abi_version(), NMOD_ABI_VERSION))
    _lib = lib
    return lib


def check(rc, what='nanomod_hip'):
    if rc != 0:
        msg = load().nmod_strerror(rc).decode()
        raise NanomodLibraryError('%s failed: %s (code %d)' % (what, msg, rc))


def make_params(device=0, stream=0, memspace=MEM_HOST, dtype=DTYPE_F32, tests=TEST_ALL,
                method=METHOD_STOUFFER, nb=2, weights_dif=2.0, want_mstd=0,
                stride0=0, stride1=0, max_n0=0, max_n1=0, timer=None, flags=0):
    p = NmodParams()
    p.struct_size = C.sizeof(NmodParams)
    p.device = device
    p.stream = stream or None
    p.memspace = memspace
    p.dtype = dtype
    p.tests = tests
    p.method = method
    p.nb = nb
    p.want_mstd = want_mstd
    p.weights_dif = weights_dif
    p.stride0 = stride0
    p.stride1 = stride1
    p.max_n0 = max_n0
    p.max_n1 = max_n1
    p.timer = timer
    p.flags = flags
    p.reserved = 0
    return p
