// nmod_rank_order — the 3-key ranking of the result table (myDetect.py:447-462) on the device.
// The reference sorts the records by the tuple (combined, KS, MWU) p-value (or statistic) with Python's stable
// sorted(); here: three stable LSD passes of rocPRIM's radix sort over order-preserving 64-bit images of the fp64
// keys (least significant key first), carrying the record index.  Outside the timed hot path;
// the sort itself is the ROCm library's, like a plain GEMM would be hipBLASLt's.
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <rocprim/device/device_radix_sort.hpp>

#include "../../include/nanomod_hip.h"

namespace nmod {

// fp64 -> uint64 with the same order; -0.0 ties with +0.0 (as in Python), NaN sorts last
__global__ __launch_bounds__(256) void rank_keys_kernel(const double* key, const uint32_t* idx, int64_t n, uint64_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    double v = key[idx[i]];
    if (v == 0.0) v = 0.0;
    uint64_t b = (uint64_t)__double_as_longlong(v);
    b = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    out[i] = (v != v) ? ~0ull : b;
  }
}
__global__ __launch_bounds__(256) void rank_iota_kernel(uint32_t* idx, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) idx[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void rank_emit_kernel(const uint32_t* idx, int64_t n, int descending, int32_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = (int32_t)idx[descending ? n - 1 - i : i];
}

struct Buf {
  void* p = nullptr;
  ~Buf() { if (p) hipFree(p); }
  hipError_t alloc(size_t b) { return hipMalloc(&p, b ? b : 1); }
};

#define NMOD_RO_HIP(call) do { if ((call) != hipSuccess) return NMOD_ERR_HIP; } while (0)

}  // namespace nmod

using namespace nmod;

extern "C" int nmod_rank_order(const nmod_params* prm, int64_t npos, const double* key_primary, const double* key_second,
                               const double* key_third, int32_t descending, int32_t* order_out) {
  if (!prm || prm->struct_size != (int32_t)sizeof(nmod_params) || npos < 0 || npos > INT32_MAX) return NMOD_ERR_INVALID_ARG;
  if (npos == 0) return NMOD_OK;
  if (!key_primary || !key_second || !key_third || !order_out) return NMOD_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || prm->device < 0 || prm->device >= ndev) return NMOD_ERR_NO_DEVICE;
  NMOD_RO_HIP(hipSetDevice(prm->device));
  hipStream_t stream = (hipStream_t)prm->stream;
  const bool host = prm->memspace == NMOD_MEM_HOST;
  const size_t n = (size_t)npos;
  Buf dk[3], ka, kb, ia, ib, tmp, dout;
  const double* keys[3] = {key_third, key_second, key_primary};          // least significant first
  if (host) {
    for (int k = 0; k < 3; ++k) {
      NMOD_RO_HIP(dk[k].alloc(n * 8));
      NMOD_RO_HIP(hipMemcpyAsync(dk[k].p, keys[k], n * 8, hipMemcpyHostToDevice, stream));
      keys[k] = (const double*)dk[k].p;
    }
    NMOD_RO_HIP(dout.alloc(n * 4));
  }
  NMOD_RO_HIP(ka.alloc(n * 8)); NMOD_RO_HIP(kb.alloc(n * 8)); NMOD_RO_HIP(ia.alloc(n * 4)); NMOD_RO_HIP(ib.alloc(n * 4));
  size_t tmp_bytes = 0;
  NMOD_RO_HIP(rocprim::radix_sort_pairs(nullptr, tmp_bytes, (const uint64_t*)ka.p, (uint64_t*)kb.p, (const uint32_t*)ia.p,
                                        (uint32_t*)ib.p, n, 0, 64, stream));
  NMOD_RO_HIP(tmp.alloc(tmp_bytes));
  const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(rank_iota_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t*)ia.p, (int64_t)n);
  uint32_t* cur = (uint32_t*)ia.p; uint32_t* nxt = (uint32_t*)ib.p;
  for (int k = 0; k < 3; ++k) {
    hipLaunchKernelGGL(rank_keys_kernel, dim3(blocks), dim3(256), 0, stream, keys[k], cur, (int64_t)n, (uint64_t*)ka.p);
    NMOD_RO_HIP(rocprim::radix_sort_pairs(tmp.p, tmp_bytes, (const uint64_t*)ka.p, (uint64_t*)kb.p, (const uint32_t*)cur, nxt,
                                          n, 0, 64, stream));
    uint32_t* t = cur; cur = nxt; nxt = t;
  }
  int32_t* dst = host ? (int32_t*)dout.p : order_out;
  hipLaunchKernelGGL(rank_emit_kernel, dim3(blocks), dim3(256), 0, stream, cur, (int64_t)n, (int)(descending != 0), dst);
  NMOD_RO_HIP(hipGetLastError());
  if (host) NMOD_RO_HIP(hipMemcpyAsync(order_out, dst, n * 4, hipMemcpyDeviceToHost, stream));
  NMOD_RO_HIP(hipStreamSynchronize(stream));          // the temporaries are freed on return
  return NMOD_OK;
}
