// Rankings of the result records on the device: nmod_rank_order and nmod_region_rank.
//
// nmod_rank_order — the 3-key ranking of the result table (myDetect.py:447-462) on the device.
// The reference sorts the records by the tuple (combined, KS, MWU) p-value (or statistic) with Python's stable
// sorted(); here: three stable LSD radix sorts (radix_sort.hpp: hand-written, eight one-byte passes each) over
// order-preserving 64-bit images of the fp64 keys (least significant key first), carrying the record index.
// Outside the timed hot path.
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>
#include <vector>

#include "../../include/nanomod_hip.h"
#include "radix_sort.hpp"

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
  NMOD_RO_HIP(tmp.alloc(rs_scratch_bytes((int64_t)n)));
  const unsigned blocks = (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
  hipLaunchKernelGGL(rank_iota_kernel, dim3(blocks), dim3(256), 0, stream, (uint32_t*)ia.p, (int64_t)n);
  uint32_t* cur = (uint32_t*)ia.p;
  for (int k = 0; k < 3; ++k) {
    hipLaunchKernelGGL(rank_keys_kernel, dim3(blocks), dim3(256), 0, stream, keys[k], cur, (int64_t)n, (uint64_t*)ka.p);
    NMOD_RO_HIP(rs_sort_pairs((uint64_t*)ka.p, cur, (uint64_t*)kb.p, (uint32_t*)ib.p, (int64_t)n, tmp.p, stream));   // (in place: eight passes)
  }
  int32_t* dst = host ? (int32_t*)dout.p : order_out;
  hipLaunchKernelGGL(rank_emit_kernel, dim3(blocks), dim3(256), 0, stream, cur, (int64_t)n, (int)(descending != 0), dst);
  NMOD_RO_HIP(hipGetLastError());
  if (host) NMOD_RO_HIP(hipMemcpyAsync(order_out, dst, n * 4, hipMemcpyDeviceToHost, stream));
  NMOD_RO_HIP(hipStreamSynchronize(stream));          // the temporaries are freed on return
  return NMOD_OK;
}

// nmod_argsort_keys — the stable ascending order of signed 64-bit keys (one radix sort of their order-preserving images): the
// grouping of a read pool's events by (chrom, strand, position) key in nanomod_amd/simulate.py (getGenomeEvents,
// mySimulat2.py:127-171), which used torch.argsort until round 5.
namespace nmod {
__global__ __launch_bounds__(256) void argsort_keys_kernel(const int64_t* key, int64_t n, uint64_t* out, uint32_t* idx) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    out[i] = (uint64_t)key[i] ^ 0x8000000000000000ull;
    idx[i] = (uint32_t)i;
  }
}
}  // namespace nmod

extern "C" int nmod_argsort_keys(const nmod_params* prm, int64_t n, const int64_t* keys, int32_t* order_out) {
  if (!prm || prm->struct_size != (int32_t)sizeof(nmod_params) || n < 0 || n > INT32_MAX) return NMOD_ERR_INVALID_ARG;
  if (n == 0) return NMOD_OK;
  if (!keys || !order_out) return NMOD_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || prm->device < 0 || prm->device >= ndev) return NMOD_ERR_NO_DEVICE;
  NMOD_RO_HIP(hipSetDevice(prm->device));
  hipStream_t stream = (hipStream_t)prm->stream;
  const bool host = prm->memspace == NMOD_MEM_HOST;
  const size_t cnt = (size_t)n;
  Buf dk, ka, kb, ia, ib, tmp;
  const int64_t* src = keys;
  if (host) {
    NMOD_RO_HIP(dk.alloc(cnt * 8));
    NMOD_RO_HIP(hipMemcpyAsync(dk.p, keys, cnt * 8, hipMemcpyHostToDevice, stream));
    src = (const int64_t*)dk.p;
  }
  NMOD_RO_HIP(ka.alloc(cnt * 8)); NMOD_RO_HIP(kb.alloc(cnt * 8)); NMOD_RO_HIP(ia.alloc(cnt * 4)); NMOD_RO_HIP(ib.alloc(cnt * 4));
  NMOD_RO_HIP(tmp.alloc(rs_scratch_bytes(n)));
  const unsigned blocks = (unsigned)((cnt + 255) / 256 < 8192 ? (cnt + 255) / 256 : 8192);
  hipLaunchKernelGGL(argsort_keys_kernel, dim3(blocks), dim3(256), 0, stream, src, n, (uint64_t*)ka.p, (uint32_t*)ia.p);
  NMOD_RO_HIP(rs_sort_pairs((uint64_t*)ka.p, (uint32_t*)ia.p, (uint64_t*)kb.p, (uint32_t*)ib.p, n, tmp.p, stream));
  NMOD_RO_HIP(hipMemcpyAsync(order_out, ia.p, cnt * 4, host ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice, stream));
  NMOD_RO_HIP(hipStreamSynchronize(stream));          // the temporaries are freed on return
  return NMOD_OK;
}

// ---------------------------------------------------------------------------------------------------------
// nmod_region_rank — window ranking of --RegionRankbyST 1 (myDetect.py:463-515).
// A window is centred on a tested position pk = pmin + t * movesize of one (chrom, strand), needs all of
// pk-w .. pk+w tested and below the strand's last position, and is keyed by the `percentile`-th smallest value
// of its (base-filtered) members, ties broken by |w - index of the window minimum|.  One thread per position
// computes the key of the window centred there (selection by counting: windows are a few dozen values); the
// windows are ranked with the same device radix sort as nmod_rank_order (radix_sort.hpp); the overlap suppression of WindOvlp == 1
// (a window is dropped when a better one on the same strand lies closer than w) is a sequential greedy pass on
// the host over the ranked list, O(w) per window with a per-position flag instead of the reference's scan of
// everything kept so far.
namespace nmod {

struct RegionArgs {
  int64_t npos; const int32_t* s_lo; const int32_t* s_hi; const int64_t* pos; const char* base; const double* value;
  int32_t w, movesize; char na; double pct;
  double* key; double* tb; uint8_t* valid;
};

__global__ __launch_bounds__(256) void region_keys_kernel(RegionArgs a) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= a.npos) return;
  const int w = a.w;
  const int64_t lo = a.s_lo[i], hi = a.s_hi[i];
  const int64_t pmin = a.pos[lo], pmax = a.pos[hi];
  const int64_t p = a.pos[i];
  bool ok = ((p - pmin) % a.movesize) == 0 && p < pmax && i - w >= lo && i + w <= hi;
  if (ok) ok = a.pos[i + w] - a.pos[i - w] == 2 * (int64_t)w && a.pos[i + w] < pmax && a.pos[i - w] >= 0;
  int cnt = 0;
  if (ok) {
    for (int l = -w; l <= w; ++l) cnt += (!a.na || a.base[i + l] == a.na) ? 1 : 0;
    ok = cnt > 5;
  }
  a.valid[i] = ok ? 1 : 0;
  if (!ok) { a.key[i] = 0.0; a.tb[i] = 0.0; return; }
  const int k = (int)(a.pct * (double)(cnt - 1) + 0.5);
  double kth = 0.0, vmin = 0.0;
  int idx_min = 0, f = 0;
  for (int l = -w; l <= w; ++l) {
    if (a.na && a.base[i + l] != a.na) continue;
    const double v = a.value[i + l];
    int rank = 0, g = 0;
    for (int m = -w; m <= w; ++m) {
      if (a.na && a.base[i + m] != a.na) continue;
      const double u = a.value[i + m];
      rank += (u < v || (u == v && g < f)) ? 1 : 0;
      ++g;
    }
    if (rank == k) kth = v;
    if (f == 0 || v < vmin) { vmin = v; idx_min = f; }
    ++f;
  }
  a.key[i] = kth;
  const int d = w - idx_min;
  a.tb[i] = (double)(d < 0 ? -d : d);
}

}  // namespace nmod

extern "C" int nmod_region_rank(const nmod_params* prm, int64_t npos, const int32_t* strand_lo, const int32_t* strand_hi,
                                const int64_t* pos, const char* base, const double* value, int32_t w, int32_t movesize,
                                char na, double percentile, int32_t wind_ovlp, int32_t* ranked_out, int64_t* n_ranked) {
  if (!prm || prm->struct_size != (int32_t)sizeof(nmod_params) || npos < 0 || npos > INT32_MAX || !n_ranked) return NMOD_ERR_INVALID_ARG;
  *n_ranked = 0;
  if (npos == 0) return NMOD_OK;
  if (!strand_lo || !strand_hi || !pos || !base || !value || !ranked_out || w < 0 || movesize < 1) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_HOST) return NMOD_ERR_INVALID_ARG;              // post-processing of host-side records
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || prm->device < 0 || prm->device >= ndev) return NMOD_ERR_NO_DEVICE;
  NMOD_RO_HIP(hipSetDevice(prm->device));
  hipStream_t stream = (hipStream_t)prm->stream;
  const size_t n = (size_t)npos;
  Buf d_lo, d_hi, d_pos, d_base, d_val, d_key, d_tb, d_ok;
  NMOD_RO_HIP(d_lo.alloc(n * 4)); NMOD_RO_HIP(d_hi.alloc(n * 4)); NMOD_RO_HIP(d_pos.alloc(n * 8)); NMOD_RO_HIP(d_base.alloc(n));
  NMOD_RO_HIP(d_val.alloc(n * 8)); NMOD_RO_HIP(d_key.alloc(n * 8)); NMOD_RO_HIP(d_tb.alloc(n * 8)); NMOD_RO_HIP(d_ok.alloc(n));
  NMOD_RO_HIP(hipMemcpyAsync(d_lo.p, strand_lo, n * 4, hipMemcpyHostToDevice, stream));
  NMOD_RO_HIP(hipMemcpyAsync(d_hi.p, strand_hi, n * 4, hipMemcpyHostToDevice, stream));
  NMOD_RO_HIP(hipMemcpyAsync(d_pos.p, pos, n * 8, hipMemcpyHostToDevice, stream));
  NMOD_RO_HIP(hipMemcpyAsync(d_base.p, base, n, hipMemcpyHostToDevice, stream));
  NMOD_RO_HIP(hipMemcpyAsync(d_val.p, value, n * 8, hipMemcpyHostToDevice, stream));
  RegionArgs ra;
  ra.npos = npos; ra.s_lo = (const int32_t*)d_lo.p; ra.s_hi = (const int32_t*)d_hi.p; ra.pos = (const int64_t*)d_pos.p;
  ra.base = (const char*)d_base.p; ra.value = (const double*)d_val.p; ra.w = w; ra.movesize = movesize; ra.na = na; ra.pct = percentile;
  ra.key = (double*)d_key.p; ra.tb = (double*)d_tb.p; ra.valid = (uint8_t*)d_ok.p;
  hipLaunchKernelGGL(region_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ra);
  NMOD_RO_HIP(hipGetLastError());
  std::vector<double> key(n), tb(n);
  std::vector<uint8_t> ok(n);
  NMOD_RO_HIP(hipMemcpyAsync(key.data(), d_key.p, n * 8, hipMemcpyDeviceToHost, stream));
  NMOD_RO_HIP(hipMemcpyAsync(tb.data(), d_tb.p, n * 8, hipMemcpyDeviceToHost, stream));
  NMOD_RO_HIP(hipMemcpyAsync(ok.data(), d_ok.p, n, hipMemcpyDeviceToHost, stream));
  NMOD_RO_HIP(hipStreamSynchronize(stream));
  // the windows, in generation order (= position order), then ranked by (key, tie-break), stable
  std::vector<int32_t> cand;
  std::vector<double> ck, ct;
  for (size_t i = 0; i < n; ++i) if (ok[i]) { cand.push_back((int32_t)i); ck.push_back(key[i]); ct.push_back(tb[i]); }
  const int64_t nc = (int64_t)cand.size();
  if (nc == 0) return NMOD_OK;
  std::vector<double> zeros((size_t)nc, 0.0);
  std::vector<int32_t> order((size_t)nc);
  int rc = nmod_rank_order(prm, nc, ck.data(), ct.data(), zeros.data(), 0, order.data());
  if (rc != NMOD_OK) return rc;
  int64_t out = 0;
  if (wind_ovlp == 1) {
    std::vector<uint8_t> kept(n, 0);
    for (int64_t r = 0; r < nc; ++r) {
      const int32_t i = cand[(size_t)order[(size_t)r]];
      bool clash = false;
      for (int64_t j = (int64_t)i - 1; j >= strand_lo[i] && pos[i] - pos[j] < w && !clash; --j) clash = kept[(size_t)j] != 0;
      for (int64_t j = (int64_t)i + 1; j <= strand_hi[i] && pos[j] - pos[i] < w && !clash; ++j) clash = kept[(size_t)j] != 0;
      if (clash) continue;
      kept[(size_t)i] = 1;
      ranked_out[out++] = i;
    }
  } else {
    for (int64_t r = 0; r < nc; ++r) ranked_out[out++] = cand[(size_t)order[(size_t)r]];
  }
  *n_ranked = out;
  return NMOD_OK;
}
