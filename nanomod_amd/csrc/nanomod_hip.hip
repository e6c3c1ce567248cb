// C-ABI implementation (include/nanomod_hip.h): argument checking, workspace
// carving, size-class binning, kernel launches on the caller's stream, the
// host-memory staging mode, the HIP-event timer and the wave-primitive self test.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>
#include <string>
#include <stdio.h>
#include <atomic>
#include <mutex>
#include <thread>
#include <functional>
#include <condition_variable>
#include <dlfcn.h>

#include "../../include/nanomod_hip.h"
#include "rank_stats.hpp"
#include "rank_stats_launch.hpp"
#include "pvalue_kernels.hpp"
#include "big_rank.hpp"
#include "rank_all.hpp"
#include "build_info.hpp"

namespace nmod {

thread_local hipError_t g_last_hip = hipSuccess;   // only feeds nmod_strerror's text
thread_local char g_errbuf[256];

#define NMOD_HIP(call)                                   \
  do {                                                   \
    hipError_t e_ = (call);                              \
    if (e_ != hipSuccess) { g_last_hip = e_; return NMOD_ERR_HIP; } \
  } while (0)

// ---------------------------------------------------------------- event timer
struct EvTimer {
  int capacity;
  std::vector<hipEvent_t> start[NMOD_KERNEL_COUNT], stop[NMOD_KERNEL_COUNT];
  int used[NMOD_KERNEL_COUNT];
};

// A slot counts only when BOTH of its events were recorded: a failed hipEventRecord drops the sample (and is
// remembered in g_last_hip) instead of leaving an event pair that nmod_evtimer_read would fail on.
struct ScopedKernelTimer {
  EvTimer* t; int k; hipStream_t s; int slot;
  ScopedKernelTimer(void* timer, int kernel, hipStream_t stream) : t((EvTimer*)timer), k(kernel), s(stream), slot(-1) {
    if (t && t->used[k] < t->capacity) {
      const hipError_t e = hipEventRecord(t->start[k][t->used[k]], s);
      if (e == hipSuccess) slot = t->used[k]; else g_last_hip = e;
    }
  }
  ~ScopedKernelTimer() {
    if (slot >= 0) {
      const hipError_t e = hipEventRecord(t->stop[k][slot], s);
      if (e == hipSuccess) t->used[k] = slot + 1; else g_last_hip = e;
    }
  }
};

// ---------------------------------------------------------------- workspace
struct Workspace {
  uint32_t* ks_num; uint64_t* mwu_s; uint64_t* tie; double* moments;
  double* tmp_ks_d; double* tmp_ks_p; double* ks_d_ref;
  int32_t* order; int32_t* redo; uint8_t* cls; uint8_t* tied; uint8_t* cnt_done; uint8_t* nonfinite; int32_t* meta;
  int32_t* work_list; int32_t* work_meta;      // counting form for any coverage (rank_count_wide.hpp)   // meta: [c] counts, [56 + c] offsets, [112 + c] cursors, [168..169] max n0/n1
  unsigned long long* stats;                                     // nmod_last_dispatch_stats: kStatsWords counters, written only on request
  int64_t bytes;
};
// raw counters of dispatch_stats_kernel: [0] the 256-capacity counting form's gate, [1] positions it produced, [2] float64 redo,
// [kStatsClass + c] positions of launch class c, [kStatsGate + c] the any-coverage counting form's gate of class c (summed over the
// chunks of a host-resident batch), [kStatsLeft + c] positions it left to the class's sorting form
constexpr int kStatsClass = 8, kStatsGate = 64, kStatsLeft = 128, kStatsTried = 192, kStatsWords = 256;
constexpr int kMetaInts = 256;
constexpr int kMetaMax = 3 * kClassStride;      // [168..169] max n0 / n1
constexpr int kMetaBigTotal = kMetaMax + 2;     // [170..171] u64: scratch floats the large positions need
constexpr int kMetaBigCursor = kMetaMax + 4;    // [172..173] u64: bump allocator of big_rank_kernel
constexpr int kMetaWideRedo = kMetaMax + 8;     // [176] positions on the WIDE form's redo list (wide_redo_kernel)
constexpr int kMetaCntGate = kMetaMax + 10;     // [178] the counting form's gate (cnt_probe_kernel: 1 = the batch is event-like)
constexpr int kMetaRedo = 184;                  // float64 front end: [184] count of positions to redo, [184 + kClassStride] = 0 (their
                                                // offset in the list), [242..243] u64 scratch keys they need, [244..245] u64 bump allocator
constexpr int kMetaRedoTotal = 242, kMetaRedoCursor = 244;
static_assert(kMetaRedo + kClassStride < kMetaRedoTotal && kMetaRedoCursor + 2 <= kMetaInts && kMetaBigCursor + 2 <= kMetaWideRedo && kMetaWideRedo < kMetaCntGate && kMetaCntGate < kMetaRedo, "meta layout");
// (kBigClass, kBigHistClass, kWideBigBase .., kNumPairs: rank_stats_launch.hpp)
static_assert(kNumPairs <= kClassStride && kStatsClass + kClassStride <= kStatsGate && kStatsGate + kClassStride <= kStatsLeft && kStatsLeft + kClassStride <= kStatsTried && kStatsTried + kClassStride <= kStatsWords, "class tables");

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

static Workspace carve(void* base, int64_t npos) {
  Workspace w;
  char* p = (char*)base;
  int64_t o = 0;
  auto take = [&](int64_t bytes) { char* r = p ? p + o : nullptr; o += align256(bytes); return r; };
  w.ks_num = (uint32_t*)take(4 * npos);
  w.mwu_s = (uint64_t*)take(8 * npos);
  w.tie = (uint64_t*)take(8 * npos);
  w.moments = (double*)take(32 * npos);
  w.tmp_ks_d = (double*)take(8 * npos);
  w.tmp_ks_p = (double*)take(8 * npos);
  w.ks_d_ref = (double*)take(8 * npos);
  w.order = (int32_t*)take(4 * npos);
  w.redo = (int32_t*)take(4 * npos);
  w.cls = (uint8_t*)take(npos);
  w.tied = (uint8_t*)take(npos);
  w.cnt_done = (uint8_t*)take(npos + 16);        // counting form (rank_count.hpp): one flag byte per entry of the work list, dword per item
  w.nonfinite = (uint8_t*)take(npos);             // NMOD_FLAG_CHECK_FINITE: nonfinite_scan_kernel's flags
  w.meta = (int32_t*)take(kMetaInts * 4);
  w.work_list = (int32_t*)take(4 * npos);          // rank_count_wide.hpp: what the sorting forms still have to do, per class
  w.work_meta = (int32_t*)take(kMetaInts * 4);      // [c] counts, [kClassStride + c] offsets, [2 kClassStride + c] the probes' gates, [3 kClassStride ...] the classes tried
  static_assert(3 * kClassStride + 1 + kClassStride <= kMetaInts, "work_meta: counts, offsets, gates, 1 + one class id per probe block");
  w.stats = (unsigned long long*)take(kStatsWords * 8);
  w.bytes = o;
  return w;
}

// ---------------------------------------------------------------- binning kernels
struct BinArgs {
  int64_t npos; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1;
  int cmax0, cmax1; int ks_only; int allow_big; int force_big; int64_t lim0, lim1; uint8_t* cls; int32_t* meta; int32_t* order;
};


__global__ __launch_bounds__(256) void max_n_kernel(int64_t npos, const int64_t* off0, const int64_t* off1, int32_t* meta) {
  __shared__ int m0s, m1s;
  if (threadIdx.x == 0) { m0s = 0; m1s = 0; }
  __syncthreads();
  int m0 = 0, m1 = 0;
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npos; p += (int64_t)gridDim.x * 256) {
    if (off0) m0 = max(m0, (int)min((int64_t)INT32_MAX, off0[p + 1] - off0[p]));
    if (off1) m1 = max(m1, (int)min((int64_t)INT32_MAX, off1[p + 1] - off1[p]));
  }
  atomicMax(&m0s, m0); atomicMax(&m1s, m1);
  __syncthreads();
  if (threadIdx.x == 0) { atomicMax(&meta[kMetaMax], m0s); atomicMax(&meta[kMetaMax + 1], m1s); }
}

__global__ __launch_bounds__(256) void classify_kernel(BinArgs a) {
  __shared__ int hist[kNumPairs];
  for (int i = threadIdx.x; i < kNumPairs; i += 256) hist[i] = 0;
  __syncthreads();
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.npos; p += (int64_t)gridDim.x * 256) {
    int64_t n0 = a.stride0 > 0 ? a.stride0 : a.off0[p + 1] - a.off0[p];
    int64_t n1 = a.stride1 > 0 ? a.stride1 : a.off1[p + 1] - a.off1[p];
    int c0 = size_class_of(n0), c1 = size_class_of(n1);
    // beyond what the caller promised (or the format allows): skipped, NMOD_STATUS_TOO_LARGE
    const bool over = n0 > a.lim0 || n1 > a.lim1 || n0 > NMOD_MAX_RANKED || n1 > NMOD_MAX_RANKED;
    // beyond the wave-resident kernels: KS-only sorts the smaller group only, all-tests mode sorts both
    const bool big = a.force_big ||                                   // fp64 keys: every position takes big_rank_kernel
                     (a.ks_only ? (c0 < c1 ? c0 : c1) >= kNumSizeClasses : (c0 >= kNumSizeClasses || c1 >= kNumSizeClasses));
    int cid;
    if (over || n0 <= 0 || n1 <= 0 || (big && !a.allow_big)) cid = 255;
    else if (big && !a.ks_only && !a.force_big && (n0 < n1 ? n0 : n1) <= 256 && (n0 < n1 ? n1 : n0) <= kWideBigMaxQ)
      cid = kWideBigBase + (c0 < c1 ? c0 : c1);
    else if (big && !a.ks_only && !a.force_big && (n0 < n1 ? n0 : n1) <= kBigHistMaxS && (n0 < n1 ? n1 : n0) <= kBigHistMaxQ) cid = kBigHistClass;
    else if (big) cid = kBigClass;
    else if (a.ks_only) cid = kKsClassBase + (c0 < c1 ? c0 : c1);
    else cid = (c0 > a.cmax0 || c1 > a.cmax1) ? 255 : launch_class_of(c0, c1);
    if (cid == kBigClass)
      atomicAdd(reinterpret_cast<unsigned long long*>(a.meta + kMetaBigTotal),
                (unsigned long long)(big_pow2_ceil(n0) + big_pow2_ceil(n1)));
    a.cls[p] = (uint8_t)cid;
    if (cid != 255) atomicAdd(&hist[cid], 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kNumPairs; i += 256) if (hist[i]) atomicAdd(&a.meta[i], hist[i]);
}

__global__ void class_offsets_kernel(int32_t* meta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    int acc = 0;
    for (int i = 0; i < kNumPairs; ++i) { meta[kClassStride + i] = acc; meta[2 * kClassStride + i] = 0; acc += meta[i]; }
  }
}

// Every block owns kScatterSpan consecutive positions: it counts their classes, reserves its slots in every class
// list with ONE atomic per class present (all chunks hammering one counter was the whole cost of this kernel:
// 0.21 ms for 4.6 M positions of one class), and writes its positions in order.  Each wave walks the DISTINCT
// classes it holds: one ballot per class present (1-3 for real coverage), not one per class that exists.
constexpr int kScatterSpan = 16 * 256;

__global__ __launch_bounds__(256) void scatter_kernel(BinArgs a) {
  __shared__ int wave_cnt[4][kNumPairs];
  __shared__ int run[kNumPairs];                 // next free slot of this block in every class list
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t span0 = (int64_t)blockIdx.x * kScatterSpan;
  const int64_t span1 = span0 + kScatterSpan < a.npos ? span0 + kScatterSpan : a.npos;
  // pass 1: class counts of the span
  for (int i = threadIdx.x; i < kNumPairs; i += 256) run[i] = 0;
  __syncthreads();
  for (int64_t p0 = span0; p0 < span1; p0 += 256) {
    const int64_t p = p0 + threadIdx.x;
    const int cid = (p < span1) ? a.cls[p] : 255;
    unsigned long long todo = __ballot(cid != 255);
    while (todo != 0ull) {
      const int c = __builtin_amdgcn_readlane(cid, __ffsll((long long)todo) - 1);
      const unsigned long long same = __ballot(cid == c);
      if (lane == 0) atomicAdd(&run[c], __popcll(same));
      todo &= ~same;
    }
  }
  __syncthreads();
  if (threadIdx.x < kNumPairs) {
    const int c = threadIdx.x;
    const int tot = run[c];
    run[c] = a.meta[kClassStride + c] + (tot ? atomicAdd(&a.meta[2 * kClassStride + c], tot) : 0);
  }
  __syncthreads();
  // pass 2: the same walk, now writing; positions of a class keep their order inside the span
  for (int64_t p0 = span0; p0 < span1; p0 += 256) {
    const int64_t p = p0 + threadIdx.x;
    const int cid = (p < span1) ? a.cls[p] : 255;
    for (int i = threadIdx.x; i < 4 * kNumPairs; i += 256) (&wave_cnt[0][0])[i] = 0;
    __syncthreads();
    int rank = 0;
    unsigned long long todo = __ballot(cid != 255);
    while (todo != 0ull) {
      const int c = __builtin_amdgcn_readlane(cid, __ffsll((long long)todo) - 1);
      const unsigned long long same = __ballot(cid == c);
      if (cid == c) rank = __popcll(same & ((1ull << lane) - 1ull));
      if (lane == 0) wave_cnt[wave][c] = __popcll(same);
      todo &= ~same;
    }
    __syncthreads();
    if (cid != 255) {
      for (int w = 0; w < wave; ++w) rank += wave_cnt[w][cid];
      a.order[run[cid] + rank] = (int32_t)p;
    }
    __syncthreads();
    if (threadIdx.x < kNumPairs) {
      const int c = threadIdx.x;
      run[c] += wave_cnt[0][c] + wave_cnt[1][c] + wave_cnt[2][c] + wave_cnt[3][c];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- float64 input (NMOD_DTYPE_F64)
// The reference holds its samples as float64 (myDetect.py:124).  All the rank statistics depend on the ORDER of a
// position's samples only, so every position gets float32 keys that preserve its order and its ties:
//   class 1  every sample of the position is float32-exact                      key = (float)x         (exact)
//   class 2  every sample is k / 1000.0 with |k| <= 2^24 (NanoMod's 3-dp Events)  key = (float)k         (exact)
//   class 3  anything else                                                       key = (float)x, rounded to nearest
// Rounding is monotone (x <= y => key(x) <= key(y)) and keeps ties, so in class 3 the only possible damage is a FALSE
// tie between two different samples.  K1 reports the positions in which keys tie (RankStatsArgs::tied); those among
// the class-3 positions — rare for real-valued signals — are redone by big_rank_kernel<2> on the float64 samples
// themselves.  The Welch moments never come from the keys: f64_moments_kernel takes them from the samples.
struct F64Args {
  const double* d0; const double* d1;          // the caller's samples (index space of the offsets)
  const int64_t* off0; const int64_t* off1; int64_t stride0, stride1;
  int64_t npos;
  float* k0; float* k1;                        // keys, same index space
  uint8_t* cls3;                               // [npos] 1 = class 3
  const uint8_t* tied; const uint8_t* cls;     // K1's tie flags; the size-class byte of the binning (null: no binning ran)
  int32_t* order; int32_t* meta;               // redo list (ws.order) and its counters (ws.meta + kMetaRedo ...)
  double* moments;
};

__device__ __forceinline__ void f64_rows(const F64Args& a, int64_t p, int64_t& o0, int& n0, int64_t& o1, int& n1) {
  if (a.stride0 > 0) { o0 = p * a.stride0; n0 = (int)a.stride0; } else { o0 = a.off0[p]; n0 = (int)(a.off0[p + 1] - o0); }
  if (a.stride1 > 0) { o1 = p * a.stride1; n1 = (int)a.stride1; } else { o1 = a.off1[p]; n1 = (int)(a.off1[p + 1] - o1); }
}

// one wave per position: classify (first sweep), write the keys (second sweep: the samples come from L2)
__global__ __launch_bounds__(256) void f64_encode_kernel(F64Args a) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), ws = (int64_t)gridDim.x * 4;
  for (int64_t p = w0; p < a.npos; p += ws) {
    int64_t o0, o1; int n0, n1;
    f64_rows(a, p, o0, n0, o1, n1);
    bool f32_ok = true, grid_ok = true;
    for (int g = 0; g < 2; ++g) {
      const double* src = (g ? a.d1 + o1 : a.d0 + o0);
      const int n = g ? n1 : n0;
      for (int i = lane; i < n; i += 64) {
        const double v = src[i];
        f32_ok = f32_ok && ((double)(float)v == v);
        const double k = rint(v * 1000.0);
        grid_ok = grid_ok && (fabs(k) <= 16777216.0) && (k / 1000.0 == v);
      }
    }
    const bool all_f32 = __ballot(!f32_ok) == 0ull;
    const bool all_grid = __ballot(!grid_ok) == 0ull;
    const bool scale = !all_f32 && all_grid;
    for (int g = 0; g < 2; ++g) {
      const double* src = (g ? a.d1 + o1 : a.d0 + o0);
      float* dst = (g ? a.k1 + o1 : a.k0 + o0);
      const int n = g ? n1 : n0;
      for (int i = lane; i < n; i += 64) {
        const double v = src[i];
        dst[i] = scale ? (float)rint(v * 1000.0) : (float)v;
      }
    }
    if (lane == 0) a.cls3[p] = (!all_f32 && !all_grid) ? 1 : 0;
  }
}

// mean and sum of squared deviations of both groups from the float64 samples, two-pass like np.mean / np.var
__global__ __launch_bounds__(256) void f64_moments_kernel(F64Args a) {
  const int lane = threadIdx.x & 63;
  const int64_t w0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), ws = (int64_t)gridDim.x * 4;
  for (int64_t p = w0; p < a.npos; p += ws) {
    int64_t o0, o1; int n0, n1;
    f64_rows(a, p, o0, n0, o1, n1);
    for (int g = 0; g < 2; ++g) {
      const double* src = (g ? a.d1 + o1 : a.d0 + o0);
      const int n = g ? n1 : n0;
      double s = 0.0;
      for (int i = lane; i < n; i += 64) s += src[i];
      s = wave_sum_f64(s);
      const double mu = s / (double)n;
      double q = 0.0;
      for (int i = lane; i < n; i += 64) { const double d = src[i] - mu; q += d * d; }
      q = wave_sum_f64(q);
      if (lane == 0) { double* mo = a.moments + p * 4 + 2 * g; mo[0] = mu; mo[1] = q; }
    }
  }
}

// the class-3 positions whose keys tied (or that went to the large-position pass, which reports no ties) -> redo list
__global__ __launch_bounds__(256) void f64_redo_list_kernel(F64Args a) {
  for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < a.npos; p += (int64_t)gridDim.x * 256) {
    if (!a.cls3[p]) continue;
    const bool big = a.cls && (a.cls[p] == (uint8_t)kBigClass || a.cls[p] == (uint8_t)kBigHistClass);
    const bool skipped = a.cls && a.cls[p] == 255;
    if (skipped || !(a.tied[p] || big)) continue;
    int64_t o0, o1; int n0, n1;
    f64_rows(a, p, o0, n0, o1, n1);
    const int idx = atomicAdd(&a.meta[kMetaRedo], 1);
    a.order[idx] = (int32_t)p;
    atomicAdd(reinterpret_cast<unsigned long long*>(a.meta + kMetaRedoTotal), (unsigned long long)(big_pow2_ceil(n0) + big_pow2_ceil(n1)));
  }
}


// ---------------------------------------------------------------- dispatch statistics (nmod_last_dispatch_stats)
// Which K1 form took how many positions of a batch is decided on the device (class lists, the probes' gates, the counting forms'
// per-position flags) and never leaves it on the hot path.  detect_device remembers where those facts sit in the caller's
// workspace; dispatch_stats_kernel reduces them to kStatsWords counters when somebody asks (device-resident batches: on request,
// nothing is launched otherwise; host-resident batches: once per chunk into a per-call accumulator, the path is PCIe-bound).
struct StatsArgs {
  int64_t npos; int32_t uniform_cls;             // >= 0: one class holds all npos positions (no class lists were built)
  const int32_t* meta; const int32_t* work_meta; const int32_t* gates; const uint8_t* cnt_done;
  int32_t cnt256_ran, cw_ran, f64;
  unsigned long long* acc;
};
__global__ __launch_bounds__(256) void dispatch_stats_kernel(StatsArgs a) {
  const int c256 = kNumGeneralClasses + 2;
  auto count_of = [&](int c) -> int64_t { return a.uniform_cls >= 0 ? (c == a.uniform_cls ? a.npos : 0) : (int64_t)a.meta[c]; };
  if (blockIdx.x == 0) {
    const int c = threadIdx.x;
    if (c < kNumPairs) {
      const int64_t n = count_of(c);
      if (n) atomicAdd(&a.acc[kStatsClass + c], (unsigned long long)n);
      if (a.cw_ran && n && a.gates[c] != 0) {
        atomicAdd(&a.acc[kStatsGate + c], 1ull);
        atomicAdd(&a.acc[kStatsTried + c], (unsigned long long)n);
        atomicAdd(&a.acc[kStatsLeft + c], (unsigned long long)a.work_meta[c]);
      }
    }
    if (c == 64 && a.cnt256_ran && a.meta[kMetaCntGate] != 0 && count_of(c256) > 0) { atomicAdd(&a.acc[0], 1ull); atomicAdd(&a.acc[kStatsTried + c256], (unsigned long long)count_of(c256)); }
    if (c == 65 && a.f64) atomicAdd(&a.acc[2], (unsigned long long)a.meta[kMetaRedo]);
  }
  if (a.cnt256_ran && a.meta[kMetaCntGate] != 0) {        // entries of the class list whose flag byte says "produced by rank_count_kernel"
    const int64_t n = count_of(c256);
    unsigned mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) mine += a.cnt_done[i] != 0 ? 1u : 0u;
    mine = (unsigned)wave_sum_u64((unsigned long long)mine);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(&a.acc[1], (unsigned long long)mine);
  }
}
struct DispatchRec {
  bool valid = false; bool host = false; int device = 0; hipStream_t stream = nullptr;
  StatsArgs args;                                // (acc = the workspace's own block for a device-resident batch)
  unsigned long long host_totals[kStatsWords];   // a host-resident batch: read back before the call returned
  int64_t npos = 0; int32_t ks_only = 0;
};
thread_local DispatchRec g_dispatch;
static hipError_t enqueue_dispatch_stats(const StatsArgs& a, hipStream_t stream) {
  const unsigned blocks = a.cnt256_ran ? (unsigned)std::max<int64_t>(1, std::min<int64_t>((a.npos + 255) / 256, 512)) : 1u;
  hipLaunchKernelGGL(dispatch_stats_kernel, dim3(blocks), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------- helpers
static int check_params(const nmod_params* prm) {
  if (!prm || prm->struct_size != (int32_t)sizeof(nmod_params)) return NMOD_ERR_INVALID_ARG;
  if (prm->dtype != NMOD_DTYPE_F32 && prm->dtype != NMOD_DTYPE_I16_MILLI && prm->dtype != NMOD_DTYPE_F64) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_HOST && prm->memspace != NMOD_MEM_DEVICE) return NMOD_ERR_INVALID_ARG;
  if (prm->method < NMOD_METHOD_KS || prm->method > NMOD_METHOD_FISHER) return NMOD_ERR_INVALID_ARG;
  if (prm->nb < 0 || prm->nb > NMOD_MAX_NB) return NMOD_ERR_INVALID_ARG;
  if ((prm->tests & ~NMOD_TEST_ALL) != 0) return NMOD_ERR_INVALID_ARG;
  if ((prm->flags & ~(NMOD_FLAG_KS_RATIONAL_D | NMOD_FLAG_CHECK_FINITE | NMOD_FLAG_NO_COUNTING | NMOD_FLAG_NO_COUNT_WIDE | NMOD_FLAG_NO_HOST_NARROW)) != 0 || prm->reserved != 0) return NMOD_ERR_INVALID_ARG;
  return NMOD_OK;
}

static int fill_combine_args(const nmod_params* prm, CombineArgs& ca) {
  ca.nb = prm->nb; ca.method = prm->method;
  double dif = prm->weights_dif;
  if (prm->method == NMOD_METHOD_STOUFFER && !(dif > 0.0)) return NMOD_ERR_INVALID_ARG;
  // myDetect.py:396-400: 100 in the middle, each step outwards divides by WeightsDif
  int nb = prm->nb;
  ca.w[nb] = 100.0;
  for (int k = 1; k <= nb; ++k) { ca.w[nb - k] = ca.w[nb - k + 1] / dif; ca.w[nb + k] = ca.w[nb + k - 1] / dif; }
  double ss = 0.0;
  for (int k = 0; k <= 2 * nb; ++k) ss += ca.w[k] * ca.w[k];
  ca.wnorm = sqrt(ss);
  return NMOD_OK;
}

static int launch_combine(const nmod_params* prm, hipStream_t stream, int64_t npos, const double* ks_d,
                          const double* ks_p, const int32_t* run_id, double* comb_st, double* comb_p) {
  CombineArgs ca;
  memset(&ca, 0, sizeof(ca));
  int rc = fill_combine_args(prm, ca);
  if (rc != NMOD_OK) return rc;
  ca.npos = npos; ca.ks_d = ks_d; ca.ks_p = ks_p; ca.run_id = run_id; ca.comb_st = comb_st; ca.comb_p = comb_p;
  unsigned blocks = (unsigned)((npos + kCombineTile - 1) / kCombineTile);
  if (blocks == 0) return NMOD_OK;
  ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_COMBINE, stream);
  hipLaunchKernelGGL(combine_kernel, dim3(blocks), dim3(kCombineTile), 0, stream, ca);
  NMOD_HIP(hipGetLastError());
  return NMOD_OK;
}

// Scratch slab of the large-position pass: stream-ordered allocation from a memory pool the LIBRARY owns (one per
// device, created on first use).  Freed slabs stay in that pool (release threshold = max: with the default
// threshold every batch would return its slab to the OS at the next synchronisation and map it again — measured
// 20 ms per 1.3 GB); the device's default pool, which the caller's own hipMallocAsync traffic uses, is never
// touched.  nmod_trim_scratch() returns the cached slabs to the driver.
constexpr int kMaxDevices = 64;
static std::mutex g_pool_mutex;
static hipMemPool_t g_pool[kMaxDevices] = {nullptr};        // guarded by g_pool_mutex

static hipMemPool_t scratch_pool(int dev) {
  if (dev < 0 || dev >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (!g_pool[dev]) {
    hipMemPoolProps props;
    memset(&props, 0, sizeof(props));
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = dev;
    hipMemPool_t pool = nullptr;
    if (hipMemPoolCreate(&pool, &props) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    uint64_t keep = UINT64_MAX;
    (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
    g_pool[dev] = pool;
  }
  return g_pool[dev];
}

struct DevScratch {
  void* p = nullptr; bool async = false; hipStream_t owner = nullptr;
  hipError_t alloc(size_t bytes, hipStream_t s, int dev) {
    owner = s;
    hipMemPool_t pool = scratch_pool(dev);
    if (pool && hipMallocFromPoolAsync(&p, bytes ? bytes : 4, pool, s) == hipSuccess) { async = true; return hipSuccess; }
    (void)hipGetLastError();
    p = nullptr;
    return hipMalloc(&p, bytes ? bytes : 4);
  }
  hipError_t release(hipStream_t s) {
    hipError_t e = hipSuccess;
    if (p) e = async ? hipFreeAsync(p, s) : (hipStreamSynchronize(s), hipFree(p));
    p = nullptr;
    return e;
  }
  // an early return (error path) with kernels of the allocation stream still queued: the slab goes back to the pool
  // ordered behind them on THAT stream — freeing on the null stream does not order against a non-blocking stream
  ~DevScratch() { if (p) { if (async) hipFreeAsync(p, owner); else { hipStreamSynchronize(owner); hipFree(p); } } }
};

// the float64 samples behind float32 keys (detect_f64): device pointers in the index space of the offsets
struct F64Src { const double* d0; const double* d1; uint8_t* cls3; };

static int detect_device(const nmod_params* prm, int64_t npos, const void* sig0, const int64_t* off0,
                         const void* sig1, const int64_t* off1, const int32_t* run_id, void* workspace,
                         int64_t workspace_bytes, nmod_out* out, const F64Src* f64 = nullptr) {
  hipStream_t stream = (hipStream_t)prm->stream;
  if (npos == 0) return NMOD_OK;
  if (!sig0 || !sig1 || !out) return NMOD_ERR_INVALID_ARG;
  if ((prm->stride0 <= 0 && !off0) || (prm->stride1 <= 0 && !off1)) return NMOD_ERR_INVALID_ARG;
  if (npos > INT32_MAX) return NMOD_ERR_INVALID_ARG;
  const bool want_comb = prm->method != NMOD_METHOD_KS && (out->comb_st || out->comb_p);
  if (want_comb && (!out->comb_st || !out->comb_p)) return NMOD_ERR_INVALID_ARG;
  if (want_comb && prm->nb > 0 && !run_id) return NMOD_ERR_INVALID_ARG;
  Workspace ws = carve(workspace, npos);
  if (!workspace || workspace_bytes < ws.bytes) return NMOD_ERR_WORKSPACE;

  // CU count per device, looked up once (the attribute query is not cheap and this runs every batch); an atomic
  // per device: concurrent first calls both query and store the same value
  static std::atomic<int> cu_cache[kMaxDevices];
  int num_cus = (prm->device >= 0 && prm->device < kMaxDevices) ? cu_cache[prm->device].load(std::memory_order_relaxed) : 0;
  if (num_cus <= 0) {
    NMOD_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, prm->device));
    if (prm->device >= 0 && prm->device < kMaxDevices) cu_cache[prm->device].store(num_cus, std::memory_order_relaxed);
  }

  int tests = prm->tests;
  if (want_comb) tests |= NMOD_TEST_KS;
  const bool all = (tests & (NMOD_TEST_MWU | NMOD_TEST_WELCH)) != 0 || prm->want_mstd;

  // ---- size classes
  const bool uniform = prm->stride0 > 0 && prm->stride1 > 0;
  int64_t max0 = prm->stride0 > 0 ? prm->stride0 : prm->max_n0;
  int64_t max1 = prm->stride1 > 0 ? prm->stride1 : prm->max_n1;
  // the class counters / cursors / maxima in ws.meta: cleared once per batch, and not at all for a fixed-stride batch of
  // wave-resident positions (one launch, no lists: the common benchmark shape pays no memset per step)
  bool meta_cleared = false;
  if (max0 <= 0 || max1 <= 0) {
    NMOD_HIP(hipMemsetAsync(ws.meta, 0, kMetaInts * 4, stream));
    meta_cleared = true;
    hipLaunchKernelGGL(max_n_kernel, dim3(1024), dim3(256), 0, stream, npos,
                       prm->stride0 > 0 ? nullptr : off0, prm->stride1 > 0 ? nullptr : off1, ws.meta);
    NMOD_HIP(hipGetLastError());
    int32_t mx[2];
    NMOD_HIP(hipMemcpyAsync(mx, ws.meta + kMetaMax, 8, hipMemcpyDeviceToHost, stream));
    NMOD_HIP(hipStreamSynchronize(stream));
    if (max0 <= 0) max0 = mx[0];
    if (max1 <= 0) max1 = mx[1];
  }
  // a group beyond NMOD_MAX_RANKED: that position is skipped by the classifier and flagged NMOD_STATUS_TOO_LARGE by K2, the rest
  // of the batch is computed (the reference has no limit: myDetect.py:327-343)
  max0 = std::min<int64_t>(max0, NMOD_MAX_RANKED); max1 = std::min<int64_t>(max1, NMOD_MAX_RANKED);
  int cmax0 = size_class_of(std::max<int64_t>(max0, 1)), cmax1 = size_class_of(std::max<int64_t>(max1, 1));
  // positions beyond the wave-resident kernels (both groups sorted in all-tests mode, the smaller one in KS-only
  // mode) go to big_rank_kernel; the maxima tell whether any can exist
  const bool big_possible = all ? (cmax0 >= kNumSizeClasses || cmax1 >= kNumSizeClasses)
                                : (std::min(cmax0, cmax1) >= kNumSizeClasses);
  cmax0 = std::min(cmax0, kNumSizeClasses - 1); cmax1 = std::min(cmax1, kNumSizeClasses - 1);
  if (!meta_cleared && (!(uniform && !big_possible) || f64)) NMOD_HIP(hipMemsetAsync(ws.meta, 0, kMetaInts * 4, stream));

  RankStatsArgs ra;
  memset(&ra, 0, sizeof(ra));
  ra.sig0 = sig0; ra.sig1 = sig1; ra.off0 = off0; ra.off1 = off1;
  ra.stride0 = prm->stride0 > 0 ? prm->stride0 : 0; ra.stride1 = prm->stride1 > 0 ? prm->stride1 : 0;
  ra.npos = npos; ra.ks_num = ws.ks_num; ra.mwu_s = ws.mwu_s; ra.tie = ws.tie; ra.moments = ws.moments; ra.ks_d_ref = ws.ks_d_ref;
  ra.ks_rational_d = (!all && (prm->flags & NMOD_FLAG_KS_RATIONAL_D)) ? 1 : 0;
  ra.tied = f64 ? ws.tied : nullptr;            // float32 keys of float64 samples: K1 reports the positions whose keys tie
  ra.redo_list = ws.redo; ra.redo_count = ws.meta + kMetaWideRedo;
  // the WIDE float32 form may put positions on its redo list: wide_redo_kernel finishes them (count read on the device)
  auto launch_wide_redo = [&]() -> int {
    WideRedoArgs wr;
    wr.sig0 = sig0; wr.sig1 = sig1; wr.off0 = off0; wr.off1 = off1; wr.stride0 = ra.stride0; wr.stride1 = ra.stride1;
    wr.list = ws.redo; wr.count = ws.meta + kMetaWideRedo; wr.tie = ws.tie;
    hipLaunchKernelGGL(wide_redo_kernel, dim3((unsigned)std::min<int64_t>(npos, (int64_t)num_cus * 2)), dim3(kBigThreads), 0, stream, wr);
    NMOD_HIP(hipGetLastError());
    return NMOD_OK;
  };
  const bool wide_f32 = all && prm->dtype == NMOD_DTYPE_F32;
  // all tests on capacity-256 positions: the counting form is tried first (rank_count.hpp; a device-side probe decides whether
  // the batch is event-like, positions it cannot take fall through to rank_hist_kernel).  The float32 images of float64 samples
  // qualify where they are whole numbers (positions on the 0.001 grid carry k as keys).  NMOD_FLAG_NO_COUNTING turns it off (A/B, parity tests).
  const bool counting_off = (prm->flags & NMOD_FLAG_NO_COUNTING) != 0;
  if (all && !counting_off) { ra.cnt_gate = ws.meta + kMetaCntGate; ra.cnt_done = ws.cnt_done; }

  auto launch = [&](int cls, int64_t work) -> hipError_t {
    ra.class_id = cls;
    if (prm->dtype == NMOD_DTYPE_F32)
      return all ? launch_rank_stats_d0_a1(cls, num_cus, work, stream, ra)
                 : launch_rank_stats_d0_a0(cls, num_cus, work, stream, ra);
    return all ? launch_rank_stats_d1_a1(cls, num_cus, work, stream, ra)
               : launch_rank_stats_d1_a0(cls, num_cus, work, stream, ra);
  };

  // all tests, classes other than the 256-capacity one: the counting form for any coverage (rank_count_wide.hpp) is tried per class
  // when its probe finds the class event-like; the sorting form of the class then runs over the work list that is left
  // (NMOD_FLAG_NO_COUNT_WIDE turns it off)
  const bool cw_off = (prm->flags & NMOD_FLAG_NO_COUNT_WIDE) != 0;
  // (KS-only batches too — the form without the tie term and the moments — when a group of the batch can reach the size it takes)
  const bool cw_on = !counting_off && !cw_off && (all || std::max(max0, max1) >= kCwKsMinQ);
  CountWideWs cww;
  cww.gates = ws.work_meta + 2 * kClassStride; cww.work_list = ws.work_list; cww.work_meta = ws.work_meta;
  auto cw_prepare = [&](const std::vector<int>& classes) -> hipError_t {
    if (classes.empty()) return hipSuccess;
    if (prm->dtype == NMOD_DTYPE_F32)
      return all ? launch_count_wide_prepare_d0_a1(classes.data(), (int)classes.size(), stream, ra, cww)
                 : launch_count_wide_prepare_d0_a0(classes.data(), (int)classes.size(), stream, ra, cww);
    return all ? launch_count_wide_prepare_d1_a1(classes.data(), (int)classes.size(), stream, ra, cww)
               : launch_count_wide_prepare_d1_a0(classes.data(), (int)classes.size(), stream, ra, cww);
  };
  // ... then the counting form over every class the probe accepted and the compaction of what it left, in one launch each
  auto cw_run = [&](const std::vector<int>& classes) -> hipError_t {
    if (classes.empty()) return hipSuccess;
    hipError_t e = cw_prepare(classes);
    if (e != hipSuccess) return e;
    bool vc = false;                                          // (both groups above 1 024 samples: the value-domain form, all tests)
    for (int c : classes) vc = vc || count_wide_rs_index(c) == 5;
    if (prm->dtype == NMOD_DTYPE_F32)
      return all ? launch_count_wide_run_d0_a1(num_cus, npos, stream, ra, cww, vc) : launch_count_wide_run_d0_a0(num_cus, npos, stream, ra, cww, vc);
    return all ? launch_count_wide_run_d1_a1(num_cus, npos, stream, ra, cww, vc) : launch_count_wide_run_d1_a0(num_cus, npos, stream, ra, cww, vc);
  };
  // a class's sorting form; where the counting form was in play (its gate is set, on the device) it walks the work list
  auto launch_class = [&](int cls, int64_t work, bool counted) -> hipError_t {
    ra.alt_gates = counted ? cww.gates : nullptr; ra.alt_list = ws.work_list; ra.alt_meta = ws.work_meta;
    const hipError_t e = launch(cls, work);
    ra.alt_gates = nullptr;
    return e;
  };

  DevScratch big_scratch;
  int st_uniform_cls = -1, st_cnt256 = 0, st_cw = 0;          // for nmod_last_dispatch_stats
  constexpr int kCls256 = kNumGeneralClasses + 2;
  if (uniform && !big_possible) {
    const int ucls = all ? launch_class_of(cmax0, cmax1) : kKsClassBase + std::min(cmax0, cmax1);
    st_uniform_cls = ucls; st_cnt256 = (ucls == kCls256 && ra.cnt_gate) ? 1 : 0;
    const bool uwide = wide_f32 && wide_class(ucls);
    if (uwide && !meta_cleared && !f64) NMOD_HIP(hipMemsetAsync(ws.meta + kMetaWideRedo, 0, 4, stream));   // (the meta block is not cleared for uniform batches)
    ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_RANK_STATS, stream);
    const bool counted = cw_on && count_wide_rs_index(ucls) >= 0;
    st_cw = counted ? 1 : 0;
    if (counted) NMOD_HIP(cw_run(std::vector<int>{ucls}));
    NMOD_HIP(launch_class(ucls, npos, counted));
    if (uwide) { const int rr = launch_wide_redo(); if (rr != NMOD_OK) return rr; }
  } else {
    BinArgs ba;
    ba.npos = npos; ba.off0 = off0; ba.off1 = off1; ba.stride0 = ra.stride0; ba.stride1 = ra.stride1;
    ba.cmax0 = cmax0; ba.cmax1 = cmax1; ba.ks_only = all ? 0 : 1; ba.allow_big = 1; ba.force_big = 0; ba.lim0 = std::max<int64_t>(max0, 1); ba.lim1 = std::max<int64_t>(max1, 1);
    ba.cls = ws.cls; ba.meta = ws.meta; ba.order = ws.order;
    unsigned blocks = (unsigned)std::min<int64_t>((npos + 255) / 256, 4096);
    hipLaunchKernelGGL(classify_kernel, dim3(blocks), dim3(256), 0, stream, ba);
    hipLaunchKernelGGL(class_offsets_kernel, dim3(1), dim3(64), 0, stream, ws.meta);
    hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((npos + kScatterSpan - 1) / kScatterSpan)), dim3(256), 0, stream, ba);
    NMOD_HIP(hipGetLastError());
    ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_RANK_STATS, stream);
    bool wanted[kNumClasses] = {false};
    for (int c0 = 0; c0 <= cmax0; ++c0)
      for (int c1 = 0; c1 <= cmax1; ++c1) wanted[all ? launch_class_of(c0, c1) : kKsClassBase + std::min(c0, c1)] = true;
    ra.pos_list = ws.order; ra.class_meta = ws.meta;
    std::vector<int> counted_classes;
    if (cw_on) {
      for (int cls = 0; cls < kNumClasses; ++cls) if (wanted[cls] && count_wide_rs_index(cls) >= 0) counted_classes.push_back(cls);
      if (big_possible && all) for (int cs = 0; cs < kNumWideBig; ++cs) counted_classes.push_back(kWideBigBase + cs);
      NMOD_HIP(cw_run(counted_classes));
      st_cw = counted_classes.empty() ? 0 : 1;
    }
    st_cnt256 = (all && wanted[kCls256] && ra.cnt_gate) ? 1 : 0;
    for (int cls = 0; cls < kNumClasses; ++cls) {
      if (!wanted[cls]) continue;
      ra.pos_list = ws.order; ra.class_meta = ws.meta; ra.class_id = cls;
      NMOD_HIP(launch_class(cls, npos, cw_on && count_wide_rs_index(cls) >= 0));
    }
    if (big_possible && all) {
      // larger group of 2 049 .. 4 096 samples against at most 256: rank_hist_kernel WIDE as well
      for (int cs = 0; cs < kNumWideBig; ++cs) {
        ra.pos_list = ws.order; ra.class_meta = ws.meta;
        NMOD_HIP(launch_class(kWideBigBase + cs, npos, cw_on));
      }
    }
    if (wide_f32) {
      bool any_wide = big_possible;
      for (int cls = 0; cls < kNumClasses && !any_wide; ++cls) any_wide = wanted[cls] && wide_class(cls);
      if (any_wide) { const int rr = launch_wide_redo(); if (rr != NMOD_OK) return rr; }
    }
    if (big_possible) {
      // the only host round trip of this path: how many large positions, how much scratch
      int32_t head[4];
      NMOD_HIP(hipMemcpyAsync(&head[0], ws.meta + kBigClass, 8, hipMemcpyDeviceToHost, stream));     // kBigClass, kBigHistClass
      NMOD_HIP(hipMemcpyAsync(&head[2], ws.meta + kMetaBigTotal, 8, hipMemcpyDeviceToHost, stream));
      NMOD_HIP(hipStreamSynchronize(stream));
      const int64_t nbig = head[0], nbig_hist = head[1];
      unsigned long long total;
      memcpy(&total, &head[2], 8);
      if (nbig > 0) {
        NMOD_HIP(big_scratch.alloc((size_t)total * 4, stream, prm->device));
        BigArgs bg;
        memset(&bg, 0, sizeof(bg));
        bg.sig0 = sig0; bg.sig1 = sig1; bg.off0 = off0; bg.off1 = off1; bg.stride0 = ra.stride0; bg.stride1 = ra.stride1;
        bg.pos_list = ws.order; bg.class_meta = ws.meta; bg.big_class = kBigClass; bg.all = all ? 1 : 0;
        bg.scratch = big_scratch.p;
        bg.cursor = reinterpret_cast<unsigned long long*>(ws.meta + kMetaBigCursor);
        bg.ks_num = ws.ks_num; bg.mwu_s = ws.mwu_s; bg.tie = ws.tie; bg.moments = ws.moments; bg.ks_d_ref = ws.ks_d_ref;
        const unsigned blocks = (unsigned)std::min<int64_t>(nbig, (int64_t)num_cus * 4);   // 4 x 33 KB of LDS per CU
        if (prm->dtype == NMOD_DTYPE_F32) hipLaunchKernelGGL(big_rank_kernel<0>, dim3(blocks), dim3(kBigThreads), 0, stream, bg);
        else hipLaunchKernelGGL(big_rank_kernel<1>, dim3(blocks), dim3(kBigThreads), 0, stream, bg);
        NMOD_HIP(hipGetLastError());
        NMOD_HIP(big_scratch.release(stream));
      }
      if (nbig_hist > 0) {
        BigArgs bg;
        memset(&bg, 0, sizeof(bg));
        bg.sig0 = sig0; bg.sig1 = sig1; bg.off0 = off0; bg.off1 = off1; bg.stride0 = ra.stride0; bg.stride1 = ra.stride1;
        bg.pos_list = ws.order; bg.class_meta = ws.meta; bg.big_class = kBigHistClass; bg.all = 1;
        bg.ks_num = ws.ks_num; bg.mwu_s = ws.mwu_s; bg.tie = ws.tie; bg.moments = ws.moments; bg.ks_d_ref = ws.ks_d_ref;
        const unsigned blocks = (unsigned)std::min<int64_t>(nbig_hist, (int64_t)num_cus * 2);        // 80 KB of LDS per block
        auto kfn = prm->dtype == NMOD_DTYPE_F32 ? big_hist_kernel<0> : big_hist_kernel<1>;
        NMOD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigHistLds));
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(kBigThreads), kBigHistLds, stream, bg);
        NMOD_HIP(hipGetLastError());
      }
    }
  }

  // ---- float64 samples behind the keys: redo the positions whose keys tied with 64-bit keys, moments from the samples
  DevScratch redo_scratch;
  if (f64) {
    F64Args fx;
    memset(&fx, 0, sizeof(fx));
    fx.d0 = f64->d0; fx.d1 = f64->d1; fx.off0 = off0; fx.off1 = off1; fx.stride0 = ra.stride0; fx.stride1 = ra.stride1;
    fx.npos = npos; fx.cls3 = f64->cls3; fx.tied = ws.tied; fx.cls = (uniform && !big_possible) ? nullptr : ws.cls;
    fx.order = ws.order; fx.meta = ws.meta; fx.moments = ws.moments;
    const unsigned gb = (unsigned)std::min<int64_t>((npos + 255) / 256, 4096);
    hipLaunchKernelGGL(f64_redo_list_kernel, dim3(gb), dim3(256), 0, stream, fx);
    NMOD_HIP(hipGetLastError());
    int32_t nredo = 0; unsigned long long total = 0;
    NMOD_HIP(hipMemcpyAsync(&nredo, ws.meta + kMetaRedo, 4, hipMemcpyDeviceToHost, stream));
    NMOD_HIP(hipMemcpyAsync(&total, ws.meta + kMetaRedoTotal, 8, hipMemcpyDeviceToHost, stream));
    NMOD_HIP(hipStreamSynchronize(stream));
    if (nredo > 0) {
      NMOD_HIP(redo_scratch.alloc((size_t)total * 8, stream, prm->device));
      BigArgs bg;
      memset(&bg, 0, sizeof(bg));
      bg.sig0 = f64->d0; bg.sig1 = f64->d1; bg.off0 = off0; bg.off1 = off1; bg.stride0 = ra.stride0; bg.stride1 = ra.stride1;
      bg.pos_list = ws.order; bg.class_meta = ws.meta + kMetaRedo; bg.big_class = 0; bg.all = all ? 1 : 0;
      bg.scratch = redo_scratch.p;
      bg.cursor = reinterpret_cast<unsigned long long*>(ws.meta + kMetaRedoCursor);
      bg.ks_num = ws.ks_num; bg.mwu_s = ws.mwu_s; bg.tie = ws.tie; bg.moments = ws.moments; bg.ks_d_ref = ws.ks_d_ref;
      const unsigned blocks = (unsigned)std::min<int64_t>(nredo, (int64_t)num_cus * 4);
      ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_RANK_STATS, stream);
      hipLaunchKernelGGL(big_rank_kernel<2>, dim3(blocks), dim3(kBigThreads), 0, stream, bg);
      NMOD_HIP(hipGetLastError());
      NMOD_HIP(redo_scratch.release(stream));
    }
    if (all) {
      const unsigned mb = (unsigned)std::min<int64_t>((npos + 3) / 4, (int64_t)num_cus * 16);
      hipLaunchKernelGGL(f64_moments_kernel, dim3(mb), dim3(256), 0, stream, fx);
      NMOD_HIP(hipGetLastError());
    }
  }

  // ---- p-values
  FinalizeArgs fa;
  memset(&fa, 0, sizeof(fa));
  fa.npos = npos; fa.off0 = off0; fa.off1 = off1; fa.stride0 = ra.stride0; fa.stride1 = ra.stride1;
  fa.ks_num = ws.ks_num; fa.mwu_s = ws.mwu_s; fa.tie = ws.tie; fa.moments = ws.moments; fa.ks_d_ref = ra.ks_rational_d ? nullptr : ws.ks_d_ref;   // every K1 form writes ks_2samp's float form of D (unless the caller opted out)
  fa.tests = tests; fa.want_mstd = prm->want_mstd; fa.out = *out;
  // what K1 covered: exactly the limits the classifier used (the promised / measured maxima), in every mode —
  // a position beyond them was skipped by K1 and must be flagged TOO_LARGE here, never read from the workspace
  fa.max_n0 = std::max<int64_t>(max0, 1);
  fa.max_n1 = std::max<int64_t>(max1, 1);
  fa.min_cap = 0;
  if ((prm->flags & NMOD_FLAG_CHECK_FINITE) && prm->dtype != NMOD_DTYPE_I16_MILLI) {
    // one pass over the samples: the float64 samples themselves where the keys are their float32 images
    NonfiniteArgs na;
    memset(&na, 0, sizeof(na));
    na.sig0 = f64 ? (const void*)f64->d0 : sig0; na.sig1 = f64 ? (const void*)f64->d1 : sig1; na.f64 = f64 ? 1 : 0;
    na.off0 = off0; na.off1 = off1; na.stride0 = ra.stride0; na.stride1 = ra.stride1; na.npos = npos;
    na.lim0 = fa.max_n0; na.lim1 = fa.max_n1; na.flag = ws.nonfinite;
    const unsigned nb_ = (unsigned)std::min<int64_t>((npos + 3) / 4, (int64_t)num_cus * 32);
    hipLaunchKernelGGL(nonfinite_scan_kernel, dim3(nb_), dim3(256), 0, stream, na);
    NMOD_HIP(hipGetLastError());
    fa.nonfinite = ws.nonfinite;
  }
  if (want_comb) {                       // the combine needs the KS track even if the caller does not
    if (!fa.out.ks_d) fa.out.ks_d = ws.tmp_ks_d;
    if (!fa.out.ks_p) fa.out.ks_p = ws.tmp_ks_p;
  }
  {
    ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_FINALIZE, stream);
    hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, stream, fa);
    NMOD_HIP(hipGetLastError());
  }
  if (want_comb) {
    int rc = launch_combine(prm, stream, npos, fa.out.ks_d, fa.out.ks_p, run_id, out->comb_st, out->comb_p);
    if (rc != NMOD_OK) return rc;
  }
  {
    DispatchRec& d = g_dispatch;
    d.valid = true; d.host = false; d.device = prm->device; d.stream = stream; d.npos = npos; d.ks_only = all ? 0 : 1;
    d.args.npos = npos; d.args.uniform_cls = st_uniform_cls; d.args.meta = ws.meta; d.args.work_meta = ws.work_meta;
    d.args.gates = ws.work_meta + 2 * kClassStride; d.args.cnt_done = ws.cnt_done; d.args.cnt256_ran = st_cnt256; d.args.cw_ran = st_cw;
    d.args.f64 = f64 ? 1 : 0; d.args.acc = ws.stats;
  }
  return NMOD_OK;
}

// ---------------------------------------------------------------- host staging (nmod_combine_track's small buffers)
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

// ---------------------------------------------------------------- float64 front end
// Gives every position order-preserving float32 keys (f64_encode_kernel) and runs the batch on them; see F64Args.
// Device memory only (the host entry hands its chunks over as device memory, host_pipeline.hpp).  bounds: the first and
// one-past-last element of each array in use, {b0, e0, b1, e1}, when the caller knows them (no round trip for the offsets).
static int detect_f64(const nmod_params* prm, int64_t npos, const void* sig0, const int64_t* off0,
                      const void* sig1, const int64_t* off1, const int32_t* run_id, void* workspace,
                      int64_t workspace_bytes, nmod_out* out, const int64_t* bounds = nullptr) {
  if (npos == 0) return NMOD_OK;
  if (!sig0 || !sig1 || !out) return NMOD_ERR_INVALID_ARG;
  if ((prm->stride0 <= 0 && !off0) || (prm->stride1 <= 0 && !off1)) return NMOD_ERR_INVALID_ARG;
  hipStream_t stream = (hipStream_t)prm->stream;
  // sample counts and the first sample of each array (offsets need not start at 0)
  int64_t b0 = 0, e0 = prm->stride0 * npos, b1 = 0, e1 = prm->stride1 * npos;
  if (bounds) { b0 = bounds[0]; e0 = bounds[1]; b1 = bounds[2]; e1 = bounds[3]; }
  else {
    if (prm->stride0 <= 0) {
      int64_t t[2]; NMOD_HIP(hipMemcpyAsync(&t[0], off0, 8, hipMemcpyDeviceToHost, stream));
      NMOD_HIP(hipMemcpyAsync(&t[1], off0 + npos, 8, hipMemcpyDeviceToHost, stream)); NMOD_HIP(hipStreamSynchronize(stream)); b0 = t[0]; e0 = t[1];
    }
    if (prm->stride1 <= 0) {
      int64_t t[2]; NMOD_HIP(hipMemcpyAsync(&t[0], off1, 8, hipMemcpyDeviceToHost, stream));
      NMOD_HIP(hipMemcpyAsync(&t[1], off1 + npos, 8, hipMemcpyDeviceToHost, stream)); NMOD_HIP(hipStreamSynchronize(stream)); b1 = t[0]; e1 = t[1];
    }
  }
  const int64_t n0 = e0 - b0, n1 = e1 - b1;
  if (n0 < 0 || n1 < 0) return NMOD_ERR_INVALID_ARG;
  DevScratch enc0, enc1, d_cls3;                    // stream-ordered, from the library's pool: no device-wide synchronisation per batch
  NMOD_HIP(enc0.alloc((size_t)n0 * 4, stream, prm->device)); NMOD_HIP(enc1.alloc((size_t)n1 * 4, stream, prm->device));
  NMOD_HIP(d_cls3.alloc((size_t)npos, stream, prm->device));
  // pointers in the index space of the offsets (sample i of the arrays sits at [i], whatever the first offset is)
  F64Src src;
  src.d0 = (const double*)sig0; src.d1 = (const double*)sig1; src.cls3 = (uint8_t*)d_cls3.p;
  F64Args fx;
  memset(&fx, 0, sizeof(fx));
  fx.d0 = src.d0; fx.d1 = src.d1; fx.off0 = off0; fx.off1 = off1;
  fx.stride0 = prm->stride0 > 0 ? prm->stride0 : 0; fx.stride1 = prm->stride1 > 0 ? prm->stride1 : 0;
  fx.npos = npos; fx.k0 = (float*)enc0.p - b0; fx.k1 = (float*)enc1.p - b1; fx.cls3 = src.cls3;
  int num_cus = 0;
  NMOD_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, prm->device));
  const unsigned eb = (unsigned)std::min<int64_t>((npos + 3) / 4, (int64_t)num_cus * 16);
  hipLaunchKernelGGL(f64_encode_kernel, dim3(eb), dim3(256), 0, stream, fx);
  NMOD_HIP(hipGetLastError());
  nmod_params ep = *prm;
  ep.dtype = NMOD_DTYPE_F32;
  int rc = detect_device(&ep, npos, fx.k0, off0, fx.k1, off1, run_id, workspace, workspace_bytes, out, &src);
  // the key buffers go back to the pool in stream order
  const hipError_t r0 = enc0.release(stream), r1 = enc1.release(stream), r2 = d_cls3.release(stream);
  if (rc == NMOD_OK && (r0 != hipSuccess || r1 != hipSuccess || r2 != hipSuccess)) { g_last_hip = r0 != hipSuccess ? r0 : (r1 != hipSuccess ? r1 : r2); rc = NMOD_ERR_HIP; }
  return rc;
}

#include "host_pipeline.hpp"


// ---------------------------------------------------------------- down-sampling branch (myDetect.py:345-361)
// Virtual rows: row v = (flagged position k, iteration it) of group g holds rsz[k] samples — drawn WITH replacement from the
// position's rn[k] samples when rn[k] > rsz[k] (np.random.choice semantics), the samples themselves otherwise.  The draws are a
// counter-based function of (seed, k, it, g, j): reproducible, any launch shape.
struct ResampleArgs {
  const char* rows; int esz; const int64_t* roff; const int32_t* rn; const int32_t* rsz; const int64_t* vbase;
  int64_t nrows; int32_t iters; uint64_t seed; int32_t group; int64_t k0;      // k0: index of the chunk's first position among all flagged ones
  char* out; int64_t* voff;                                                     // voff[nrows * iters + 1]: offsets of the virtual rows
};
__device__ __forceinline__ uint64_t draw_mix(uint64_t seed, int64_t k, int32_t it, int32_t g, uint32_t j) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (uint64_t)((k * 2 + g) * 1000003ll + it);
  x ^= (uint64_t)j * 0xD1B54A32D192ED03ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
__global__ __launch_bounds__(256) void resample_kernel(ResampleArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t nv = a.nrows * a.iters;
  for (int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); v < nv; v += (int64_t)gridDim.x * 4) {
    const int64_t k = v / a.iters;
    const int32_t it = (int32_t)(v - k * a.iters);
    const int n = a.rn[k], sz = a.rsz[k];
    const int64_t vo = a.vbase[k] + (int64_t)it * sz;
    if (lane == 0) { a.voff[v] = vo; if (v == nv - 1) a.voff[nv] = vo + sz; }
    const char* src = a.rows + a.roff[k] * a.esz;
    char* dst = a.out + vo * a.esz;
    const bool draw = n > sz;
    for (int j = lane; j < sz; j += 64) {
      int64_t idx = j;
      if (draw) idx = (int64_t)__umul64hi(draw_mix(a.seed, a.k0 + k, it, a.group, (uint32_t)j), (uint64_t)n);   // floor(u n / 2^64): uniform on [0, n)
      if (a.esz == 4) reinterpret_cast<uint32_t*>(dst)[j] = reinterpret_cast<const uint32_t*>(src)[idx];
      else if (a.esz == 2) reinterpret_cast<uint16_t*>(dst)[j] = reinterpret_cast<const uint16_t*>(src)[idx];
      else reinterpret_cast<uint64_t*>(dst)[j] = reinterpret_cast<const uint64_t*>(src)[idx];
    }
  }
}
// the pair at index `kth` of the p-sorted iterations of every position (np.argsort(p_array)[kth]; ties in p by iteration number)
__global__ __launch_bounds__(256) void select_quantile_kernel(int64_t nrows, int32_t iters, int32_t kth, const double* d, const double* p,
                                                              double* out_d, double* out_p) {
  const int lane = threadIdx.x & 63;
  for (int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < nrows; k += (int64_t)gridDim.x * 4) {
    const double* pk = p + k * iters;
    for (int e = lane; e < iters; e += 64) {
      const double pe = pk[e];
      int rank = 0;
      for (int j = 0; j < iters; ++j) { const double pj = pk[j]; rank += (pj < pe || (pj == pe && j < e) || (pe != pe && pj == pj)) ? 1 : 0; }   // (NaN last)
      if (rank == kth) { out_p[k] = pe; out_d[k] = d[k * iters + e]; }
    }
  }
}

// ---------------------------------------------------------------- self test kernels
__global__ void selftest_perm_kernel(int* out) {
  int lane = threadIdx.x & 63;
  float x = (float)lane;
  int k = 0;
  out[64 * k++ + lane] = (int)lane_xor<1>(x);
  out[64 * k++ + lane] = (int)lane_xor<2>(x);
  out[64 * k++ + lane] = (int)lane_xor<4>(x);
  out[64 * k++ + lane] = (int)lane_xor<8>(x);
  out[64 * k++ + lane] = (int)lane_xor<16>(x);
  out[64 * k++ + lane] = (int)lane_mirror<2>(x, lane);
  out[64 * k++ + lane] = (int)lane_mirror<4>(x, lane);
  out[64 * k++ + lane] = (int)lane_mirror<8>(x, lane);
  out[64 * k++ + lane] = (int)lane_mirror<16>(x, lane);
  out[64 * k++ + lane] = (int)lane_mirror<32>(x, lane);
  out[64 * k++ + lane] = (int)lane_mirror<64>(x, lane);
  out[64 * k++ + lane] = (int)lane_prev(x, -1.0f);
  out[64 * k++ + lane] = (int)lane_next(x, -1.0f);
  out[64 * k++ + lane] = wave_scan_max_i32((lane * 37) % 64 == 5 ? 1000 : (lane * 7) % 23);
  out[64 * k++ + lane] = (int)wave_max_u32((unsigned)((lane * 29) % 61));
  out[64 * k++ + lane] = (int)wave_sum_u64((unsigned long long)lane * 3ull + (1ull << 33)) ;
  out[64 * k++ + lane] = (int)(wave_sum_u64((unsigned long long)lane * 3ull + (1ull << 33)) >> 32);
  out[64 * k++ + lane] = (int)wave_sum_f64(0.5 * lane);
}

template <int R>
__global__ void selftest_sort_kernel(const float* in, float* out, int* runs) {
  int lane = threadIdx.x & 63;
  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  float x[R];
#pragma unroll
  for (int r = 0; r < R; ++r) x[r] = in[r * 64 + lane];
  wave_sort<R>(x, sel, lane);
  store_sorted<R>(out, x, lane);
  unsigned pp;
  seg_runs_and_ties<R, 64, 1>(runs + lane * R, x, lane, false, pp);
}

template <int R>
static int selftest_sort(int code) {
  const int N = 64 * R;
  std::vector<float> h(N), sorted(N), got(N);
  std::vector<int> runs(N);
  uint32_t s = 12345u + R;
  for (int i = 0; i < N; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)((int)((s >> 8) % 97) - 48) * 0.25f; }   // many ties
  sorted = h; std::sort(sorted.begin(), sorted.end());
  float *din, *dout; int* druns;
  NMOD_HIP(hipMalloc(&din, N * 4)); NMOD_HIP(hipMalloc(&dout, N * 4)); NMOD_HIP(hipMalloc(&druns, N * 4));
  NMOD_HIP(hipMemcpy(din, h.data(), N * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(selftest_sort_kernel<R>, dim3(1), dim3(64), 0, 0, din, dout, druns);
  NMOD_HIP(hipMemcpy(got.data(), dout, N * 4, hipMemcpyDeviceToHost));
  NMOD_HIP(hipMemcpy(runs.data(), druns, N * 4, hipMemcpyDeviceToHost));
  hipFree(din); hipFree(dout); hipFree(druns);
  for (int i = 0; i < N; ++i) if (got[i] != sorted[i]) return code;
  for (int i = 0; i < N; ++i) {
    int st = i, en = i + 1;
    while (st > 0 && sorted[st - 1] == sorted[i]) --st;
    while (en < N && sorted[en] == sorted[i]) ++en;
    if (runs[i] != (st | (en << 16))) return code + 1;
  }
  return NMOD_OK;
}

}  // namespace nmod

// =================================================================== C ABI
using namespace nmod;

namespace nmod {
// ---------------------------------------------------------------- RCCL, bound at run time (nmod_comm_*, nmod_allgather_tracks)
// No link-time dependency: a PyTorch process already holds torch's librccl.so (its "nccl" backend) and a second copy beside it
// would be a second set of communicator state; a process without one takes librccl.so.1 from the loader path.
struct IdByValue { char internal[NMOD_COMM_ID_BYTES]; };      // ncclUniqueId is passed by value
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, IdByValue, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
thread_local int g_last_rccl = 0;
thread_local const char* g_last_rccl_text = nullptr;
static RcclApi* rccl_api() {
  static std::once_flag once;
  static RcclApi api;
  static bool ok = false;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) { api.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (api.handle) break; }     // a copy the process already holds
    if (!api.handle) for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (api.handle) break; }
    if (!api.handle) return;
    auto sym = [&](const char* n) { return dlsym(api.handle, n); };
    api.GetUniqueId = (int (*)(void*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(void**, int, IdByValue, int))sym("ncclCommInitRank");
    api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))sym("ncclAllGather");
    api.GroupStart = (int (*)())sym("ncclGroupStart");
    api.GroupEnd = (int (*)())sym("ncclGroupEnd");
    api.CommDestroy = (int (*)(void*))sym("ncclCommDestroy");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    ok = api.GetUniqueId && api.CommInitRank && api.AllGather && api.GroupStart && api.GroupEnd && api.CommDestroy;
  });
  return ok ? &api : nullptr;
}
#define NMOD_RCCL(api, call)                                                                         \
  do {                                                                                               \
    const int r_ = (call);                                                                           \
    if (r_ != 0) { g_last_rccl = r_; g_last_rccl_text = (api)->GetErrorString ? (api)->GetErrorString(r_) : nullptr; return NMOD_ERR_RCCL; } \
  } while (0)
constexpr int kNcclFloat64 = 8;                  // ncclDataType_t: ncclFloat64 / ncclDouble

}  // namespace nmod
struct nmod_comm { void* comm; int nranks, rank, device; };

extern "C" {

int nmod_abi_version(void) { return NMOD_ABI_VERSION; }

int nmod_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

const char* nmod_strerror(int rc) {
  switch (rc) {
    case NMOD_OK: return "ok";
    case NMOD_ERR_INVALID_ARG: return "invalid argument";
    case NMOD_ERR_HIP:
      snprintf(g_errbuf, sizeof(g_errbuf), "HIP runtime error: %s", hipGetErrorString(g_last_hip));
      return g_errbuf;
    case NMOD_ERR_TOO_LARGE: return "a position has more samples in a group than NMOD_MAX_RANKED (65535)";
    case NMOD_ERR_WORKSPACE: return "workspace missing or smaller than nmod_workspace_bytes()";
    case NMOD_ERR_NO_DEVICE: return "no HIP device";
    case NMOD_ERR_NO_RCCL: return "librccl.so could not be bound (neither loaded in this process nor on the loader path)";
    case NMOD_ERR_RCCL:
      snprintf(g_errbuf, sizeof(g_errbuf), "RCCL error %d: %s", g_last_rccl, g_last_rccl_text ? g_last_rccl_text : "?");
      return g_errbuf;
    default: return "unknown error code";
  }
}

int64_t nmod_workspace_bytes(const nmod_params* prm, int64_t npos) {
  (void)prm;
  if (npos < 0) return 0;
  return carve(nullptr, npos).bytes;
}

int nmod_detect_batch(const nmod_params* prm, int64_t npos, const void* sig0, const int64_t* off0,
                      const void* sig1, const int64_t* off1, const int32_t* run_id, void* workspace,
                      int64_t workspace_bytes, nmod_out* out) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (npos < 0) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  if (prm->memspace == NMOD_MEM_HOST) return detect_host_pipelined(prm, npos, sig0, off0, sig1, off1, run_id, out);
  if (prm->dtype == NMOD_DTYPE_F64) return detect_f64(prm, npos, sig0, off0, sig1, off1, run_id, workspace, workspace_bytes, out);
  return detect_device(prm, npos, sig0, off0, sig1, off1, run_id, workspace, workspace_bytes, out);
}

int nmod_combine_track(const nmod_params* prm, int64_t npos, const double* ks_d, const double* ks_p,
                       const int32_t* run_id, double* comb_st, double* comb_p) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (npos < 0 || prm->method == NMOD_METHOD_KS) return NMOD_ERR_INVALID_ARG;
  if (npos == 0) return NMOD_OK;
  if (!ks_p || !comb_st || !comb_p || (prm->nb == 0 && !ks_d) || (prm->nb > 0 && !run_id)) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  hipStream_t stream = (hipStream_t)prm->stream;
  if (prm->memspace == NMOD_MEM_DEVICE)
    return launch_combine(prm, stream, npos, ks_d, ks_p, run_id, comb_st, comb_p);
  DevBuf d_d, d_p, d_r, d_o;
  NMOD_HIP(d_d.alloc(npos * 8)); NMOD_HIP(d_p.alloc(npos * 8)); NMOD_HIP(d_r.alloc(npos * 4)); NMOD_HIP(d_o.alloc(npos * 16));
  if (ks_d) NMOD_HIP(hipMemcpyAsync(d_d.p, ks_d, npos * 8, hipMemcpyHostToDevice, stream));
  NMOD_HIP(hipMemcpyAsync(d_p.p, ks_p, npos * 8, hipMemcpyHostToDevice, stream));
  if (run_id) NMOD_HIP(hipMemcpyAsync(d_r.p, run_id, npos * 4, hipMemcpyHostToDevice, stream));
  double* o = (double*)d_o.p;
  rc = launch_combine(prm, stream, npos, (const double*)d_d.p, (const double*)d_p.p, (const int32_t*)d_r.p, o, o + npos);
  if (rc != NMOD_OK) { hipStreamSynchronize(stream); return rc; }
  NMOD_HIP(hipMemcpyAsync(comb_st, o, npos * 8, hipMemcpyDeviceToHost, stream));
  NMOD_HIP(hipMemcpyAsync(comb_p, o + npos, npos * 8, hipMemcpyDeviceToHost, stream));
  NMOD_HIP(hipStreamSynchronize(stream));
  return NMOD_OK;
}

int nmod_synth_fill(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos, int32_t group,
                    int32_t n_per_pos, int64_t plant_period, float plant_shift, void* sig_out) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (npos < 0 || n_per_pos <= 0 || !sig_out || (group != 0 && group != 1)) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_DEVICE) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  if (npos == 0) return NMOD_OK;
  hipStream_t stream = (hipStream_t)prm->stream;
  SynthArgs sa;
  sa.seed = seed; sa.pos_begin = pos_begin; sa.npos = npos; sa.group = group; sa.n_per_pos = n_per_pos;
  sa.plant_period = plant_period; sa.plant_shift = plant_shift; sa.dtype = prm->dtype; sa.out = sig_out;
  int64_t total = npos * (int64_t)n_per_pos;
  unsigned blocks = (unsigned)std::min<int64_t>((total + 255) / 256, 256 * 32);
  ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_SYNTH, stream);
  hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, stream, sa);
  NMOD_HIP(hipGetLastError());
  return NMOD_OK;
}

int nmod_synth_fill_csr(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos, int32_t group,
                        const int64_t* off, int64_t plant_period, float plant_shift, void* sig_out) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (npos < 0 || !off || !sig_out || (group != 0 && group != 1)) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_DEVICE || (prm->dtype != NMOD_DTYPE_F32 && prm->dtype != NMOD_DTYPE_I16_MILLI)) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  if (npos == 0) return NMOD_OK;
  hipStream_t stream = (hipStream_t)prm->stream;
  SynthCsrArgs sa;
  sa.seed = seed; sa.pos_begin = pos_begin; sa.npos = npos; sa.group = group; sa.dtype = prm->dtype;
  sa.plant_period = plant_period; sa.plant_shift = plant_shift; sa.off = off; sa.out = sig_out;
  unsigned blocks = (unsigned)std::min<int64_t>((npos + 3) / 4, 256 * 32);
  ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_SYNTH, stream);
  hipLaunchKernelGGL(synth_csr_kernel, dim3(blocks), dim3(256), 0, stream, sa);
  NMOD_HIP(hipGetLastError());
  return NMOD_OK;
}

int nmod_synth_fill_events(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos, int32_t group,
                           int32_t n_per_pos, const int64_t* off, int64_t plant_period, int32_t plant_shift_milli,
                           int32_t spread_milli, int32_t outlier_permille, void* sig_out) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (npos < 0 || !sig_out || (group != 0 && group != 1) || n_per_pos < 0 || (n_per_pos == 0 && !off)) return NMOD_ERR_INVALID_ARG;
  if (spread_milli < 0 || spread_milli > 8000 || plant_shift_milli < -16000 || plant_shift_milli > 16000 || outlier_permille < 0 || outlier_permille > 1000) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_DEVICE || (prm->dtype != NMOD_DTYPE_F32 && prm->dtype != NMOD_DTYPE_I16_MILLI)) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  if (npos == 0) return NMOD_OK;
  hipStream_t stream = (hipStream_t)prm->stream;
  SynthEventArgs sa;
  sa.seed = seed; sa.pos_begin = pos_begin; sa.npos = npos; sa.group = group; sa.n_per_pos = n_per_pos; sa.off = off;
  sa.plant_period = plant_period; sa.plant_shift_milli = plant_shift_milli; sa.spread_milli = spread_milli; sa.dtype = prm->dtype; sa.outlier_permille = outlier_permille; sa.out = sig_out;
  unsigned blocks = (unsigned)std::min<int64_t>((npos + 3) / 4, 256 * 32);
  ScopedKernelTimer tm(prm->timer, NMOD_KERNEL_SYNTH, stream);
  hipLaunchKernelGGL(synth_event_kernel, dim3(blocks), dim3(256), 0, stream, sa);
  NMOD_HIP(hipGetLastError());
  return NMOD_OK;
}

int nmod_describe_dispatch(const nmod_params* prm, int64_t n0, int64_t n1, char* buf, int32_t buflen) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (!buf || buflen < 8 || n0 <= 0 || n1 <= 0) return NMOD_ERR_INVALID_ARG;
  if (std::max(n0, n1) > NMOD_MAX_RANKED) return NMOD_ERR_TOO_LARGE;
  const bool want_comb = prm->method != NMOD_METHOD_KS;
  int tests = prm->tests | (want_comb ? NMOD_TEST_KS : 0);
  const bool all = (tests & (NMOD_TEST_MWU | NMOD_TEST_WELCH)) != 0 || prm->want_mstd;
  // (NMOD_DTYPE_F64 runs on per-position float32 keys: the float32 instances)
  const char* dt = prm->dtype == NMOD_DTYPE_I16_MILLI ? "i16" : "f32";
  const int c0 = size_class_of(n0), c1 = size_class_of(n1);
  // the same decisions classify_kernel / detect_device take (NMOD_DTYPE_F64: the narrower dtype is a property of the data)
  const bool big = all ? (c0 >= kNumSizeClasses || c1 >= kNumSizeClasses) : (std::min(c0, c1) >= kNumSizeClasses);
  // (all tests, smaller group <= 1 024, larger <= 4 095, not the classes of eight or four positions per wave: the counting form for
  // any coverage when the device-side probe finds the class event-like — a property of the data; the sorting form takes the rest)
  const char* kd = prm->dtype == NMOD_DTYPE_F64 ? "f32 keys" : dt;
  if (big && all && std::min(n0, n1) <= 256 && std::max(n0, n1) <= kWideBigMaxQ) {
    snprintf(buf, buflen, "rank_count_wide_kernel<%s> (event-like rows) | rank_hist_kernel<%d,64,%s,wide>", kd, 1 << std::min(c0, c1), dt); return NMOD_OK;
  }
  if (big && all && std::min(n0, n1) <= kBigHistMaxS && std::max(n0, n1) <= kBigHistMaxQ) { snprintf(buf, buflen, "big_hist_kernel<%s>", dt); return NMOD_OK; }
  if (big) { snprintf(buf, buflen, "big_rank_kernel<%s>", dt); return NMOD_OK; }
  if (!all) {
    const int cs = std::min(c0, c1);
    const int LG = ksonly_lanes_per_group(cs), R = (64 << cs) / LG;
    // (the larger group of at least kCwKsMinQ samples, the smaller one of at most 1 024: the counting form when the probe finds the class event-like)
    if (cs == 5 && std::max(n0, n1) <= 4095)
      snprintf(buf, buflen, "rank_count_value_kernel<%s,ks> (event-like rows) | ks_rank_kernel<%d,%d,%s>", prm->dtype == NMOD_DTYPE_F64 ? "f32 keys" : dt, R, LG, dt);
    else if (cs <= 4 && std::max(n0, n1) >= kCwKsMinQ && std::max(n0, n1) <= 4095)
      snprintf(buf, buflen, "rank_count_wide_kernel<%s,ks> (event-like rows) | ks_rank_kernel<%d,%d,%s>", prm->dtype == NMOD_DTYPE_F64 ? "f32 keys" : dt, R, LG, dt);
    else snprintf(buf, buflen, "ks_rank_kernel<%d,%d,%s>", R, LG, dt);
    return NMOD_OK;
  }
  const int cls = launch_class_of(c0, c1);
  if (cls >= kNumGeneralClasses) {
    const int cm = cls - kNumGeneralClasses;
    const int LG = packed_lanes_per_group(cm), R = (64 << cm) / LG;
    // capacity-256 positions: the counting form when the device-side probe finds the batch event-like (a property of the data)
    if (cm == 2 && prm->dtype != NMOD_DTYPE_F64) snprintf(buf, buflen, "rank_count_kernel<%s> (event-like rows) | rank_hist_kernel<%d,%d,%s>", dt, R, LG, dt);
    else if (cm == 2) snprintf(buf, buflen, "rank_count_kernel<f32 keys> (event-like rows) | rank_hist_kernel<%d,%d,%s>", R, LG, dt);
    else if (count_wide_rs_index(cls) >= 0) snprintf(buf, buflen, "rank_count_wide_kernel<%s> (event-like rows) | rank_hist_kernel<%d,%d,%s>", kd, R, LG, dt);
    else snprintf(buf, buflen, "rank_hist_kernel<%d,%d,%s>", R, LG, dt);
  } else if (wide_class(cls)) {
    if (count_wide_rs_index(cls) >= 0) snprintf(buf, buflen, "rank_count_wide_kernel<%s> (event-like rows) | rank_hist_kernel<%d,64,%s,wide>", kd, 1 << std::min(c0, c1), dt);
    else snprintf(buf, buflen, "rank_hist_kernel<%d,64,%s,wide>", 1 << std::min(c0, c1), dt);
  } else {
    if (count_wide_rs_index(cls) == 5) snprintf(buf, buflen, "rank_count_value_kernel<%s> (event-like rows) | rank_pair_kernel<%d,%d,%s>", kd, 1 << c0, 1 << c1, dt);
    else if (count_wide_rs_index(cls) >= 0) snprintf(buf, buflen, "rank_count_wide_kernel<%s> (event-like rows) | rank_pair_kernel<%d,%d,%s>", kd, 1 << c0, 1 << c1, dt);
    else snprintf(buf, buflen, "rank_pair_kernel<%d,%d,%s>", 1 << c0, 1 << c1, dt);
  }
  return NMOD_OK;
}

int nmod_host_pipeline_config(int64_t chunk_bytes, int32_t slots, int32_t threads, int32_t mode) {
  if (chunk_bytes < 0 || slots < 0 || slots > kHpMaxSlots || threads < 0 || threads > 256 || mode < 0 || mode > 2) return NMOD_ERR_INVALID_ARG;
  g_hp_chunk_bytes.store(chunk_bytes); g_hp_slots.store(slots); g_hp_threads.store(threads); g_hp_mode.store(mode);
  return NMOD_OK;
}

int nmod_narrow_probe(const double* v, int64_t n, int16_t* out) {
  if (!v || !out || n < 0) return NMOD_ERR_INVALID_ARG;
  return narrow_f64_to_i16(v, out, (size_t)n) ? 1 : 0;
}

int nmod_last_host_stats(nmod_host_stats* st) {
  if (!st) return NMOD_ERR_INVALID_ARG;
  *st = g_host_stats;
  return NMOD_OK;
}


static void assemble_dispatch_stats(const unsigned long long* raw, int64_t npos, nmod_dispatch_stats* st) {
  memset(st, 0, sizeof(*st));
  st->positions = npos;
  constexpr int c256 = kNumGeneralClasses + 2;
  int64_t placed = 0;
  for (int c = 0; c < kNumPairs; ++c) {
    const int64_t n = (int64_t)raw[kStatsClass + c];
    if (n == 0) continue;
    placed += n;
    int64_t* sorting = c >= kWideBigBase ? &st->rank_hist_wide
                     : (c == kBigClass || c == kBigHistClass) ? &st->big
                     : c >= kKsClassBase ? &st->ks_rank
                     : c >= kNumGeneralClasses ? &st->rank_hist
                     : wide_class(c) ? &st->rank_hist_wide : &st->rank_pair;
    const int64_t tried = (int64_t)raw[kStatsTried + c];
    const int64_t counted = c == c256 ? (int64_t)raw[1] : tried - (int64_t)raw[kStatsLeft + c];
    if (c == c256) st->rank_count += counted; else st->rank_count_wide += counted;
    *sorting += n - counted;
    st->count_tried += tried;
    st->count_rejected += tried - counted;
  }
  st->skipped = npos - placed;
  st->f64_redo = (int64_t)raw[2];
}

int nmod_last_dispatch_stats(nmod_dispatch_stats* st) {
  if (!st) return NMOD_ERR_INVALID_ARG;
  DispatchRec& d = g_dispatch;
  if (!d.valid) { memset(st, 0, sizeof(*st)); return NMOD_ERR_INVALID_ARG; }
  if (d.host) { assemble_dispatch_stats(d.host_totals, d.npos, st); return NMOD_OK; }
  unsigned long long raw[kStatsWords];
  NMOD_HIP(hipSetDevice(d.device));
  NMOD_HIP(hipMemsetAsync(d.args.acc, 0, kStatsWords * 8, d.stream));
  NMOD_HIP(enqueue_dispatch_stats(d.args, d.stream));
  NMOD_HIP(hipMemcpyAsync(raw, d.args.acc, kStatsWords * 8, hipMemcpyDeviceToHost, d.stream));
  NMOD_HIP(hipStreamSynchronize(d.stream));
  assemble_dispatch_stats(raw, d.npos, st);
  return NMOD_OK;
}

const char* nmod_build_info(void) {
  static std::once_flag once;
  static std::string info;
  std::call_once(once, [] {
    char head[96];
    snprintf(head, sizeof(head), "arch=gfx950 abi=%d hip=%d.%d", NMOD_ABI_VERSION, HIP_VERSION_MAJOR, HIP_VERSION_MINOR);
    info = head;
    info += " | abi_tu: " NMOD_BUILD_FLAGS;
    info += std::string(" | k1_f32_ks: ") + rank_stats_build_flags_d0_a0();
    info += std::string(" | k1_f32_all: ") + rank_stats_build_flags_d0_a1();
    info += std::string(" | k1_i16_ks: ") + rank_stats_build_flags_d1_a0();
    info += std::string(" | k1_i16_all: ") + rank_stats_build_flags_d1_a1();
  });
  return info.c_str();
}

int nmod_downsample_ks(const nmod_params* prm, int64_t nflag, const void* sig0, const int64_t* off0, const void* sig1, const int64_t* off1,
                       const int64_t* positions, const int64_t* cov, int32_t iters, double quantile, uint64_t seed,
                       double* ks_d, double* ks_p) {
  int rc = check_params(prm);
  if (rc != NMOD_OK) return rc;
  if (nflag < 0 || iters <= 0 || iters > 65536 || !(quantile >= 0.0 && quantile < 1.0)) return NMOD_ERR_INVALID_ARG;
  if (prm->memspace != NMOD_MEM_HOST) return NMOD_ERR_INVALID_ARG;                 // (the caller's arrays are host memory: mtest2's batch)
  if (nflag == 0) return NMOD_OK;
  if (!sig0 || !sig1 || !off0 || !off1 || !positions || !cov || !ks_d || !ks_p) return NMOD_ERR_INVALID_ARG;
  if (nmod_device_count() <= prm->device || prm->device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(prm->device));
  hipStream_t stream = (hipStream_t)prm->stream;
  const int esz = prm->dtype == NMOD_DTYPE_F32 ? 4 : (prm->dtype == NMOD_DTYPE_F64 ? 8 : 2);
  const int kth = (int)((double)iters * quantile);
  int num_cus = 0;
  NMOD_HIP(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, prm->device));
  const int64_t max_elements = env_i64("NMOD_DOWNSAMPLE_ELEMENTS", (int64_t)1 << 27);
  std::vector<int32_t> rn[2], rsz[2];
  std::vector<int64_t> roff[2], vbase[2];
  std::vector<char> rows[2];
  const void* sig[2] = {sig0, sig1};
  const int64_t* off[2] = {off0, off1};
  int64_t lo = 0;
  while (lo < nflag) {
    // ---- the chunk: as many flagged positions as fit max_elements virtual samples (at least one)
    int64_t hi = lo, tot = 0, maxsz[2] = {1, 1};
    for (int g = 0; g < 2; ++g) { rn[g].clear(); rsz[g].clear(); roff[g].clear(); vbase[g].clear(); rows[g].clear(); }
    int64_t at[2] = {0, 0}, vat[2] = {0, 0};
    while (hi < nflag) {
      const int64_t p = positions[hi];
      int64_t n[2], z[2];
      for (int g = 0; g < 2; ++g) {
        n[g] = off[g][p + 1] - off[g][p];
        if (n[g] <= 0 || n[g] > NMOD_MAX_RANKED) return n[g] > NMOD_MAX_RANKED ? NMOD_ERR_TOO_LARGE : NMOD_ERR_INVALID_ARG;
        z[g] = (cov[hi] > 0 && n[g] > cov[hi]) ? cov[hi] : n[g];
      }
      const int64_t add = (z[0] + z[1]) * (int64_t)iters;
      if (hi > lo && tot + add > max_elements) break;
      tot += add;
      for (int g = 0; g < 2; ++g) {
        rn[g].push_back((int32_t)n[g]); rsz[g].push_back((int32_t)z[g]); roff[g].push_back(at[g]); vbase[g].push_back(vat[g]);
        const char* src = (const char*)sig[g] + off[g][p] * esz;
        rows[g].insert(rows[g].end(), src, src + n[g] * esz);
        at[g] += n[g]; vat[g] += z[g] * (int64_t)iters; maxsz[g] = std::max(maxsz[g], z[g]);
      }
      ++hi;
    }
    const int64_t nrows = hi - lo, nv = nrows * (int64_t)iters;
    if (nv > INT32_MAX) return NMOD_ERR_INVALID_ARG;
    // every exit below this point — an allocation or launch failing included — first waits for the stream: the asynchronous copies
    // read the host vectors of this scope (rows / roff / vbase / rn / rsz)
    struct StreamDrain { hipStream_t s; ~StreamDrain() { (void)hipStreamSynchronize(s); } } drain{stream};
    // ---- device buffers of the chunk (stream-ordered, from the library's pool)
    nmod_params dp = *prm;
    dp.memspace = NMOD_MEM_DEVICE; dp.tests = NMOD_TEST_KS; dp.method = NMOD_METHOD_KS; dp.want_mstd = 0; dp.flags = 0;
    dp.stride0 = 0; dp.stride1 = 0; dp.max_n0 = (int32_t)maxsz[0]; dp.max_n1 = (int32_t)maxsz[1];
    const int64_t wsb = nmod_workspace_bytes(&dp, nv);
    DevScratch d_rows[2], d_meta[2], d_virt[2], d_voff[2], d_ws, d_res;
    for (int g = 0; g < 2; ++g) {
      const size_t mbytes = (size_t)nrows * (4 + 4 + 8 + 8);
      NMOD_HIP(d_rows[g].alloc(rows[g].size(), stream, prm->device));
      NMOD_HIP(d_meta[g].alloc(mbytes, stream, prm->device));
      NMOD_HIP(d_virt[g].alloc((size_t)vat[g] * esz + 256, stream, prm->device));
      NMOD_HIP(d_voff[g].alloc((size_t)(nv + 1) * 8, stream, prm->device));
      char* m = (char*)d_meta[g].p;
      NMOD_HIP(hipMemcpyAsync(d_rows[g].p, rows[g].data(), rows[g].size(), hipMemcpyHostToDevice, stream));
      NMOD_HIP(hipMemcpyAsync(m, roff[g].data(), (size_t)nrows * 8, hipMemcpyHostToDevice, stream));
      NMOD_HIP(hipMemcpyAsync(m + nrows * 8, vbase[g].data(), (size_t)nrows * 8, hipMemcpyHostToDevice, stream));
      NMOD_HIP(hipMemcpyAsync(m + nrows * 16, rn[g].data(), (size_t)nrows * 4, hipMemcpyHostToDevice, stream));
      NMOD_HIP(hipMemcpyAsync(m + nrows * 20, rsz[g].data(), (size_t)nrows * 4, hipMemcpyHostToDevice, stream));
      ResampleArgs ra;
      ra.rows = (const char*)d_rows[g].p; ra.esz = esz; ra.roff = (const int64_t*)m; ra.vbase = (const int64_t*)(m + nrows * 8);
      ra.rn = (const int32_t*)(m + nrows * 16); ra.rsz = (const int32_t*)(m + nrows * 20);
      ra.nrows = nrows; ra.iters = iters; ra.seed = seed; ra.group = g; ra.k0 = lo; ra.out = (char*)d_virt[g].p; ra.voff = (int64_t*)d_voff[g].p;
      hipLaunchKernelGGL(resample_kernel, dim3((unsigned)std::min<int64_t>((nv + 3) / 4, (int64_t)num_cus * 32)), dim3(256), 0, stream, ra);
      NMOD_HIP(hipGetLastError());
    }
    NMOD_HIP(d_ws.alloc((size_t)wsb, stream, prm->device));
    NMOD_HIP(d_res.alloc((size_t)nv * 17 + (size_t)nrows * 16 + 64, stream, prm->device));
    double* v_d = (double*)d_res.p; double* v_p = v_d + nv; double* o_d = v_p + nv; double* o_p = o_d + nrows;
    nmod_out vout;
    memset(&vout, 0, sizeof(vout));
    vout.ks_d = v_d; vout.ks_p = v_p; vout.status = (uint8_t*)(o_p + nrows);
    // the 100 resamples of every flagged position through the same KS kernel as everything else
    if (prm->dtype == NMOD_DTYPE_F64) {
      const int64_t bounds[4] = {0, vat[0], 0, vat[1]};
      rc = detect_f64(&dp, nv, d_virt[0].p, (const int64_t*)d_voff[0].p, d_virt[1].p, (const int64_t*)d_voff[1].p, nullptr, d_ws.p, wsb, &vout, bounds);
    } else {
      rc = detect_device(&dp, nv, d_virt[0].p, (const int64_t*)d_voff[0].p, d_virt[1].p, (const int64_t*)d_voff[1].p, nullptr, d_ws.p, wsb, &vout);
    }
    if (rc != NMOD_OK) { hipStreamSynchronize(stream); return rc; }
    hipLaunchKernelGGL(select_quantile_kernel, dim3((unsigned)std::min<int64_t>((nrows + 3) / 4, (int64_t)num_cus * 32)), dim3(256), 0, stream,
                       nrows, iters, kth, v_d, v_p, o_d, o_p);
    NMOD_HIP(hipGetLastError());
    NMOD_HIP(hipMemcpyAsync(ks_d + lo, o_d, (size_t)nrows * 8, hipMemcpyDeviceToHost, stream));
    NMOD_HIP(hipMemcpyAsync(ks_p + lo, o_p, (size_t)nrows * 8, hipMemcpyDeviceToHost, stream));
    NMOD_HIP(hipStreamSynchronize(stream));          // (the host vectors are reused by the next chunk)
    lo = hi;
  }
  return NMOD_OK;
}

int nmod_trim_scratch(int32_t device) {
  if (device < 0 || device >= kMaxDevices) return NMOD_ERR_INVALID_ARG;
  hp_trim(device);                                // the pinned ring + streams of the host-resident entry
  std::lock_guard<std::mutex> lock(g_pool_mutex);
  if (g_pool[device]) NMOD_HIP(hipMemPoolTrimTo(g_pool[device], 0));
  return NMOD_OK;
}

int nmod_evtimer_create(int32_t capacity_per_kernel, void** timer) {
  if (!timer || capacity_per_kernel <= 0) return NMOD_ERR_INVALID_ARG;
  EvTimer* t = new EvTimer();
  t->capacity = capacity_per_kernel;
  for (int k = 0; k < NMOD_KERNEL_COUNT; ++k) {
    t->used[k] = 0;
    t->start[k].assign(capacity_per_kernel, nullptr); t->stop[k].assign(capacity_per_kernel, nullptr);
  }
  for (int k = 0; k < NMOD_KERNEL_COUNT; ++k) {
    for (int i = 0; i < capacity_per_kernel; ++i) {
      hipError_t e = hipEventCreate(&t->start[k][i]);
      if (e == hipSuccess) e = hipEventCreate(&t->stop[k][i]);
      if (e != hipSuccess) {                       // free what was created so far
        g_last_hip = e;
        for (int kk = 0; kk < NMOD_KERNEL_COUNT; ++kk)
          for (int ii = 0; ii < capacity_per_kernel; ++ii) {
            if (t->start[kk][ii]) hipEventDestroy(t->start[kk][ii]);
            if (t->stop[kk][ii]) hipEventDestroy(t->stop[kk][ii]);
          }
        delete t;
        return NMOD_ERR_HIP;
      }
    }
  }
  *timer = t;
  return NMOD_OK;
}

int nmod_evtimer_reset(void* timer) {
  if (!timer) return NMOD_ERR_INVALID_ARG;
  EvTimer* t = (EvTimer*)timer;
  for (int k = 0; k < NMOD_KERNEL_COUNT; ++k) t->used[k] = 0;
  return NMOD_OK;
}

int nmod_evtimer_read(void* timer, int32_t kernel, double* total_ms, int32_t* launches) {
  if (!timer || kernel < 0 || kernel >= NMOD_KERNEL_COUNT || !total_ms || !launches) return NMOD_ERR_INVALID_ARG;
  EvTimer* t = (EvTimer*)timer;
  double tot = 0.0;
  for (int i = 0; i < t->used[kernel]; ++i) {
    float ms = 0.f;
    NMOD_HIP(hipEventSynchronize(t->stop[kernel][i]));
    NMOD_HIP(hipEventElapsedTime(&ms, t->start[kernel][i], t->stop[kernel][i]));
    tot += ms;
  }
  *total_ms = tot; *launches = t->used[kernel];
  return NMOD_OK;
}

int nmod_evtimer_destroy(void* timer) {
  if (!timer) return NMOD_ERR_INVALID_ARG;
  EvTimer* t = (EvTimer*)timer;
  for (int k = 0; k < NMOD_KERNEL_COUNT; ++k)
    for (int i = 0; i < t->capacity; ++i) { hipEventDestroy(t->start[k][i]); hipEventDestroy(t->stop[k][i]); }
  delete t;
  return NMOD_OK;
}

// ---- save_test's table (myDetect.py:532-536), buffered
// '%.3f' and '%.3E' exactly as printf (and Python) round them — the exact binary value, half to even — without going
// through printf for every number (350 ns each: the table writer was the largest piece of a 4.6 M-position mtest2).
// x87 extended precision carries 64 bits: |v| * 1000 is exact in it for every double below 2^63 / 1000, so '%.3f' is
// decided exactly; '%.3E' scales by a power of ten (relative error ~2^-62) and hands every value whose fourth
// significant digit is within 1e-7 of a rounding boundary to snprintf.
#if !defined(__HIP_DEVICE_COMPILE__)                      // (host code; the device pass parses it with a 64-bit long double)
static_assert(sizeof(long double) >= 10 && __LDBL_MANT_DIG__ >= 64, "the fast formats need a 64-bit significand");
#endif
static inline char* put_digits(char* p, unsigned long long v) {
  char tmp[24]; int n = 0;
  do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (n) *p++ = tmp[--n];
  return p;
}
static inline char* put_3(char* p, unsigned v) { p[0] = (char)('0' + v / 100); p[1] = (char)('0' + v / 10 % 10); p[2] = (char)('0' + v % 10); return p + 3; }
static inline char* put_f3(char* p, double v) {          // '%.3f' as Python formats it
  if (v != v) { memcpy(p, "nan", 3); return p + 3; }
  if (v == INFINITY) { memcpy(p, "inf", 3); return p + 3; }
  if (v == -INFINITY) { memcpy(p, "-inf", 4); return p + 4; }
  const double a = fabs(v);
  if (!(a < 9.0e15)) return p + snprintf(p, 400, "%.3f", v);
  const long double x = (long double)a * 1000.0L;         // exact: 53 + 10 bits
  unsigned long long d = (unsigned long long)x;           // truncation
  const long double frac = x - (long double)d;            // exact
  if (frac > 0.5L || (frac == 0.5L && (d & 1ull))) ++d;   // half to even on the exact value
  if (signbit(v)) *p++ = '-';
  p = put_digits(p, d / 1000);
  *p++ = '.';
  return put_3(p, (unsigned)(d % 1000));
}
struct Pow10Table {                                       // 10^i, i = -360 .. 360, as long double
  long double v[721];
  Pow10Table() { for (int i = 0; i <= 720; ++i) v[i] = powl(10.0L, (long double)(i - 360)); }
  long double operator()(int i) const { return v[i + 360]; }
};
static inline char* put_e3(char* p, double v) {          // '%.3E'
  static const Pow10Table pow10;
  if (v != v) { memcpy(p, "NAN", 3); return p + 3; }      // Python upper-cases non-finite values under %E
  if (v == INFINITY) { memcpy(p, "INF", 3); return p + 3; }
  if (v == -INFINITY) { memcpy(p, "-INF", 4); return p + 4; }
  const double a = fabs(v);
  if (a == 0.0) { if (signbit(v)) *p++ = '-'; memcpy(p, "0.000E+00", 9); return p + 9; }
  int e2;
  frexp(a, &e2);                                          // a = f * 2^e2, 0.5 <= f < 1
  int k = (int)floor((e2 - 1) * 0.30102999566398120);     // floor(log10 a) or one less
  long double x = (long double)a * pow10(3 - k);          // want 1000 <= x < 10000
  if (x >= 10000.0L) { ++k; x = (long double)a * pow10(3 - k); }
  else if (x < 1000.0L) { --k; x = (long double)a * pow10(3 - k); }
  if (!(x >= 1000.0L && x < 10000.0L)) return p + snprintf(p, 64, "%.3E", v);
  unsigned d = (unsigned)x;
  const long double frac = x - (long double)d;
  // (near a boundary of the fourth digit — or of the power of ten itself — the scaled value is not trusted)
  if (fabsl(frac - 0.5L) < 1e-7L || frac < 1e-7L || frac > 1.0L - 1e-7L) return p + snprintf(p, 64, "%.3E", v);
  if (frac > 0.5L) ++d;
  if (d == 10000u) { d = 1000u; ++k; }
  if (signbit(v)) *p++ = '-';
  *p++ = (char)('0' + d / 1000); *p++ = '.';
  p = put_3(p, d % 1000);
  *p++ = 'E'; *p++ = k < 0 ? '-' : '+';
  const unsigned ak = (unsigned)(k < 0 ? -k : k);
  if (ak >= 100) { return put_3(p, ak); }
  p[0] = (char)('0' + ak / 10); p[1] = (char)('0' + ak % 10);
  return p + 2;
}

// test hook: the two formats on an array of values, NUL-separated (tests/test_abi_and_host.py compares with Python)
extern "C" int nmod_format_probe(const double* v, int64_t n, int32_t sci, char* out, int64_t cap) {
  if (!v || !out || n < 0) return NMOD_ERR_INVALID_ARG;
  char* p = out;
  for (int64_t i = 0; i < n; ++i) {
    if (p - out + 420 > cap) return NMOD_ERR_INVALID_ARG;
    p = sci ? put_e3(p, v[i]) : put_f3(p, v[i]);
    *p++ = '\0';
  }
  return NMOD_OK;
}

int nmod_write_sign_test(const char* path, int64_t npos, const int32_t* chrom_id, const char* chrom_names,
                         int32_t n_chroms, const char* strand, const int64_t* pos0, const char* base,
                         const int32_t* n0, const int32_t* n1, const double* mwu_u, const double* mwu_p,
                         const double* t_t, const double* t_p, const double* ks_d, const double* ks_p,
                         const double* comb_st, const double* comb_p, int32_t with_comb) {
  if (!path || npos < 0 || n_chroms < 0) return NMOD_ERR_INVALID_ARG;
  if (npos > 0 && (!chrom_id || !chrom_names || !strand || !pos0 || !base || !n0 || !n1 || !mwu_u || !mwu_p || !t_t ||
                   !t_p || !ks_d || !ks_p || (with_comb && (!comb_st || !comb_p)))) return NMOD_ERR_INVALID_ARG;
  std::vector<const char*> names(n_chroms);
  size_t longest = 0;
  const char* q = chrom_names;
  std::vector<size_t> name_len(n_chroms);
  for (int i = 0; i < n_chroms; ++i) { names[i] = q; name_len[i] = strlen(q); longest = std::max(longest, name_len[i]); q += name_len[i] + 1; }
  for (int64_t i = 0; i < npos; ++i) if (chrom_id[i] < 0 || chrom_id[i] >= n_chroms) return NMOD_ERR_INVALID_ARG;
  FILE* f = fopen(path, "w");
  if (!f) return NMOD_ERR_INVALID_ARG;
  // lines are formatted by several host threads, a block of positions each, and written in order
  const size_t line_cap = longest + 4096;              // '%.3f' of a value near DBL_MAX prints ~310 digits, eight of them never do
  auto format_block = [&](int64_t lo, int64_t hi, std::vector<char>& buf) {
    buf.clear();
    buf.reserve((size_t)(hi - lo) * 160);
    std::vector<char> line(line_cap);
    for (int64_t i = lo; i < hi; ++i) {
      char* p = line.data();
      {                                                       // "%s %c %lld %c %d %d "
        const char* nm = names[chrom_id[i]];
        const size_t ln = name_len[chrom_id[i]];
        memcpy(p, nm, ln); p += ln;
        *p++ = ' '; *p++ = strand[i]; *p++ = ' ';
        const long long ps = (long long)(pos0[i] + 1);
        if (ps < 0) { *p++ = '-'; p = put_digits(p, 0ull - (unsigned long long)ps); } else p = put_digits(p, (unsigned long long)ps);
        *p++ = ' '; *p++ = base[i]; *p++ = ' ';
        if (n0[i] < 0) { *p++ = '-'; p = put_digits(p, (unsigned long long)(-(long long)n0[i])); } else p = put_digits(p, (unsigned long long)n0[i]);
        *p++ = ' ';
        if (n1[i] < 0) { *p++ = '-'; p = put_digits(p, (unsigned long long)(-(long long)n1[i])); } else p = put_digits(p, (unsigned long long)n1[i]);
        *p++ = ' ';
      }
      p = put_f3(p, mwu_u[i]); *p++ = ' '; p = put_e3(p, mwu_p[i]); *p++ = ' ';
      p = put_f3(p, t_t[i]); *p++ = ' '; p = put_e3(p, t_p[i]); *p++ = ' ';
      p = put_f3(p, ks_d[i]); *p++ = ' '; p = put_e3(p, ks_p[i]);
      if (with_comb) { *p++ = ' '; p = put_f3(p, comb_st[i]); *p++ = ' '; p = put_e3(p, comb_p[i]); }
      *p++ = '\n';
      buf.insert(buf.end(), line.data(), p);
    }
  };
  const int64_t kBlock = 1 << 16;
  unsigned hw = std::thread::hardware_concurrency();
  const int nthreads = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)(hw ? hw : 1), 16, (npos + kBlock - 1) / kBlock}));
  bool ok = true;
  std::vector<std::vector<char>> bufs(nthreads);
  for (int64_t base_pos = 0; base_pos < npos && ok; base_pos += kBlock * nthreads) {
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t) {
      const int64_t lo = base_pos + (int64_t)t * kBlock, hi = std::min(npos, lo + kBlock);
      if (lo >= npos) { bufs[t].clear(); continue; }
      if (nthreads == 1) format_block(lo, hi, bufs[t]);
      else th.emplace_back(format_block, lo, hi, std::ref(bufs[t]));
    }
    for (auto& x : th) x.join();
    for (int t = 0; t < nthreads && ok; ++t)
      if (!bufs[t].empty()) ok = fwrite(bufs[t].data(), 1, bufs[t].size(), f) == bufs[t].size();
  }
  const bool closed = fclose(f) == 0;
  return (ok && closed) ? NMOD_OK : NMOD_ERR_INVALID_ARG;
}

int nmod_comm_unique_id(void* id_out) {
  if (!id_out) return NMOD_ERR_INVALID_ARG;
  RcclApi* api = rccl_api();
  if (!api) return NMOD_ERR_NO_RCCL;
  NMOD_RCCL(api, api->GetUniqueId(id_out));
  return NMOD_OK;
}

int nmod_comm_init_rank(const void* unique_id, int32_t nranks, int32_t rank, int32_t device, nmod_comm** comm_out) {
  if (!unique_id || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) return NMOD_ERR_INVALID_ARG;
  *comm_out = nullptr;
  if (nmod_device_count() <= device || device < 0) return NMOD_ERR_NO_DEVICE;
  RcclApi* api = rccl_api();
  if (!api) return NMOD_ERR_NO_RCCL;
  NMOD_HIP(hipSetDevice(device));
  IdByValue id;
  memcpy(id.internal, unique_id, NMOD_COMM_ID_BYTES);
  void* c = nullptr;
  NMOD_RCCL(api, api->CommInitRank(&c, nranks, id, rank));
  nmod_comm* h = new nmod_comm();
  h->comm = c; h->nranks = nranks; h->rank = rank; h->device = device;
  *comm_out = h;
  return NMOD_OK;
}

int nmod_allgather_tracks(nmod_comm* comm, void* stream, int64_t block_len, int32_t ntracks, const double* const* local, double* const* full) {
  if (!comm || !comm->comm || block_len < 0 || ntracks < 0 || ntracks > 64 || (ntracks > 0 && (!local || !full))) return NMOD_ERR_INVALID_ARG;
  for (int t = 0; t < ntracks; ++t) if (!local[t] || !full[t]) return NMOD_ERR_INVALID_ARG;
  if (block_len == 0 || ntracks == 0) return NMOD_OK;
  RcclApi* api = rccl_api();
  if (!api) return NMOD_ERR_NO_RCCL;
  NMOD_HIP(hipSetDevice(comm->device));
  // one group: the tracks' gathers are fused into one launch on the caller's stream; nothing is synchronised here
  NMOD_RCCL(api, api->GroupStart());
  int first_bad = 0;
  for (int t = 0; t < ntracks; ++t) {
    const int r = api->AllGather(local[t], full[t], (size_t)block_len, kNcclFloat64, comm->comm, (hipStream_t)stream);
    if (r != 0 && first_bad == 0) first_bad = r;
  }
  const int rend = api->GroupEnd();
  if (first_bad != 0 || rend != 0) { g_last_rccl = first_bad ? first_bad : rend; g_last_rccl_text = api->GetErrorString ? api->GetErrorString(g_last_rccl) : nullptr; return NMOD_ERR_RCCL; }
  return NMOD_OK;
}

int nmod_comm_destroy(nmod_comm* comm) {
  if (!comm) return NMOD_ERR_INVALID_ARG;
  RcclApi* api = rccl_api();
  int rc = NMOD_OK;
  if (api && comm->comm) { const int r = api->CommDestroy(comm->comm); if (r != 0) { g_last_rccl = r; g_last_rccl_text = api->GetErrorString ? api->GetErrorString(r) : nullptr; rc = NMOD_ERR_RCCL; } }
  delete comm;
  return rc;
}

int nmod_selftest(int32_t device) {
  if (nmod_device_count() <= device || device < 0) return NMOD_ERR_NO_DEVICE;
  NMOD_HIP(hipSetDevice(device));
  const int NP = 18;
  int* d; std::vector<int> h(64 * NP);
  NMOD_HIP(hipMalloc(&d, 64 * NP * 4));
  hipLaunchKernelGGL(selftest_perm_kernel, dim3(1), dim3(64), 0, 0, d);
  NMOD_HIP(hipMemcpy(h.data(), d, 64 * NP * 4, hipMemcpyDeviceToHost));
  hipFree(d);
  int scan_in[64], run = 0, mx = 0;
  for (int l = 0; l < 64; ++l) { scan_in[l] = (l * 37) % 64 == 5 ? 1000 : (l * 7) % 23; mx = std::max(mx, (l * 29) % 61); }
  unsigned long long tot = 0; double ftot = 0;
  for (int l = 0; l < 64; ++l) { tot += (unsigned long long)l * 3ull + (1ull << 33); ftot += 0.5 * l; }
  for (int l = 0; l < 64; ++l) {
    run = std::max(run, scan_in[l]);
    const int expect[NP] = {l ^ 1, l ^ 2, l ^ 4, l ^ 8, l ^ 16, l ^ 1, l ^ 3, l ^ 7, l ^ 15, l ^ 31, l ^ 63,
                            l == 0 ? -1 : l - 1, l == 63 ? -1 : l + 1, run, mx, (int)(unsigned)tot, (int)(tot >> 32), (int)ftot};
    for (int k = 0; k < NP; ++k) if (h[64 * k + l] != expect[k]) return 1 + k;
  }
  int rc;
  if ((rc = selftest_sort<1>(100)) != NMOD_OK) return rc;
  if ((rc = selftest_sort<2>(110)) != NMOD_OK) return rc;
  if ((rc = selftest_sort<4>(120)) != NMOD_OK) return rc;
  if ((rc = selftest_sort<8>(130)) != NMOD_OK) return rc;
  if ((rc = selftest_sort<16>(140)) != NMOD_OK) return rc;
  if ((rc = selftest_sort<32>(150)) != NMOD_OK) return rc;
  return NMOD_OK;
}

}  // extern "C"
