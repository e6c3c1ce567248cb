// K1 — per-position rank statistics, one wavefront (64 lanes) per genomic position.
//
// Replaces the data-dependent part of getKStest (myDetect.py:327-343): the
// sorts / searchsorted of ks_2samp, the rankdata + tiecorrect of mannwhitneyu
// and the mean / var reductions of ttest_ind.  It emits exact integers
//     ks_num = max_v |c0(v)*n1 - c1(v)*n0|,  c = #{x <= v}           (KS D numerator)
//     mwu_s  = sum_{a in group 1} (#{b < a} + #{b <= a})             (U1 = n0*n1 - mwu_s/2)
//     tie    = sum_{pooled tie groups} (t^3 - t)                      (tiecorrect)
// and fp64 (mean, M2) per group; the p-values are a separate kernel (K2).
//
// Algorithm, per wave:
//   1. coalesced dword loads: lane l takes samples l, l+64, ... into R registers
//      per group (R = 1..32 -> up to 64*R samples, padded with +inf);
//   2. each group is sorted in registers by a bitonic network in its "mirror"
//      form (every merge ascending).  Compare-exchanges between registers of
//      one lane are v_min/v_max; between lanes they are one DPP move (or
//      ds_swizzle / ds_bpermute beyond a 16-lane row) + one v_med3_f32 whose
//      third operand is -inf (keep min) or +inf (keep max) per lane;
//   3. both sorted groups go to wave-private LDS; each lane finds its merge-path
//      split by binary search and merges ceil((n0+n1)/64) pooled elements
//      sequentially, carrying the exact counts (c0, c1) — so the KS numerator,
//      the rank sums and the tie term come out of one pass, ties included.
// No MFMA: this is sort / scan / reduction work (BASELINE.json).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.hpp"

namespace nmod {

constexpr int kWavesPerBlock = 4;
constexpr int kLdsPad = 4;   // +inf sentinels after each sorted group
constexpr int kClassStride = 48;   // class_meta: [c] count, [kClassStride + c] offset, [2*kClassStride + c] cursor

struct RankStatsArgs {
  const void* sig0; const void* sig1;
  const int64_t* off0; const int64_t* off1;    // may be null with stride > 0
  int64_t stride0, stride1;
  int64_t npos;
  const int32_t* pos_list;                     // ragged batches: positions grouped by size class (null: all of [0, npos))
  const int32_t* class_meta;                   // device ints: [c] = count of class c, [kClassStride + c] = its offset into pos_list
  int32_t class_id;                            // size class of this launch (rank_stats_launch.hpp)
  uint32_t* ks_num;                            // [npos]
  uint64_t* mwu_s;                             // [npos]  (MWU)
  uint64_t* tie;                               // [npos]  (MWU)
  double* moments;                             // [npos][4] mean0, M2_0, mean1, M2_1 (WELCH)
  double* ks_d_ref;                            // [npos] max |fl(c0/n0) - fl(c1/n1)| exactly as ks_2samp forms it (all-tests mode)
};

__device__ __forceinline__ void ce(float& lo, float& hi) {
  float a = fminf(lo, hi), b = fmaxf(lo, hi);
  lo = a; hi = b;
}

// full ascending sort of the R registers of one lane: Batcher's odd-even merge sort
// (191 compare-exchanges for 32 keys against 240 for the bitonic network), indices all static
template <int LO, int N, int RR, int R>
__device__ __forceinline__ void oe_merge(float (&x)[R]) {
  constexpr int M = RR * 2;
  if constexpr (M < N) {
    oe_merge<LO, N, M, R>(x);
    oe_merge<LO + RR, N, M, R>(x);
#pragma unroll
    for (int i = LO + RR; i + RR < LO + N; i += M) ce(x[i], x[i + RR]);
  } else {
    ce(x[LO], x[LO + RR]);
  }
}
template <int LO, int N, int R>
__device__ __forceinline__ void oe_sort(float (&x)[R]) {
  if constexpr (N > 1) {
    oe_sort<LO, N / 2, R>(x);
    oe_sort<LO + N / 2, N / 2, R>(x);
    oe_merge<LO, N, 1, R>(x);
  }
}
template <int R>
__device__ __forceinline__ void sort_in_lane(float (&x)[R]) { oe_sort<0, R, R>(x); }

// half-cleaners between registers of one lane: distances R/2 .. 1
template <int R>
__device__ __forceinline__ void clean_in_lane(float (&x)[R]) {
#pragma unroll
  for (int j = R >> 1; j >= 1; j >>= 1) {
#pragma unroll
    for (int i = 0; i < R; ++i)
      if ((i & j) == 0) ce(x[i], x[i | j]);
  }
}

// sel[b] = (lane bit b set) ? +inf : -inf ; med3(x, y, -inf) = min, med3(x, y, +inf) = max
struct LaneSel { float s[6]; };

template <int R, int M, int SELBIT>
__device__ __forceinline__ void xor_stage(float (&x)[R], const LaneSel& sel) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float y = lane_xor<M>(x[r]);
    x[r] = __builtin_amdgcn_fmed3f(x[r], y, sel.s[SELBIT]);
  }
}

// merge sorted runs of (G/2)*R elements into runs of G*R elements; blocked layout e = lane*R + r
template <int R, int G>
__device__ __forceinline__ void merge_lanes(float (&x)[R], const LaneSel& sel, int lane) {
  constexpr int TOPBIT = (G == 2) ? 0 : (G == 4) ? 1 : (G == 8) ? 2 : (G == 16) ? 3 : (G == 32) ? 4 : 5;
  // mirror stage: partner of (lane, r) is (lane ^ (G-1), R-1-r)
  if constexpr (R == 1) {
    float y = lane_mirror<G>(x[0], lane);
    x[0] = __builtin_amdgcn_fmed3f(x[0], y, sel.s[TOPBIT]);
  } else {
#pragma unroll
    for (int r = 0; r < R / 2; ++r) {
      float y0 = lane_mirror<G>(x[R - 1 - r], lane);
      float y1 = lane_mirror<G>(x[r], lane);
      x[r] = __builtin_amdgcn_fmed3f(x[r], y0, sel.s[TOPBIT]);
      x[R - 1 - r] = __builtin_amdgcn_fmed3f(x[R - 1 - r], y1, sel.s[TOPBIT]);
    }
  }
  if constexpr (G >= 64) xor_stage<R, 16, 4>(x, sel);
  if constexpr (G >= 32) xor_stage<R, 8, 3>(x, sel);
  if constexpr (G >= 16) xor_stage<R, 4, 2>(x, sel);
  if constexpr (G >= 8) xor_stage<R, 2, 1>(x, sel);
  if constexpr (G >= 4) xor_stage<R, 1, 0>(x, sel);
  clean_in_lane<R>(x);
}

template <int R>
__device__ __forceinline__ void wave_sort(float (&x)[R], const LaneSel& sel, int lane) {
  sort_in_lane<R>(x);
  merge_lanes<R, 2>(x, sel, lane);
  merge_lanes<R, 4>(x, sel, lane);
  merge_lanes<R, 8>(x, sel, lane);
  merge_lanes<R, 16>(x, sel, lane);
  merge_lanes<R, 32>(x, sel, lane);
  merge_lanes<R, 64>(x, sel, lane);
}

// ---- loads ---------------------------------------------------------------
template <int DTYPE>
__device__ __forceinline__ float load_sample(const void* base, int64_t idx) {
  if constexpr (DTYPE == 0) return reinterpret_cast<const float*>(base)[idx];
  else return (float)reinterpret_cast<const int16_t*>(base)[idx];   // exact
}

template <int R, int DTYPE>
__device__ __forceinline__ void load_group(float (&x)[R], const void* sig, int64_t off, int n, int lane) {
  const float inf = __builtin_inff();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int idx = r * 64 + lane;
    x[r] = (idx < n) ? load_sample<DTYPE>(sig, off + idx) : inf;
  }
}

// mean and sum of squared deviations in fp64 (two-pass, as np.mean / np.var do)
template <int R, int DTYPE>
__device__ __forceinline__ void group_moments(const float (&x)[R], int n, int lane, double& mean, double& m2) {
  const double scale = (DTYPE == 0) ? 1.0 : 1e-3;
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) if (r * 64 + lane < n) s += (double)x[r];
  s = wave_sum_f64(s);
  double mu = s / (double)n;                 // in sample units
  double q = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) if (r * 64 + lane < n) { double d = (double)x[r] - mu; q += d * d; }
  q = wave_sum_f64(q);
  if constexpr (DTYPE == 0) { mean = mu; m2 = q; }
  else { mean = s / 1000.0 / (double)n; m2 = q * (scale * scale); }
}

// store the sorted registers (blocked layout) to LDS: element lane*R + r
template <int R>
__device__ __forceinline__ void store_sorted(float* dst, const float (&x)[R], int lane) {
  if constexpr (R >= 4) {
#pragma unroll
    for (int r = 0; r < R; r += 4)
      *reinterpret_cast<float4*>(dst + lane * R + r) = make_float4(x[r], x[r + 1], x[r + 2], x[r + 3]);
  } else if constexpr (R == 2) {
    *reinterpret_cast<float2*>(dst + lane * 2) = make_float2(x[0], x[1]);
  } else {
    dst[lane] = x[0];
  }
}

// run extents of equal keys inside one sorted group, packed (start | end << 16), end exclusive
template <int R>
__device__ __forceinline__ void store_runs(int* dst, const float (&x)[R], int lane) {
  constexpr int N = 64 * R;
  const float nanv = __builtin_nanf("");
  float prev_last = lane_prev(x[R - 1], nanv);     // NaN != anything: lane 0 starts a run
  float next_first = lane_next(x[0], nanv);        // lane 63 ends a run
  int inc[R];      // running max of run-start indices inside the lane
  int run = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float p = (r == 0) ? prev_last : x[r - 1];
    int e = lane * R + r;
    run = (x[r] != p) ? e : run;
    inc[r] = run;
  }
  int carry = lane_prev_i(wave_scan_max_i32(run), 0);   // exclusive scan over lanes
  int suf[R];      // running max of (N - end) from the right: end = N - suf
  int acc = 0;
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    float q = (r == R - 1) ? next_first : x[r + 1];
    int e = lane * R + r;
    acc = (x[r] != q) ? max(acc, N - (e + 1)) : acc;
    suf[r] = acc;
  }
  // suffix max over lanes: mirror, prefix-scan, mirror back, shift
  int m = __builtin_amdgcn_ds_bpermute((63 - lane) << 2, acc);
  m = wave_scan_max_i32(m);
  m = __builtin_amdgcn_ds_bpermute((63 - lane) << 2, m);     // inclusive suffix max
  int carry_r = lane_next_i(m, 0);                              // exclusive
#pragma unroll
  for (int r = 0; r < R; ++r) {
    int start = max(inc[r], carry);
    int end = N - max(suf[r], carry_r);
    dst[lane * R + r] = start | (end << 16);
  }
}

// ---- the kernel ----------------------------------------------------------
template <int R0, int R1, int DTYPE, bool MWU, bool WELCH>
__global__ __launch_bounds__(64 * kWavesPerBlock)
void rank_stats_kernel(RankStatsArgs args) {
  constexpr int NA = 64 * R0, NB = 64 * R1;
  constexpr int KEYS = NA + kLdsPad + NB + kLdsPad;
  constexpr int WAVE_LDS = MWU ? 2 * KEYS : KEYS;       // in 4-byte words
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* keysA = lds_all + wave * WAVE_LDS;
  float* keysB = keysA + NA + kLdsPad;
  int* runA = reinterpret_cast<int*>(keysA + KEYS);
  int* runB = runA + NA + kLdsPad;

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  if (lane < kLdsPad) {
    keysA[NA + lane] = inf;
    keysB[NB + lane] = inf;
    if constexpr (MWU) { runA[NA + lane] = 0; runB[NB + lane] = 0; }
  }

  int64_t count = args.npos;
  const int32_t* list = nullptr;
  if (args.pos_list) {
    count = args.class_meta[args.class_id];
    list = args.pos_list + args.class_meta[kClassStride + args.class_id];
  }
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerBlock;

  for (int64_t it = wave_global; it < count; it += wave_stride) {
    const int64_t pos = list ? (int64_t)list[it] : it;
    int64_t o0, o1; int n0, n1;
    if (args.stride0 > 0) { o0 = pos * args.stride0; n0 = (int)args.stride0; }
    else { o0 = args.off0[pos]; n0 = (int)(args.off0[pos + 1] - o0); }
    if (args.stride1 > 0) { o1 = pos * args.stride1; n1 = (int)args.stride1; }
    else { o1 = args.off1[pos]; n1 = (int)(args.off1[pos + 1] - o1); }

    float xa[R0], xb[R1];
    load_group<R0, DTYPE>(xa, args.sig0, o0, n0, lane);
    load_group<R1, DTYPE>(xb, args.sig1, o1, n1, lane);

    if constexpr (WELCH) {
      double mean0, m20, mean1, m21;
      group_moments<R0, DTYPE>(xa, n0, lane, mean0, m20);
      group_moments<R1, DTYPE>(xb, n1, lane, mean1, m21);
      if (lane == 0) {
        double* mo = args.moments + pos * 4;
        mo[0] = mean0; mo[1] = m20; mo[2] = mean1; mo[3] = m21;
      }
    }

    wave_sort<R0>(xa, sel, lane);
    wave_sort<R1>(xb, sel, lane);

    store_sorted<R0>(keysA, xa, lane);
    store_sorted<R1>(keysB, xb, lane);
    if constexpr (MWU) {
      store_runs<R0>(runA, xa, lane);
      store_runs<R1>(runB, xb, lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- merge path: lane handles pooled elements [d0, d1)
    const int total = n0 + n1;
    const int per = __builtin_amdgcn_readfirstlane((total + 63) >> 6);
    const int d0 = min(lane * per, total);
    const int d1 = min(d0 + per, total);
    int lo = max(0, d0 - n1), hi = min(d0, n0);
    // wave-uniform iteration count: enough for the widest range
    const int span = min(min(n0, n1), total);
    int iters = 32 - __builtin_clz((unsigned)span | 1u);
    iters = __builtin_amdgcn_readfirstlane(iters);
#pragma unroll 1
    for (int s = 0; s < iters; ++s) {
      int mid = (lo + hi) >> 1;
      int jb = max(d0 - 1 - mid, 0);
      float a = keysA[mid];
      float b = keysB[jb];
      bool act = lo < hi;
      bool pred = a <= b;                 // A[mid] precedes B[d0-1-mid] (ties: group 1 first)
      lo = (act && pred) ? mid + 1 : lo;
      hi = (act && !pred) ? mid : hi;
    }
    int i = lo, j = d0 - lo;
    float a = keysA[i], b = keysB[j];
    float la = (i > 0) ? keysA[i - 1] : __builtin_nanf("");
    int num = i * n1 - j * n0;
    unsigned best = 0, s_lane = 0, tie_lane = 0;
    // all-tests mode also reproduces ks_2samp's float expression bit for bit: remember which steps
    // took group 1 (amask) and which steps tie the lane's running maximum (tmask)
    const int i_start = i;
    unsigned long long amask = 0, tmask = 0;
#pragma unroll 1
    for (int s = 0; s < per; ++s) {
      const bool act = (d0 + s) < d1;
      const bool takeA = a <= b;
      const float v = takeA ? a : b;
      if constexpr (MWU) {
        int rb = runB[j];
        int ra = runA[max(takeA ? i : i - 1, 0)];
        int ra_s = ra & 0xffff, ra_e = ra >> 16, rb_s = rb & 0xffff, rb_e = rb >> 16;
        int ownlen = takeA ? (ra_e - ra_s) : (rb_e - rb_s);
        int cross = takeA ? ((b == v) ? (rb_e - j) : 0) : ((la == v) ? (i - ra_s) : 0);
        int t = ownlen + cross;
        if (act) {
          s_lane += takeA ? (unsigned)(2 * j + cross) : 0u;
          tie_lane += (unsigned)(t * t - 1);
        }
      }
      if (act) {
        i += takeA ? 1 : 0;
        j += takeA ? 0 : 1;
        num += takeA ? n1 : -n0;
        la = takeA ? v : la;
      }
      float nv = takeA ? keysA[i] : keysB[j];
      a = (act && takeA) ? nv : a;
      b = (act && !takeA) ? nv : b;
      const bool run_end = fminf(a, b) != v;
      unsigned mag = (unsigned)abs(num);
      if constexpr (MWU) {
        const unsigned long long bit = 1ull << s;
        const bool cand = act && run_end;
        const unsigned magc = cand ? mag : 0u;
        amask |= (act && takeA) ? bit : 0ull;
        tmask = (magc > best) ? bit : ((cand && magc == best) ? (tmask | bit) : tmask);
        best = max(best, magc);
      } else {
        best = (act && run_end) ? max(best, mag) : best;
      }
    }
    const unsigned lane_best = best;
    best = wave_max_u32(best);
    if constexpr (MWU) {
      // D = max over the pooled points that attain the integer maximum of |fl(c0/n0) - fl(c1/n1)|
      // (scipy 1.2.1 ks_2samp: cdf = searchsorted(...)/(1.0*n); d = max(|cdf1 - cdf2|))
      unsigned long long tm = (lane_best == best && best > 0) ? tmask : 0ull;
      double dmax = 0.0;
      while (__ballot(tm != 0ull)) {
        if (tm != 0ull) {
          const int st = __ffsll((long long)tm) - 1;
          tm &= tm - 1ull;
          const int c0 = i_start + __popcll(amask & ((2ull << st) - 1ull));
          const int c1 = (d0 + st + 1) - c0;
          dmax = fmax(dmax, fabs((double)c0 / (double)n0 - (double)c1 / (double)n1));
        }
      }
      dmax = wave_max_f64(dmax);
      if (lane == 0) args.ks_d_ref[pos] = dmax;
    }
    if constexpr (MWU) {
      unsigned long long S = wave_sum_u64((unsigned long long)s_lane);
      unsigned long long T = wave_sum_u64((unsigned long long)tie_lane);
      if (lane == 0) { args.mwu_s[pos] = S; args.tie[pos] = T; }
    }
    if (lane == 0) args.ks_num[pos] = best;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nmod
