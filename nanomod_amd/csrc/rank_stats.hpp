// K1 — per-position rank statistics: shared types and the register sort of one 64-lane wave.
//
// K1 replaces the data-dependent part of getKStest (myDetect.py:327-343): the sorts / searchsorted of
// ks_2samp, the rankdata + tiecorrect of mannwhitneyu and the mean / var reductions of ttest_ind.
// It emits exact integers
//     ks_num = max_v |c0(v)*n1 - c1(v)*n0|,  c = #{x <= v}           (KS D numerator, KS-only mode)
//     mwu_s  = sum_{a in group 1} (#{b < a} + #{b <= a})             (U1 = n0*n1 - mwu_s/2)
//     tie    = sum_{pooled tie groups} (t^3 - t)                      (tiecorrect)
// the float form of D (all-tests mode) and fp64 (mean, M2) per group; the p-values are a separate kernel (K2).
// The forms of K1: ks_rank.hpp (KS-only), rank_all.hpp (all tests: same-class and any-class positions),
// big_rank.hpp (groups beyond 2048 samples).  This header holds what they share:
//   * RankStatsArgs;
//   * coalesced dword loads: lane l takes samples l, l+64, ... into R registers (padded with +inf);
//   * the bitonic sort of R registers x 64 lanes in its "mirror" form (every merge ascending):
//     compare-exchanges between registers of one lane are v_min / v_max; between lanes one DPP move (or
//     ds_swizzle / ds_bpermute beyond a 16-lane row) + one v_med3_f32 whose third operand is -inf (keep
//     min) or +inf (keep max) per lane.
// No MFMA: this is sort / scan / reduction work (BASELINE.json).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.hpp"

namespace nmod {

constexpr int kWavesPerBlock = 4;
constexpr int kLdsPad = 4;   // +inf sentinels after each sorted group
constexpr int kClassStride = 56;   // class_meta: [c] count, [kClassStride + c] offset, [2*kClassStride + c] cursor

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
  int32_t ks_rational_d;                       // KS-only mode: skip the float form of D (NMOD_FLAG_KS_RATIONAL_D)
  double* ks_d_ref;                            // [npos] max |fl(c0/n0) - fl(c1/n1)| exactly as ks_2samp forms it (all-tests mode)
  uint8_t* tied;                               // [npos] or null: 1 where the position's keys tie (see ks_rank_kernel FLAGS; all-tests: any tie)
  int32_t* redo_list; int32_t* redo_count;     // WIDE float32 form: positions whose streamed group's ties are counted by wide_redo_kernel
  // counting form (rank_count.hpp): gate[0] != 0 <=> cnt_probe_kernel found the batch event-like and rank_count_kernel ran;
  // cnt_done: one byte per entry of the work list (four per item, read as a dword), 1 where it produced the position's results
  // (0: left to the rank_hist_kernel<.., AFTER> launch that follows).  Both null: the counting form is not tried.
  int32_t* cnt_gate; uint8_t* cnt_done;
  // counting form for any coverage (rank_count_wide.hpp): where alt_gates[class_id] != 0 the class's positions were tried by it and
  // the sorting form walks what is left: alt_list / alt_meta (the layout of pos_list / class_meta).  Null: not in play.
  const int32_t* alt_gates; const int32_t* alt_list; const int32_t* alt_meta;
  int32_t cnt_mode;                            // rank_hist_kernel launches around the counting form: 1 = run only when the gate is clear (the plain
                                               // instance takes the whole list), 2 = only when it is set (the AFTER instance takes what is left); 0 = always
};

// compare-exchange of two registers.  (fminf / fmaxf put a canonicalising v_max x, x in front of every value of unknown
// origin — one per key loaded from memory; the keys are ordinary numbers or +-inf, so the bare instructions are used)
__device__ __forceinline__ void ce(float& lo, float& hi) {
#if defined(NMOD_CE_BUILTIN)
  float a = fminf(lo, hi), b = fmaxf(lo, hi);
#else
  float a, b;
  asm("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3" : "=&v"(a), "=v"(b) : "v"(lo), "v"(hi));
#endif
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

}  // namespace nmod
