// Building blocks of the K1 forms that keep a group of C = R*LG samples in R registers of LG lanes
// (LG = 8, 16 or 32), so that one wavefront holds 64/LG groups and sorts all of them with ONE instruction
// stream (ks_rank.hpp: the smaller group of 1-8 positions; rank_all.hpp: both groups of 1-4 positions).
// Measured on gfx950 (tools/valu_rate.hip): a compare-exchange between registers of a lane costs ~2 cycles per
// element (v_min / v_max), between lanes ~8 (v_mov_b32_dpp ~4 + v_med3_f32 ~4), so the layout keeps as many
// stages of the network inside a lane as the register file allows, and every cross-lane stage of the LG <= 16
// forms is a single-row DPP move + v_med3_f32 — no ds_swizzle / ds_bpermute.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rank_stats.hpp"

namespace nmod {

// ---- segmented (LG-lane) wave helpers --------------------------------------------------------
template <int LG>
__device__ __forceinline__ int seg_mirror_i(int x) {
  if constexpr (LG == 64) return __builtin_amdgcn_ds_bpermute((63 - (int)(threadIdx.x & 63)) << 2, x);
  else if constexpr (LG == 8) return dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, x);
  else if constexpr (LG == 16) return dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, x);
  else return __builtin_amdgcn_ds_swizzle(x, 0x7C1F);
}

// inclusive max-scan inside each LG-lane segment (values >= 0)
template <int LG>
__device__ __forceinline__ int seg_scan_max_i32(int v) {
  v = max(v, dpp_i<kDppRowShr + 1>(0, v));
  v = max(v, dpp_i<kDppRowShr + 2>(0, v));
  v = max(v, dpp_i<kDppRowShr + 4>(0, v));
  if constexpr (LG >= 16) v = max(v, dpp_i<kDppRowShr + 8>(0, v));
  if constexpr (LG >= 32) v = max(v, dpp_i<kDppRowBcast15, 0xA>(0, v));
  if constexpr (LG == 64) v = max(v, dpp_i<kDppRowBcast31, 0xC>(0, v));
  return v;   // LG == 8: lanes 8..15 of a row also see lanes 0..7; callers bias the second group
}

__device__ __forceinline__ double dpp_f64_row(double x, int which) {
  // (bound_ctrl with old = 0: every lane has a valid source in these patterns, and a tied `old` would cost a copy per half)
  long long b = __double_as_longlong(x);
  int lo = (int)(unsigned)b, hi = (int)(unsigned)((unsigned long long)b >> 32);
  switch (which) {
    case 0: lo = dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, lo); hi = dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, hi); break;
    case 1: lo = dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, lo); hi = dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, hi); break;
    case 2: lo = dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, lo); hi = dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, hi); break;
    default: lo = dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, lo); hi = dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, hi); break;
  }
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ double xor16_f64(double x) {
  long long b = __double_as_longlong(x);
  int lo = __builtin_amdgcn_ds_swizzle((int)(unsigned)b, 0x401F);
  int hi = __builtin_amdgcn_ds_swizzle((int)(unsigned)((unsigned long long)b >> 32), 0x401F);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}

// sum over the LG lanes of a group, result in every lane of the group
template <int LG>
__device__ __forceinline__ double seg_allsum_f64(double v) {
  v += dpp_f64_row(v, 0); v += dpp_f64_row(v, 1); v += dpp_f64_row(v, 2);
  if constexpr (LG >= 16) v += dpp_f64_row(v, 3);
  if constexpr (LG >= 32) v += xor16_f64(v);
  if constexpr (LG == 64) {                        // both 32-lane halves hold their sum in every lane
    auto lane_of = [](double x, int l) {
      const long long b = __double_as_longlong(x);
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), l);
      return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    v = lane_of(v, 0) + lane_of(v, 32);
  }
  return v;
}

// ---- sort of every LG-lane group of the wave --------------------------------------------------
template <int R, int LG>
__device__ __forceinline__ void seg_sort(float (&x)[R], const LaneSel& sel, int lane) {
  sort_in_lane<R>(x);
  merge_lanes<R, 2>(x, sel, lane);
  merge_lanes<R, 4>(x, sel, lane);
  merge_lanes<R, 8>(x, sel, lane);
  if constexpr (LG >= 16) merge_lanes<R, 16>(x, sel, lane);
  if constexpr (LG == 32) merge_lanes<R, 32>(x, sel, lane);
}

// ---- loads: 16 bytes (f32) / 8 bytes (i16) per lane per instruction when the row start is aligned
template <int R, int LG, int DTYPE>
__device__ __forceinline__ void load_packed(float (&x)[R], const void* sig, int64_t off, int n, int gl) {
  const float inf = __builtin_inff();
  bool aligned = (off & 3) == 0;
  if constexpr (R < 4) aligned = false;
  if (__ballot(!aligned) == 0ull) {
    if constexpr (R >= 4) {
#pragma unroll
      for (int k = 0; k < R / 4; ++k) {
        const int idx = k * (4 * LG) + 4 * gl;
        float v0 = inf, v1 = inf, v2 = inf, v3 = inf;
        if (idx + 3 < n) {
          if constexpr (DTYPE == 0) {
            float4 q = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(sig) + off + idx);
            v0 = q.x; v1 = q.y; v2 = q.z; v3 = q.w;
          } else {
            short4 q = *reinterpret_cast<const short4*>(reinterpret_cast<const int16_t*>(sig) + off + idx);
            v0 = (float)q.x; v1 = (float)q.y; v2 = (float)q.z; v3 = (float)q.w;
          }
        } else if (idx < n) {
          v0 = load_sample<DTYPE>(sig, off + idx);
          if (idx + 1 < n) v1 = load_sample<DTYPE>(sig, off + idx + 1);
          if (idx + 2 < n) v2 = load_sample<DTYPE>(sig, off + idx + 2);
        }
        x[4 * k] = v0; x[4 * k + 1] = v1; x[4 * k + 2] = v2; x[4 * k + 3] = v3;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int idx = r * LG + gl;
      x[r] = (idx < n) ? load_sample<DTYPE>(sig, off + idx) : inf;
    }
  }
}

// mean and sum of squared deviations of one group in fp64; pads are +inf.  One pass over the keys, shifted by the
// group's first sample K:  mean = K + S1/n,  M2 = S2 - S1^2/n  with S1 = sum (x - K), S2 = sum (x - K)^2.
// The shift keeps the cancellation in M2 at ~(1 + (mean - K)^2 / var) ulps of fp64 — np.var's two-pass result to
// ~1e-14 — for half the instructions of two masked passes (a pad becomes K: it adds 0 to both sums).
template <int R, int LG, int DTYPE>
__device__ __forceinline__ void seg_moments(const float (&x)[R], int n, double& mean, double& m2, double rcp_n = 0.0) {
  const float inf = __builtin_inff();
  const int lane = threadIdx.x & 63;
  float kf = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, __float_as_int(x[0])));
  kf = (kf == inf) ? 0.0f : kf;                                   // empty group
  const double K = (double)kf;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float xm = (x[r] != inf) ? x[r] : kf;
    const double d = (double)xm - K;
    s1 += d;
    s2 = __fma_rn(d, d, s2);
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);          // do not convert all R keys to fp64 at once
  }
  s1 = seg_allsum_f64<LG>(s1);
  s2 = seg_allsum_f64<LG>(s2);
  const double dn = (double)n;
  const double rn = (rcp_n != 0.0) ? rcp_n : 1.0 / dn;                // one division for mean and M2 (none when the caller has fl(1/n))
  const double mu = K + s1 * rn;
  const double q = s2 - s1 * s1 * rn;
  if constexpr (DTYPE == 0) { mean = mu; m2 = q; }
  else { mean = mu * 1e-3; m2 = q * 1e-6; }
}

}  // namespace nmod
