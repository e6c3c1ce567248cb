// Wave64 lane-permutation and reduction primitives for gfx950 (CDNA4).
// Everything here assumes a 64-lane wavefront; there is no other target.
#pragma once
#include <hip/hip_runtime.h>

namespace nmod {

#define NMOD_QP(a, b, c, d) ((a) | ((b) << 2) | ((c) << 4) | ((d) << 6))
constexpr int kDppRowShl = 0x100;      // + n : lane i reads lane i+n of its 16-lane row
constexpr int kDppRowShr = 0x110;      // + n : lane i reads lane i-n
constexpr int kDppWaveShr1 = 0x138;    // lane i reads lane i-1 across the whole wave
constexpr int kDppWaveShl1 = 0x130;    // lane i reads lane i+1 across the whole wave
constexpr int kDppRowMirror = 0x140;
constexpr int kDppRowHalfMirror = 0x141;
constexpr int kDppRowBcast15 = 0x142;
constexpr int kDppRowBcast31 = 0x143;

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ int dpp_i(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL);
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_f(float old, float src) {
  return __int_as_float(dpp_i<CTRL, ROW_MASK, BANK_MASK, BOUND_CTRL>(__float_as_int(old), __float_as_int(src)));
}

// Which exchanges go through the LDS crossbar (ds_swizzle, bit mode: lane ^ mask inside 32 lanes) instead of
// a DPP move.  A DPP move is a VALU instruction (~6 cycles of the SIMD with the v_med3 that consumes it,
// tools/valu_rate.hip); ds_swizzle issues on the LDS pipe, which K1's sort leaves mostly idle.
// bits: 1 xor1, 2 xor2, 4 xor4, 8 xor8, 16 mirror2, 32 mirror4, 64 mirror8, 128 mirror16
#ifndef NMOD_SWZ_MASK
#define NMOD_SWZ_MASK 0
#endif
template <int XORMASK>
__device__ __forceinline__ float lane_swizzle_xor(float x) {
  return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x1F | (XORMASK << 10)));
}

// value of lane (l ^ M) for M in {1,2,4,8,16}
template <int M>
__device__ __forceinline__ float lane_xor(float x) {
  if constexpr (M <= 8 && (NMOD_SWZ_MASK & M) != 0) return lane_swizzle_xor<M>(x);
  else
  // every lane has a valid source in these patterns, so bound_ctrl:1 with old = 0 lets the
  // compiler emit the bare v_mov_b32_dpp (a tied `old` costs an extra v_mov per move)
  if constexpr (M == 1) return dpp_f<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0.0f, x);
  else if constexpr (M == 2) return dpp_f<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0.0f, x);
  else if constexpr (M == 8) return dpp_f<0x120 + 8, 0xf, 0xf, true>(0.0f, x);   // row_ror:8 == xor 8 inside a row
  else if constexpr (M == 4) {
#if defined(NMOD_XOR4_BANKS)
    float y = dpp_f<kDppRowShl + 4, 0xf, 0x5>(x, x);   // banks 0,2 read lane+4
    return dpp_f<kDppRowShr + 4, 0xf, 0xA>(y, x);      // banks 1,3 read lane-4
#else
    // l ^ 4 = (l ^ 7) ^ 3: a mirror of the 8-lane half, then of the quad — two full moves (the bank-masked pair above
    // needs a copy of x first and writes one register twice: three dependent instructions)
    return dpp_f<NMOD_QP(3, 2, 1, 0), 0xf, 0xf, true>(0.0f, dpp_f<kDppRowHalfMirror, 0xf, 0xf, true>(0.0f, x));
#endif
  } else {
    static_assert(M == 16, "lane_xor: unsupported distance");
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x401F));  // bit mode: xor 0x10
  }
}

// value of lane (l ^ (G-1)): reversal inside aligned groups of G lanes
template <int G>
__device__ __forceinline__ float lane_mirror(float x, int lane) {
  if constexpr (G <= 16 && (NMOD_SWZ_MASK & (G * 8)) != 0) return lane_swizzle_xor<G - 1>(x);
  else if constexpr (G == 2) return dpp_f<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0.0f, x);
  else if constexpr (G == 4) return dpp_f<NMOD_QP(3, 2, 1, 0), 0xf, 0xf, true>(0.0f, x);
  else if constexpr (G == 8) return dpp_f<kDppRowHalfMirror, 0xf, 0xf, true>(0.0f, x);
  else if constexpr (G == 16) return dpp_f<kDppRowMirror, 0xf, 0xf, true>(0.0f, x);
  else if constexpr (G == 32) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(x), 0x7C1F));  // xor 0x1f
  else {
    static_assert(G == 64, "lane_mirror: unsupported group");
    return __int_as_float(__builtin_amdgcn_ds_bpermute((63 - lane) << 2, __float_as_int(x)));
  }
}

// value of lane l-1 (lane 0 gets `fill`) / lane l+1 (lane 63 gets `fill`)
__device__ __forceinline__ float lane_prev(float x, float fill) { return dpp_f<kDppWaveShr1>(fill, x); }
__device__ __forceinline__ float lane_next(float x, float fill) { return dpp_f<kDppWaveShl1>(fill, x); }
__device__ __forceinline__ int lane_prev_i(int x, int fill) { return dpp_i<kDppWaveShr1>(fill, x); }
__device__ __forceinline__ int lane_next_i(int x, int fill) { return dpp_i<kDppWaveShl1>(fill, x); }

// ---- wave-wide reductions: 4 DPP steps inside each 16-lane row, then the four row results
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = max(v, (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2)>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1)>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowHalfMirror>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowMirror>((int)v, (int)v));
  unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
  auto step = [](unsigned long long x, auto tag) {
    constexpr int C = decltype(tag)::value;
    int lo = dpp_i<C>(0, (int)(unsigned)x);
    int hi = dpp_i<C>(0, (int)(unsigned)(x >> 32));
    return x + (((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
  };
  v = step(v, std::integral_constant<int, NMOD_QP(1, 0, 3, 2)>{});
  v = step(v, std::integral_constant<int, NMOD_QP(2, 3, 0, 1)>{});
  v = step(v, std::integral_constant<int, kDppRowHalfMirror>{});
  v = step(v, std::integral_constant<int, kDppRowMirror>{});
  unsigned long long r = 0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    unsigned lo = __builtin_amdgcn_readlane((unsigned)v, row * 16);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), row * 16);
    r += ((unsigned long long)hi << 32) | lo;
  }
  return r;
}

__device__ __forceinline__ double wave_sum_f64(double v) {
  auto step = [](double x, auto tag) {
    constexpr int C = decltype(tag)::value;
    long long b = __double_as_longlong(x);
    int lo = dpp_i<C>(0, (int)(unsigned)b);
    int hi = dpp_i<C>(0, (int)(unsigned)((unsigned long long)b >> 32));
    return x + __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
  };
  v = step(v, std::integral_constant<int, NMOD_QP(1, 0, 3, 2)>{});
  v = step(v, std::integral_constant<int, NMOD_QP(2, 3, 0, 1)>{});
  v = step(v, std::integral_constant<int, kDppRowHalfMirror>{});
  v = step(v, std::integral_constant<int, kDppRowMirror>{});
  double r = 0.0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    long long b = __double_as_longlong(v);
    unsigned lo = __builtin_amdgcn_readlane((unsigned)b, row * 16);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), row * 16);
    r += __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
  }
  return r;
}

__device__ __forceinline__ double wave_max_f64(double v) {
  auto step = [](double x, auto tag) {
    constexpr int C = decltype(tag)::value;
    long long b = __double_as_longlong(x);
    int lo = dpp_i<C>(0, (int)(unsigned)b);
    int hi = dpp_i<C>(0, (int)(unsigned)((unsigned long long)b >> 32));
    return fmax(x, __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo)));
  };
  v = step(v, std::integral_constant<int, NMOD_QP(1, 0, 3, 2)>{});
  v = step(v, std::integral_constant<int, NMOD_QP(2, 3, 0, 1)>{});
  v = step(v, std::integral_constant<int, kDppRowHalfMirror>{});
  v = step(v, std::integral_constant<int, kDppRowMirror>{});
  double r = 0.0;
#pragma unroll
  for (int row = 0; row < 4; ++row) {
    long long b = __double_as_longlong(v);
    unsigned lo = __builtin_amdgcn_readlane((unsigned)b, row * 16);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), row * 16);
    r = fmax(r, __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo)));
  }
  return r;
}

// inclusive max-scan over lanes (lane l gets max over lanes 0..l), ints >= 0
__device__ __forceinline__ int wave_scan_max_i32(int v) {
  v = max(v, dpp_i<kDppRowShr + 1>(0, v));        // invalid source lanes keep old = 0
  v = max(v, dpp_i<kDppRowShr + 2>(0, v));
  v = max(v, dpp_i<kDppRowShr + 4>(0, v));
  v = max(v, dpp_i<kDppRowShr + 8>(0, v));
  v = max(v, dpp_i<kDppRowBcast15, 0xA>(0, v));   // rows 1,3 take lane 15 of the previous row
  v = max(v, dpp_i<kDppRowBcast31, 0xC>(0, v));   // rows 2,3 take lane 31
  return v;
}

}  // namespace nmod
