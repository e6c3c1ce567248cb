// K1, packed form — for positions whose two groups fall in the SAME size class.
//
// A group of C = R*LG samples lives in R registers of LG lanes (LG = 8, 16 or 32).
// One wavefront therefore holds 64/LG groups = 32/LG positions (LG = 8: four positions per
// wave) and sorts all of them with ONE instruction stream.  Measured on gfx950 (tools/valu_rate.hip):
// a compare-exchange between registers of a lane costs ~2 cycles per element (v_min / v_max),
// between lanes ~8 (v_mov_b32_dpp ~4 + v_med3_f32 ~4), so the layout keeps as many stages of the
// network inside a lane as the register file allows (R = 32).  Compared with the general kernel
// (64 lanes per group) more of the bitonic network runs between registers of a lane
// (v_min/v_max, 1 instruction per element per stage) and every cross-lane stage of the LG = 16
// form is a single-row DPP move + v_med3_f32 — no ds_swizzle / ds_bpermute at all.
// The merge-path phase gives each position 2*LG lanes; its per-step bookkeeping is arranged so
// that the compiler emits select / add-with-carry forms and no branches.
// Same outputs, bit for bit, as rank_stats_kernel (tests compare both against the oracle).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rank_stats.hpp"

namespace nmod {

// ---- segmented (LG-lane) wave helpers --------------------------------------------------------
template <int LG>
__device__ __forceinline__ int seg_mirror_i(int x) {
  if constexpr (LG == 8) return dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, x);
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
  if constexpr (LG == 32) v = max(v, dpp_i<kDppRowBcast15, 0xA>(0, v));
  return v;   // LG == 8: lanes 8..15 of a row also see lanes 0..7; callers bias the second group
}

// all lanes of a 16-lane row get the row's reduction
template <typename T, typename F>
__device__ __forceinline__ T row_allreduce_u32(T v, F f) {
  v = f(v, (T)dpp_i<NMOD_QP(1, 0, 3, 2)>((int)v, (int)v));
  v = f(v, (T)dpp_i<NMOD_QP(2, 3, 0, 1)>((int)v, (int)v));
  v = f(v, (T)dpp_i<kDppRowHalfMirror>((int)v, (int)v));
  v = f(v, (T)dpp_i<kDppRowMirror>((int)v, (int)v));
  return v;
}

__device__ __forceinline__ double dpp_f64_row(double x, int which) {
  long long b = __double_as_longlong(x);
  int lo = (int)(unsigned)b, hi = (int)(unsigned)((unsigned long long)b >> 32);
  switch (which) {
    case 0: lo = dpp_i<NMOD_QP(1, 0, 3, 2)>(lo, lo); hi = dpp_i<NMOD_QP(1, 0, 3, 2)>(hi, hi); break;
    case 1: lo = dpp_i<NMOD_QP(2, 3, 0, 1)>(lo, lo); hi = dpp_i<NMOD_QP(2, 3, 0, 1)>(hi, hi); break;
    case 2: lo = dpp_i<kDppRowHalfMirror>(lo, lo); hi = dpp_i<kDppRowHalfMirror>(hi, hi); break;
    default: lo = dpp_i<kDppRowMirror>(lo, lo); hi = dpp_i<kDppRowMirror>(hi, hi); break;
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
  if constexpr (LG == 32) v += xor16_f64(v);
  return v;
}

// reductions over the 2*LG lanes of one position; every lane gets its own position's result
template <int LG>
__device__ __forceinline__ unsigned pos_max_u32(unsigned v, int lane) {
  v = row_allreduce_u32<unsigned>(v, [](unsigned a, unsigned b) { return max(a, b); });
  if constexpr (LG == 8) return v;                 // a position is one 16-lane row
  unsigned r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  unsigned r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  if constexpr (LG == 16) return (lane < 32) ? max(r0, r1) : max(r2, r3);
  else return max(max(r0, r1), max(r2, r3));
}
template <int LG>
__device__ __forceinline__ unsigned long long pos_sum_u32(unsigned v, int lane) {   // per-lane u32, exact u64 total
  // row totals fit in u32 only if each lane's value < 2^28; callers guarantee that
  v = row_allreduce_u32<unsigned>(v, [](unsigned a, unsigned b) { return a + b; });
  if constexpr (LG == 8) return (unsigned long long)v;
  unsigned long long r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16);
  unsigned long long r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
  if constexpr (LG == 16) return (lane < 32) ? (r0 + r1) : (r2 + r3);
  else return r0 + r1 + r2 + r3;
}
template <int LG>
__device__ __forceinline__ double pos_max_f64(double v, int lane) {
  v = fmax(v, dpp_f64_row(v, 0)); v = fmax(v, dpp_f64_row(v, 1));
  v = fmax(v, dpp_f64_row(v, 2)); v = fmax(v, dpp_f64_row(v, 3));
  if constexpr (LG >= 16) v = fmax(v, xor16_f64(v));
  if constexpr (LG == 32) {
    long long b = __double_as_longlong(v);
    unsigned lo0 = __builtin_amdgcn_readlane((unsigned)b, 0), hi0 = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), 0);
    unsigned lo1 = __builtin_amdgcn_readlane((unsigned)b, 32), hi1 = __builtin_amdgcn_readlane((unsigned)((unsigned long long)b >> 32), 32);
    v = fmax(__longlong_as_double((long long)(((unsigned long long)hi0 << 32) | lo0)),
             __longlong_as_double((long long)(((unsigned long long)hi1 << 32) | lo1)));
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

// mean and sum of squared deviations of one group (fp64, two-pass); pads are +inf
template <int R, int LG, int DTYPE>
__device__ __forceinline__ void seg_moments(const float (&x)[R], int n, double& mean, double& m2) {
  const float inf = __builtin_inff();
  double s = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    s += (x[r] != inf) ? (double)x[r] : 0.0;
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // do not convert all R keys to fp64 at once
  }
  s = seg_allsum_f64<LG>(s);
  const double mu = s / (double)n;
  double q = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float xr = x[r];
    asm volatile("" : "+v"(xr));          // opaque copy: stops the compiler keeping R fp64 conversions live from pass 1
    double d = (double)xr - mu;
    q += (xr != inf) ? d * d : 0.0;
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  q = seg_allsum_f64<LG>(q);
  if constexpr (DTYPE == 0) { mean = mu; m2 = q; }
  else { mean = s / 1000.0 / (double)n; m2 = q * 1e-6; }
}

// run extents of equal keys inside each sorted group: (start | end << 16), indices inside the group
template <int R, int LG>
__device__ __forceinline__ void seg_store_runs(int* dst, float (&y)[R], int gl, bool is_b) {   // clobbers y
  constexpr int N = R * LG;
  // LG == 8: both groups of a position share one DPP row, so the second group's scan values are
  // biased by N: whatever leaks in from the first group (< N) can never win a max
  const int bias = (LG == 8 && is_b) ? N : 0;
  const float nanv = __builtin_nanf("");
  float prev_last = lane_prev(y[R - 1], nanv);
  float next_first = lane_next(y[0], nanv);
  prev_last = (gl == 0) ? nanv : prev_last;
  next_first = (gl == LG - 1) ? nanv : next_first;
  // Each of the four sweeps below works on an opaque in-place copy of the keys (empty asm): without
  // that the compiler shares the 2*R comparison masks between the sweeps and keeps them all alive,
  // which costs ~90 VGPRs through SGPR spills.
  auto launder = [&]() {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(y[r]));
  };
  // pass 1: per-lane totals only (keeping per-element arrays would cost 2*R registers)
  int run = bias;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float p = (r == 0) ? prev_last : y[r - 1];
    run = (y[r] != p) ? (gl * R + r + bias) : run;
  }
  int carry = lane_prev_i(seg_scan_max_i32<LG>(run), 0);
  carry = (gl == 0) ? bias : carry;
  launder();
  int acc = bias;
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    float q = (r == R - 1) ? next_first : y[r + 1];
    acc = (y[r] != q) ? max(acc, N - (gl * R + r + 1) + bias) : acc;
  }
  int m = seg_mirror_i<LG>(acc);
  m = seg_scan_max_i32<LG>(m);
  m = seg_mirror_i<LG>(m);
  int carry_r = lane_next_i(m, 0);
  carry_r = (gl == LG - 1) ? bias : carry_r;
  // pass 2: replay with the carries; start and end are the two 16-bit halves of one LDS word
  unsigned short* dst16 = reinterpret_cast<unsigned short*>(dst);
  launder();
  run = carry;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float p = (r == 0) ? prev_last : y[r - 1];
    run = (y[r] != p) ? (gl * R + r + bias) : run;
    dst16[2 * (gl * R + r)] = (unsigned short)(run - bias);
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  launder();
  acc = carry_r;
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    float q = (r == R - 1) ? next_first : y[r + 1];
    acc = (y[r] != q) ? max(acc, N - (gl * R + r + 1) + bias) : acc;
    dst16[2 * (gl * R + r) + 1] = (unsigned short)(N - (acc - bias));
    if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- the kernel -------------------------------------------------------------------------------
template <int R, int LG, int DTYPE, bool ALL>
__global__ __launch_bounds__(64 * kWavesPerBlock)
void rank_stats_packed_kernel(RankStatsArgs args) {
  static_assert(LG == 8 || LG == 16 || LG == 32, "lanes per group");
  static_assert(R <= 32, "the per-lane step bitmasks of the merge loop are 32 bits wide");
  constexpr int C = R * LG;                       // capacity per group
  constexpr int LP = 2 * LG;                      // lanes per position
  constexpr int PW = 64 / LP;                     // positions per wave
  constexpr int GROUP_WORDS = C + kLdsPad;
  constexpr int POS_WORDS = 2 * GROUP_WORDS;
  constexpr int KEY_WORDS = PW * POS_WORDS;
  constexpr int WAVE_WORDS = ALL ? 2 * KEY_WORDS : KEY_WORDS;
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);                 // lane inside its group
  const int pl = lane & (LP - 1);                 // lane inside its position
  const int slot = lane / LP;                     // which of the wave's positions
  const bool is_b = (lane & LG) != 0;             // second group of the position

  float* wave_lds = lds_all + wave * WAVE_WORDS;
  float* keysA = wave_lds + slot * POS_WORDS;     // this lane's position: group 1 keys, then group 2 keys
  float* keysB = keysA + GROUP_WORDS;
  float* my_keys = is_b ? keysB : keysA;
  int* runA = reinterpret_cast<int*>(keysA + KEY_WORDS);
  int* runB = runA + GROUP_WORDS;
  int* my_runs = is_b ? runB : runA;

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  if (gl < kLdsPad) {
    my_keys[C + gl] = inf;
    if constexpr (ALL) my_runs[C + gl] = 0;
  }

  int64_t count = args.npos;
  const int32_t* list = nullptr;
  if (args.pos_list) {
    count = args.class_meta[args.class_id];
    list = args.pos_list + args.class_meta[kClassStride + args.class_id];
  }
  const int64_t items = (count + PW - 1) / PW;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerBlock;

  for (int64_t it = wave_global; it < items; it += wave_stride) {
    const int64_t li = it * PW + slot;
    const bool valid = li < count;
    const int64_t pos = valid ? (list ? (int64_t)list[li] : li) : 0;
    int64_t o0 = 0, o1 = 0; int n0 = 0, n1 = 0;
    if (valid) {
      if (args.stride0 > 0) { o0 = pos * args.stride0; n0 = (int)args.stride0; }
      else { o0 = args.off0[pos]; n0 = (int)(args.off0[pos + 1] - o0); }
      if (args.stride1 > 0) { o1 = pos * args.stride1; n1 = (int)args.stride1; }
      else { o1 = args.off1[pos]; n1 = (int)(args.off1[pos + 1] - o1); }
    }

    float x[R];
    load_packed<R, LG, DTYPE>(x, is_b ? args.sig1 : args.sig0, is_b ? o1 : o0, is_b ? n1 : n0, gl);

    if constexpr (ALL) {
      double mean, m2;
      seg_moments<R, LG, DTYPE>(x, is_b ? n1 : n0, mean, m2);
      if (valid && gl == 0) {
        double* mo = args.moments + pos * 4 + (is_b ? 2 : 0);
        mo[0] = mean; mo[1] = m2;
      }
    }

    seg_sort<R, LG>(x, sel, lane);
    store_sorted<R>(my_keys, x, gl);
    if constexpr (ALL) seg_store_runs<R, LG>(my_runs, x, gl, is_b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- merge path: the position's 2*LG lanes each take `per` consecutive pooled elements
    const int total = n0 + n1;
    const int per = (total + LP - 1) / LP;
    const int d0 = min(pl * per, total);
    const int my_steps = min(d0 + per, total) - d0;
    int steps_w = per, span = min(n0, n1);
    if constexpr (PW == 4) {
      steps_w = max(max(__builtin_amdgcn_readlane(per, 0), __builtin_amdgcn_readlane(per, 16)),
                    max(__builtin_amdgcn_readlane(per, 32), __builtin_amdgcn_readlane(per, 48)));
      span = max(max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 16)),
                 max(__builtin_amdgcn_readlane(span, 32), __builtin_amdgcn_readlane(span, 48)));
    } else if constexpr (PW == 2) {
      steps_w = max(__builtin_amdgcn_readlane(per, 0), __builtin_amdgcn_readlane(per, 32));
      span = max(__builtin_amdgcn_readlane(span, 0), __builtin_amdgcn_readlane(span, 32));
    } else {
      steps_w = __builtin_amdgcn_readfirstlane(per);
      span = __builtin_amdgcn_readfirstlane(span);
    }
    const int iters = 32 - __builtin_clz((unsigned)span | 1u);
    int lo = max(0, d0 - n1), hi = min(d0, n0);
#pragma unroll 1
    for (int s = 0; s < iters; ++s) {
      const int mid = (lo + hi) >> 1;
      const float a = keysA[mid];
      const float b = keysB[max(d0 - 1 - mid, 0)];
      const bool act = lo < hi;
      const bool pred = a <= b;                    // A[mid] precedes B[d0-1-mid] (ties: group 1 first)
      lo = (act && pred) ? mid + 1 : lo;
      hi = (act && !pred) ? mid : hi;
    }
    int i = lo;                                    // taken from group 1
    int jb = (d0 - lo) + GROUP_WORDS;              // taken from group 2, as a word index relative to keysA
    float a = keysA[i], b = keysA[jb];
    float la = keysA[max(i - 1, 0)];
    la = (i > 0) ? la : __builtin_nanf("");
    const int nn = n0 + n1;
    int tn0 = d0 * n0;                             // (i + j) * n0
    unsigned best = 0, s_lane = 0, tie_lane = 0;
    const int i_start = i;
    unsigned amask = 0, tmask = 0;                 // one bit per step: per <= 2*C/LP = R <= 32
#pragma unroll 1
    for (int s = 0; s < steps_w; ++s) {
      const bool act = s < my_steps;
      const bool takeA = a <= b;
      const bool tA = act && takeA, tB = act && !takeA;
      const float v = takeA ? a : b;
      if constexpr (ALL) {
        const int j = jb - GROUP_WORDS;
        const int rb = runB[j];
        const int ra = runA[max(takeA ? i : i - 1, 0)];
        const int ra_s = ra & 0xffff, ra_e = ra >> 16, rb_s = rb & 0xffff, rb_e = rb >> 16;
        const int ownlen = takeA ? (ra_e - ra_s) : (rb_e - rb_s);
        const int cross = takeA ? ((b == v) ? (rb_e - j) : 0) : ((la == v) ? (i - ra_s) : 0);
        const int t = ownlen + cross;
        s_lane += tA ? (unsigned)(2 * j + cross) : 0u;
        tie_lane += act ? (unsigned)(__mul24(t, t) - 1) : 0u;     // t <= 2*C <= 4096
        la = tA ? v : la;
      }
      i += tA ? 1 : 0;
      jb += tB ? 1 : 0;
      tn0 += n0;
      const float nv = keysA[takeA ? i : jb];
      a = tA ? nv : a;
      b = tB ? nv : b;
      const bool run_end = (a != v) && (b != v);   // next pooled value differs (both heads are >= v)
      const bool cand = act && run_end;
      const int num = __mul24(i, nn) - tn0;        // c0*n1 - c1*n0 with c0 = i, c0 + c1 = i + j (operands < 2^23)
      const unsigned mag = (unsigned)abs(num);
      const unsigned magc = cand ? mag : 0u;
      if constexpr (ALL) {
        const unsigned bit = 1u << s;
        amask |= tA ? bit : 0u;
        tmask = (magc > best) ? bit : ((cand && magc == best) ? (tmask | bit) : tmask);
      }
      best = max(best, magc);
    }
    const unsigned lane_best = best;
    best = pos_max_u32<LG>(best, lane);
    const bool writer = valid && pl == 0;
    if constexpr (ALL) {
      unsigned tm = (lane_best == best && best > 0) ? tmask : 0u;
      double dmax = 0.0;
      while (__ballot(tm != 0u)) {
        if (tm != 0u) {
          const int st = __ffs((int)tm) - 1;
          tm &= tm - 1u;
          const int c0 = i_start + __popc(amask & (unsigned)((2ull << st) - 1ull));
          const int c1 = (d0 + st + 1) - c0;
          dmax = fmax(dmax, fabs((double)c0 / (double)n0 - (double)c1 / (double)n1));
        }
      }
      dmax = pos_max_f64<LG>(dmax, lane);
      const unsigned long long S = pos_sum_u32<LG>(s_lane, lane);
      const unsigned long long T = pos_sum_u32<LG>(tie_lane, lane);
      if (writer) { args.ks_d_ref[pos] = dmax; args.mwu_s[pos] = S; args.tie[pos] = T; }
    }
    if (writer) args.ks_num[pos] = best;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nmod
