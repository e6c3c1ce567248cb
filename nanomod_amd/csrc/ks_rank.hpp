// K1, KS-only form (tests mask = KS: BASELINE.json configs[1], "KS + weighted Stouffer").
//
// ks_2samp (myDetect.py:341 -> scipy 1.2.1) needs D = max_v |F0(v) - F1(v)| over the pooled points.
// D is symmetric in the two groups, so call the smaller one S (m samples) and the other Q (q samples):
//   1. S is sorted in registers exactly like the packed kernel (R registers x LG lanes, bitonic
//      network, DPP + v_med3 across lanes) and written to wave-private LDS;
//   2. every sample x of Q finds L(x) = #{s < x} by a branchless 1+log2(C)-step binary search in LDS
//      and, only when x ties with an S value, U(x) = #{s <= x} by a second search;
//   3. one ds_add_u32 per sample builds the histograms of L and U (two 16-bit halves of a word);
//   4. with cumL / cumU their prefix sums, the pooled points are
//        v = an S value with upper rank k :  (#{x <= v}, #{s <= v}) = (cumL(k-1), k)
//        v = largest Q sample with U = k   :                          (cumU(k),   k)
//      (k restricted to run ends of S); every other pooled point is dominated by these two, so
//        ks_num = max_k max(|cumL(k-1)*m - k*q|, |cumU(k)*m - k*q|)
//      is the same exact integer max|c0*n1 - c1*n0| the merge-path kernels produce.
//      Without ties cumL == cumU and the maximum collapses to max_{k<m} max(a_k, q - a_k),
//      a_k = cumU(k)*m - k*q: five VALU instructions per histogram bin.
// Against the two-sort + merge-path form this removes one sort and the whole sequential merge:
// ~300 instead of ~500 VALU instructions per 200 v 200 position (rocprof SQ_INSTS_VALU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rank_stats_packed.hpp"

namespace nmod {

// LDS layout: sorted keys and histogram bins are skewed by 4 pad words per 32 (word(i) = i + 4*(i>>5)).
// A power-of-two binary search probes indices == 2^j - 1 (mod 2^(j+1)); unskewed, every probe of the
// first steps lands on one bank (rocprof: 85 % of the LDS cycles were bank conflicts).  With the skew
// the probes of different 32-blocks fall on different banks, and because the search position is
// always a multiple of the current step the skewed offsets are compile-time constants: no extra VALU.
constexpr int kKsTail = 8;         // +inf sentinels / spare bins after the last skewed word

__host__ __device__ constexpr int ks_skew(int i) { return i + ((i >> 5) << 2); }
__device__ __forceinline__ int ks_skew_rt(int i) { return i + ((i >> 5) << 2); }
__host__ __device__ constexpr int ks_region_words(int C) { return ks_skew(C) + kKsTail; }
__host__ __device__ constexpr int ks_rank_pos_words(int C) {
  // keys + histogram, padded so that consecutive positions start 8 banks apart
  int w = 2 * ks_region_words(C);
  while ((w & 31) != 8) w += 4;
  return w;
}

template <int LG>
__device__ __forceinline__ unsigned seg_allmax_u32(unsigned v) {
  v = max(v, (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, (int)v));
  v = max(v, (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, (int)v));
  if constexpr (LG >= 16) v = max(v, (unsigned)dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, (int)v));
  if constexpr (LG >= 32) v = max(v, (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F));
  if constexpr (LG == 64) v = max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 32));
  return v;
}

// exclusive prefix sum over the LG lanes of a segment (values may be packed 16|16 counters)
template <int LG>
__device__ __forceinline__ unsigned seg_exscan_add_u32(unsigned v, int gl) {
  unsigned inc = v;
  auto step = [&](auto tag, int dist) {
    constexpr int C = decltype(tag)::value;
    unsigned t = (unsigned)dpp_i<C, 0xf, 0xf, true>(0, (int)inc);   // bound_ctrl: lanes without a source read 0
    inc += (gl >= dist) ? t : 0u;                                     // do not cross into the previous segment
  };
  step(std::integral_constant<int, kDppRowShr + 1>{}, 1);
  step(std::integral_constant<int, kDppRowShr + 2>{}, 2);
  step(std::integral_constant<int, kDppRowShr + 4>{}, 4);
  if constexpr (LG >= 16) step(std::integral_constant<int, kDppRowShr + 8>{}, 8);
  if constexpr (LG >= 32) {
    unsigned t = (unsigned)dpp_i<kDppRowBcast15, 0xA>(0, (int)inc);  // rows 1,3 <- lane 15 of rows 0,2
    inc += ((gl & 16) != 0) ? t : 0u;
  }
  if constexpr (LG == 64) {
    unsigned t = (unsigned)dpp_i<kDppRowBcast31, 0xC>(0, (int)inc);  // rows 2,3 <- lane 31
    inc += (gl >= 32) ? t : 0u;
  }
  return inc - v;
}

template <int R, int LG>
__device__ __forceinline__ void seg_sort_any(float (&x)[R], const LaneSel& sel, int lane) {
  seg_sort<R, (LG == 64 ? 32 : LG)>(x, sel, lane);
  if constexpr (LG == 64) merge_lanes<R, 64>(x, sel, lane);
}

// branchless binary search in the skewed key array: returns the pointer to skewed word L (LE = false:
// L = #{s < x}) or U (LE = true: U = #{s <= x}); `base` points at key 0.
template <int C, int STEPS, bool LE>
__device__ __forceinline__ const float* ks_search(const float* base, float x, const float** block_start = nullptr) {
  const float* p = base;
  const float last = base[ks_skew(C - 1)];
  const bool all = LE ? (last <= x) : (last < x);                 // rank C: every key is below x
#pragma unroll
  for (int st = STEPS - 1; st >= 0; --st) {
    constexpr int dummy = 0; (void)dummy;
    const int h = 1 << st;
    const int hp = ks_skew(h);                                     // skewed step (p is a multiple of 2h)
    const int probe = (h >= 32) ? hp - 5 : h - 1;                  // skewed offset of key p + h - 1
    const float t = p[probe];
    const bool right = LE ? (t <= x) : (t < x);
    p = right ? p + hp : p;
    if (st == 5 && block_start) *block_start = p;                  // skewed word of the 32-block that holds rank L
  }
  if (block_start && all) *block_start = base + ks_skew(C);
  return all ? base + ks_skew(C) : p;
}

// second launch-bound argument = minimum waves per SIMD: keeps every form whose LDS footprint allows
// four waves per SIMD at <= 128 VGPRs (the compiler otherwise spends 130-175 registers on scheduling
// freedom and occupancy drops to 2-3); no spills result
template <int R, int LG, int DTYPE>
__global__ __launch_bounds__(64 * kWavesPerBlock, (LG <= 32 ? 4 : 2))
void ks_rank_kernel(RankStatsArgs args) {
  static_assert(LG == 8 || LG == 16 || LG == 32 || LG == 64, "lanes per sorted group");
  static_assert(R >= 8 && R <= 32 && (R & (R - 1)) == 0, "registers per lane");
  constexpr int C = R * LG;                    // capacity of the sorted group
  constexpr int PW = 64 / LG;                  // positions per wave
  constexpr int POS_WORDS = ks_rank_pos_words(C);
  constexpr int HIST_OFF = ks_region_words(C); // words from key 0 to bin 0
  constexpr int STEPS = (C == 64) ? 6 : (C == 128) ? 7 : (C == 256) ? 8 : (C == 512) ? 9 : (C == 1024) ? 10 : 11;
  static_assert((1 << STEPS) == C, "capacity must be a power of two");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);
  const int slot = lane / LG;
  float* keys = lds_all + (wave * PW + slot) * POS_WORDS;        // sorted S of this lane's position (skewed)
  unsigned* hist = reinterpret_cast<unsigned*>(keys + HIST_OFF);    // skewed bin k: (#L == k) << 16 | (#U == k)
  const int e0 = gl * R;                                             // first key / bin this lane owns
  const int w0 = ks_skew_rt(e0);                                     // its skewed word (R consecutive words)
  const int w_next = ks_skew_rt(e0 + R);                             // skewed word of key / bin e0 + R

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  if (gl < kKsTail) keys[ks_skew(C) + gl] = inf;

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
    // S = the smaller group (D is symmetric in the groups)
    const bool swap = n1 < n0;
    const int m = swap ? n1 : n0, q = swap ? n0 : n1;
    const void* sig_s = swap ? args.sig1 : args.sig0;
    const void* sig_q = swap ? args.sig0 : args.sig1;
    const int64_t off_s = swap ? o1 : o0, off_q = swap ? o0 : o1;

    float x[R];
    load_packed<R, LG, DTYPE>(x, sig_s, off_s, m, gl);
    seg_sort_any<R, LG>(x, sel, lane);
    store_sorted<R>(keys + w0, x, 0);
    // ties inside S (pads are +inf: excluded by the finite test on the upper element)
    bool s_tie = false;
    {
      const float nxt = lane_next(x[0], inf);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float up = (r == R - 1) ? ((gl == LG - 1) ? inf : nxt) : x[r + 1];
        s_tie = s_tie || (x[r] == up && up < inf);
      }
    }
    // clear this lane's bins e0 .. e0 + R - 1; the last lane also clears bin C
#pragma unroll
    for (int r = 0; r < R; r += 4) *reinterpret_cast<uint4*>(hist + w0 + r) = make_uint4(0, 0, 0, 0);
    if (gl == LG - 1) *reinterpret_cast<uint4*>(hist + ks_skew(C)) = make_uint4(0, 0, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- rank every Q sample into S.  Slots past the end of Q carry FLT_MAX: they rank at L = U = m
    // without ever tying, so the loop needs no validity masks; their count is taken out of bin m below.
    const float big = 3.4028234663852886e38f;
    // full rounds of one 16-byte load per lane, then the remaining < 4*LG samples one per lane
    const int full = q / (4 * LG);
    const int tail = (q - full * (4 * LG) + LG - 1) / LG;
    int full_w = full, tail_w = tail;
    if constexpr (PW > 1) {
      full_w = 0; tail_w = 0;
#pragma unroll
      for (int s = 0; s < PW; ++s) {
        full_w = max(full_w, __builtin_amdgcn_readlane(full, s * LG));
        tail_w = max(tail_w, __builtin_amdgcn_readlane(tail, s * LG));
      }
    } else {
      full_w = __builtin_amdgcn_readfirstlane(full);
      tail_w = __builtin_amdgcn_readfirstlane(tail);
    }
    const bool q_vec = __ballot((off_q & 3) != 0) == 0ull;
    bool any_tie = false;

    // rank NV samples (xq) and add them to the histograms
    auto rank_and_count = [&](auto nv_tag, const float* kbase, const float* xq) {
      constexpr int NV = decltype(nv_tag)::value;
      const float* lp[NV];
      const float* lb32[NV];
      bool tie_here = false;
#pragma unroll
      for (int e = 0; e < NV; ++e) lp[e] = ks_search<C, STEPS, false>(kbase, xq[e], &lb32[e]);
#pragma unroll
      for (int e = 0; e < NV; ++e) tie_here = tie_here || (*lp[e] == xq[e]);      // keys[skew(C)] is +inf
      if (__ballot(tie_here) != 0ull) {          // ties with S: common for 3-dp rounded signals and the synthetic grid
        any_tie = true;
        // a tied sample almost always ties with ONE key: U = L + 1 (the next skewed word is +1, or +5 when
        // L is the last key of its 32-block); only if that next key ties again (duplicates inside S) fall
        // back to the full upper-bound search
        const float* up[NV];
        bool again = false;
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          const bool eq = (*lp[e] == xq[e]);
          // (32-bit LDS offsets: a generic-pointer difference would be computed in 64 bits)
          const unsigned dl = (unsigned)(uintptr_t)lp[e] - (unsigned)(uintptr_t)lb32[e];
          const int step = (dl == 31u * 4u) ? 5 : 1;
          up[e] = eq ? lp[e] + step : lp[e];
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) again = again || (*up[e] == xq[e]);
        if (__ballot(again) != 0ull) {
#pragma unroll
          for (int e = 0; e < NV; ++e) up[e] = ks_search<C, STEPS, true>(kbase, xq[e]);
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
          atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(lp[e])) + HIST_OFF, 0x10000u);
          atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(up[e])) + HIST_OFF, 1u);
        }
      } else {
#pragma unroll
        for (int e = 0; e < NV; ++e)
          atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(lp[e])) + HIST_OFF, 0x10001u);
      }
    };

    // Two schedules.  "own": the LG lanes of a position rank that position's Q (all positions of the wave
    // in lock step; the wave runs as long as its largest Q).  "coop": the 64 lanes rank one position's Q
    // after the other — balanced when the Q sizes of the wave's positions differ (ragged coverage).
    int slots = (full_w * 4 + tail_w) * LG;
    bool coop = false;
    if constexpr (PW > 1) {
      int coop_cost = 0;
#pragma unroll
      for (int sl = 0; sl < PW; ++sl) {
        const int qs = __builtin_amdgcn_readlane(q, sl * LG);
        const int fs = qs / 256;
        coop_cost += fs * 4 + (qs - fs * 256 + 63) / 64;
      }
      coop = coop_cost < full_w * 4 + tail_w;
    }
    if (!coop) {
#pragma unroll 1
      for (int c = 0; c < full_w; ++c) {
        const int idx = c * (4 * LG) + 4 * gl;
        float xq[4] = {big, big, big, big};
        if (c < full) {
          if (q_vec) {
            if constexpr (DTYPE == 0) {
              float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(sig_q) + off_q + idx);
              xq[0] = t.x; xq[1] = t.y; xq[2] = t.z; xq[3] = t.w;
            } else {
              short4 t = *reinterpret_cast<const short4*>(reinterpret_cast<const int16_t*>(sig_q) + off_q + idx);
              xq[0] = (float)t.x; xq[1] = (float)t.y; xq[2] = (float)t.z; xq[3] = (float)t.w;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[e] = load_sample<DTYPE>(sig_q, off_q + idx + e);
          }
        }
        rank_and_count(std::integral_constant<int, 4>{}, keys, xq);
      }
#pragma unroll 1
      for (int c = 0; c < tail_w; ++c) {
        const int idx = full * (4 * LG) + c * LG + gl;
        float xq[1] = {big};
        if (idx < q) xq[0] = load_sample<DTYPE>(sig_q, off_q + idx);
        rank_and_count(std::integral_constant<int, 1>{}, keys, xq);
      }
    } else {
      const int cfull = q / 256;
      slots = (cfull * 4 + (q - cfull * 256 + 63) / 64) * 64;        // per position: what the 64 lanes will process
#pragma unroll 1
      for (int sl = 0; sl < PW; ++sl) {
        const int src = sl * LG;
        const int qs = __builtin_amdgcn_readlane(q, src);
        const unsigned long long sp = (unsigned long long)(uintptr_t)sig_q;
        // (readlane returns int: go through unsigned, or a low word >= 2^31 would sign-extend into the high word)
        auto rl64 = [&](unsigned long long v) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
          return ((unsigned long long)hi << 32) | (unsigned long long)lo;
        };
        const void* sigs = reinterpret_cast<const void*>((uintptr_t)rl64(sp));
        const int64_t offs = (int64_t)rl64((unsigned long long)off_q);
        const float* kb = lds_all + (wave * PW + sl) * POS_WORDS;
        const int fs = qs / 256, ts = (qs - fs * 256 + 63) / 64;
        const bool vec = (offs & 3) == 0;
#pragma unroll 1
        for (int c = 0; c < fs; ++c) {
          const int idx = c * 256 + 4 * lane;
          float xq[4];
          if (vec) {
            if constexpr (DTYPE == 0) {
              float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(sigs) + offs + idx);
              xq[0] = t.x; xq[1] = t.y; xq[2] = t.z; xq[3] = t.w;
            } else {
              short4 t = *reinterpret_cast<const short4*>(reinterpret_cast<const int16_t*>(sigs) + offs + idx);
              xq[0] = (float)t.x; xq[1] = (float)t.y; xq[2] = (float)t.z; xq[3] = (float)t.w;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xq[e] = load_sample<DTYPE>(sigs, offs + idx + e);
          }
          rank_and_count(std::integral_constant<int, 4>{}, kb, xq);
        }
#pragma unroll 1
        for (int c = 0; c < ts; ++c) {
          const int idx = fs * 256 + c * 64 + lane;
          float xq[1] = {big};
          if (idx < qs) xq[0] = load_sample<DTYPE>(sigs, offs + idx);
          rank_and_count(std::integral_constant<int, 1>{}, kb, xq);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) hist[ks_skew_rt(m)] -= (unsigned)(slots - q) * 0x10001u;   // the FLT_MAX slots
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- prefix sums of the histograms and the KS numerator; lane gl owns bins e0 + 1 .. e0 + R
    unsigned h[R];
#pragma unroll
    for (int r = 0; r < R; r += 4) {
      uint4 t = *reinterpret_cast<const uint4*>(hist + w0 + r);         // bins e0 + r .. e0 + r + 3
      if (r > 0) h[r - 1] = t.x;
      h[r] = t.y; h[r + 1] = t.z; h[r + 2] = t.w;
    }
    h[R - 1] = hist[w_next];                                              // bin e0 + R
    unsigned tot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tot += h[r];
    const unsigned cum = seg_exscan_add_u32<LG>(tot, gl) + hist[0];      // both cumulative counts up to bin e0
    unsigned best = 0;
    const bool slow = __ballot(s_tie || any_tie) != 0ull;
    if (!slow) {
      // no ties in this wave: cumL == cumU; D_num = max_{k < m} max(a_k, q - a_k), a_k = cumU(k)*m - k*q.
      // Q samples above every s sit in bin m; clamping k*q at (m-1)*q and the running count at
      // cumU(m-1) makes every bin >= m repeat a_{m-1}.
      const int kq_max = (m - 1) * q;
      int kq = min(e0 * q, kq_max);
      const int cmax = q - (int)(hist[ks_skew_rt(m)] & 0xffffu);         // cumU(m-1)
      int c = min((int)(cum & 0xffffu), cmax);
      int hi = __mul24(c, m) - kq, lo = hi;                              // bin e0 itself: a valid a_k
#pragma unroll
      for (int r = 0; r < R; ++r) {
        c = min(c + (int)(h[r] & 0xffffu), cmax);
        kq = min(kq + q, kq_max);
        const int a = __mul24(c, m) - kq;
        hi = max(hi, a);
        lo = min(lo, a);
      }
      best = (unsigned)max(hi, q - lo);
    } else {
      // general form with the run ends of S as masks
      float s_own[R];
#pragma unroll
      for (int r = 0; r < R; r += 4) {
        float4 t = *reinterpret_cast<const float4*>(keys + w0 + r);
        s_own[r] = t.x; s_own[r + 1] = t.y; s_own[r + 2] = t.z; s_own[r + 3] = t.w;
      }
      const float s_next = keys[w_next];                 // key e0 + R (or the +inf sentinel)
      // Pads are +inf, so "s_{k-1} != s_k" alone marks the run ends: it holds at k = m and fails for k > m.
      int cl = (int)(cum >> 16), cu = (int)(cum & 0xffffu);   // cumL(k-1), cumU(k-1) entering bin k = e0 + 1
      int hi = (gl == 0) ? cu * m : 0, lo = 0;                  // k = 0: (cumU(0), 0)
      int kq = e0 * q;
      int clm = __mul24(cl, m);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float up = (r == R - 1) ? s_next : s_own[r + 1];
        const bool run_end = s_own[r] != up;
        kq += q;
        const int cand_b = clm - kq;                             // v = the S value with upper rank k: cumL(k-1)*m - k*q
        cu += (int)(h[r] & 0xffffu);
        cl += (int)(h[r] >> 16);
        clm = __mul24(cl, m);
        const int cand_a = __mul24(cu, m) - kq;                  // v = largest Q sample with U = k: cumU(k)*m - k*q
        const int ca = run_end ? cand_a : 0, cb = run_end ? cand_b : 0;
        hi = max(hi, max(ca, cb));
        lo = min(lo, min(ca, cb));
      }
      best = (unsigned)max(hi, -lo);
    }
    best = seg_allmax_u32<LG>(best);
    if (valid && gl == 0) args.ks_num[pos] = (m > 0 && q > 0) ? best : 0u;
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nmod
