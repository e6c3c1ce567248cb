// K1, KS-only form (tests mask = KS: BASELINE.json configs[1], "KS + weighted Stouffer").
//
// ks_2samp (myDetect.py:341 -> scipy 1.2.1) needs D = max_v |F0(v) - F1(v)| over the pooled points.
// D is symmetric in the two groups, so call the smaller one S (m samples) and the other Q (q samples):
//   1. S is sorted in registers (R registers x LG lanes, rank_stats_packed.hpp: bitonic network, DPP +
//      v_med3 across lanes) and written to wave-private LDS;
//   2. every sample x of Q finds L(x) = #{s < x} by a branchless 1+log2(C)-step binary search in LDS and
//      looks at the key it lands on: e(x) = (s_{L+1} == x), "x ties with the run of S that starts there";
//   3. ONE ds_add_u32 per sample builds two histograms in the halves of a word: bin L gets
//      (1 << 16) + e(x), i.e. cntL[j] = #{x : L(x) = j} and eq[j] = #{x : x = S[j], the first key of its run};
//   4. with cumL the prefix sum of cntL, the pooled points that can carry the maximum are, per run end k of S
//      (s_k != s_{k+1}; k = #{s <= s_k}),
//        v = s_k                          :  (#{x <= v}, #{s <= v}) = (cumL(k-1), k)
//        v = the largest sample below s_{k+1} :                        (cumU(k),   k),  cumU(k) = cumL(k) - eq[k]
//      (x < s_{k+1}  <=>  L(x) <= k and x != s_{k+1}; the samples equal to s_{k+1} are exactly eq[k], counted at
//      the start of the next run); every other pooled point is dominated by these two, so
//        ks_num = max_k max(|cumL(k-1)*m - k*q|, |cumU(k)*m - k*q|)
//      is the exact integer max|c0*n1 - c1*n0| over the pooled points.  No second search for
//      U(x) = #{s <= x} and no branch on ties in the ranking loop: tie-heavy input (3-decimal events) costs
//      what continuous input costs.
//      Without ties cumL == cumU and the maximum collapses to max_{k<m} max(a_k, q - a_k),
//      a_k = cumL(k)*m - k*q: five VALU instructions per histogram bin.
// Against sorting both groups and walking the pooled sample this removes one sort and the whole sequential
// merge: ~400 instead of ~500 VALU instructions per 200 v 200 position (rocprof SQ_INSTS_VALU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "rank_stats_packed.hpp"
#include "packed_sort_i16.hpp"

namespace nmod {

// what a lane without a sample reads in the unconditional Q loads (see the software pipeline in ks_rank_kernel)
static __device__ const float kKsBig4[4] = {3.4028234663852886e38f, 3.4028234663852886e38f, 3.4028234663852886e38f, 3.4028234663852886e38f};
// ... and in the unconditional S loads: +inf, the pad value of the sort (int16 rows: 32767, replaced in finish())
static __device__ const unsigned kKsInf4[4] = {0x7f800000u, 0x7f800000u, 0x7f800000u, 0x7f800000u};
// ... and of the int16 rows that are sorted as packed keys (packed_sort_i16.hpp): 32767, the pad value of that sort
static __device__ const unsigned kKsPad16[4] = {0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu, 0x7fff7fffu};

// int16 rows of the 16-keys-per-lane classes are sorted two keys per register (v_pk_min_i16 / v_pk_max_i16)
__host__ __device__ constexpr bool ks_packed_sort(int R, int LG, int DTYPE) { return DTYPE == 1 && R == 16 && (LG == 8 || LG == 16); }

// LDS layout: the sorted keys (and the histogram bins) of a position form an R x (LG + 1) matrix, key
// i = R * lane + r at word r * (LG + 1) + lane: row = register, column = lane, one spare column.
//   * the lanes store / load their registers with R ds_write_b32 / ds_read_b32 whose addresses are
//     consecutive across the lanes (conflict-free, immediate offsets);
//   * a power-of-two binary search probes index p + h - 1 with p a multiple of 2h.  While h >= R that is
//     row R - 1 of column (p + h) / R - 1: the first log2(LG) steps walk along one row and different columns
//     are different banks; the last log2(R) steps walk down the column with steps of h rows, and the odd
//     row stride spreads them over the banks.  In the blocked layout every probe of a step shared its index
//     modulo 2h, i.e. its bank (rocprof: 64-85 % of the LDS cycles were bank conflicts);
//   * the search position is a pointer and every probe offset / step a compile-time constant;
//   * key C (rank C: every key is below x) is row 0 of the spare column: the +inf sentinel and bin C live there.
template <int R, int LG>
struct KsLayout {
  static constexpr int C = R * LG;
  static constexpr int ROW = LG + 1;                 // words per row
  static constexpr int REGION = R * ROW;             // keys or bins of one position
  static constexpr int LAST = (R - 1) * ROW + LG - 1;   // key C - 1
  static constexpr int END = LG;                      // key / bin C
  __device__ static __forceinline__ int word(int i) { return __mul24(i & (R - 1), ROW) + i / R; }
};

// words of one position (keys + histogram), padded so that the positions sharing a 32-lane half
// of the wave start LG banks apart (their row walks then use disjoint banks)
__host__ __device__ constexpr int ks_rank_pos_words(int R, int LG) {
  int w = 2 * R * (LG + 1);
  if (LG <= 16) while ((w & 31) != LG) ++w;
  return w;
}

// A pointer selected between a kernel argument and a __device__ constant is generic to the compiler, and a generic
// load (flat_load) counts against lgkmcnt as well as vmcnt: say explicitly that these are global addresses.
template <typename T>
__device__ __forceinline__ T ks_global_load(const void* p) {
  typedef const T __attribute__((address_space(1)))* GlobalPtr;
  return *(GlobalPtr)(p);
}

// rows start at any sample: the 16- / 8-byte loads are declared with the alignment of one sample
typedef float KsF4v __attribute__((ext_vector_type(4)));
typedef short KsS4v __attribute__((ext_vector_type(4)));
typedef KsF4v KsF4 __attribute__((aligned(4)));
typedef KsS4v KsS4 __attribute__((aligned(2)));

// The S rows of a work item, requested long before they are needed (software pipeline of ks_rank_kernel).
// request() issues the same loads in every lane, with no branch and no use of the data — a load that only
// some paths issue, or a select on its result, makes the compiler wait at the request — and finish()
// turns the raw registers into keys (+inf pads) once the data has arrived.  Lane gl takes samples
// k*4*LG + 4*gl .. + 3 (k < R/4); the chunk that holds the end of the row reads the LAST four samples of the
// row instead (the order of S does not matter: it is about to be sorted) and drops the ones the previous lane
// already has; rows shorter than four samples are read one sample at a time by lane 0.
template <int R, int LG, int DTYPE>
struct KsRows {
  using T = typename std::conditional<DTYPE == 0, float, int16_t>::type;
  using V4 = typename std::conditional<DTYPE == 0, KsF4, KsS4>::type;
  static constexpr int NK = R / 4;
  V4 v[NK];
  T s[3];

  __device__ __forceinline__ void request(const void* sig, int64_t off, int n, int gl) {
    const T* row = reinterpret_cast<const T*>(sig) + off;
    const T* dummy = ks_packed_sort(R, LG, DTYPE) ? reinterpret_cast<const T*>(kKsPad16) : reinterpret_cast<const T*>(kKsInf4);
    const bool long_row = n >= 4;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int idx = k * (4 * LG) + 4 * gl;
      const int t = n - idx;
      const T* p = (long_row && t > 0) ? row + ((t >= 4) ? idx : n - 4) : dummy;
      v[k] = ks_global_load<V4>(p);
    }
    // rows shorter than four samples (rare): lane 0 reads them one by one.  These loads are conditional; the caller
    // waits for all row loads with an explicit vmcnt(0) before it starts counting outstanding loads again.
    s[0] = s[1] = s[2] = T(0);
    if (__ballot(!long_row && n > 0) != 0ull) {
#pragma unroll
      for (int e = 0; e < 3; ++e)
        if (!long_row && gl == 0 && e < n) s[e] = ks_global_load<T>(row + e);
    }
  }

  __device__ __forceinline__ void finish(float (&x)[R], int n, int gl) const {
    const float inf = __builtin_inff();
    const bool long_row = n >= 4;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const int idx = k * (4 * LG) + 4 * gl;
      const int t = long_row ? n - idx : 0;          // samples of this chunk: component j is one of them iff j >= 4 - t
      if constexpr (DTYPE == 0) {
        x[4 * k + 0] = v[k].x; x[4 * k + 1] = v[k].y; x[4 * k + 2] = v[k].z; x[4 * k + 3] = v[k].w;   // empty chunks read +inf
        if (__ballot(t > 0 && t < 4) != 0ull) {      // a chunk that holds the end of a row: drop the overlap
          x[4 * k + 0] = (t >= 4 || t <= 0) ? x[4 * k + 0] : inf;
          x[4 * k + 1] = (t >= 3 || t <= 0) ? x[4 * k + 1] : inf;
          x[4 * k + 2] = (t >= 2 || t <= 0) ? x[4 * k + 2] : inf;
        }
      } else {
        x[4 * k + 0] = (t >= 4) ? (float)v[k].x : inf;
        x[4 * k + 1] = (t >= 3) ? (float)v[k].y : inf;
        x[4 * k + 2] = (t >= 2) ? (float)v[k].z : inf;
        x[4 * k + 3] = (t >= 1) ? (float)v[k].w : inf;
      }
    }
    if (__ballot(!long_row && n > 0) != 0ull) {
#pragma unroll
      for (int e = 0; e < 3; ++e) x[e] = (!long_row && gl == 0 && e < n) ? (float)s[e] : x[e];
    }
  }

  // the same rows as packed int16 keys for seg_sort_packed16: two samples per register in load order (the sort does not
  // care which), 32767 pads — empty chunks already read a block of them
  __device__ __forceinline__ void finish_packed(unsigned (&p)[8], int n, int gl) const {
    static_assert(R == 16, "packed rows: 16 keys per lane");
    const bool long_row = n >= 4;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      p[2 * k] = ((unsigned)(unsigned short)v[k].y << 16) | (unsigned)(unsigned short)v[k].x;
      p[2 * k + 1] = ((unsigned)(unsigned short)v[k].w << 16) | (unsigned)(unsigned short)v[k].z;
      const int idx = k * (4 * LG) + 4 * gl;
      const int t = long_row ? n - idx : 0;          // samples of this chunk: component j is one of them iff j >= 4 - t
      if (__ballot(t > 0 && t < 4) != 0ull) {        // a chunk that holds the end of a row: drop the overlap
        if (t == 3) p[2 * k] = (p[2 * k] & 0xffff0000u) | 0x7fffu;           // component 0 belongs to the previous lane
        if (t == 2 || t == 1) p[2 * k] = 0x7fff7fffu;                        // components 0 and 1 do
        if (t == 1) p[2 * k + 1] = (p[2 * k + 1] & 0xffff0000u) | 0x7fffu;   // only component 3 is ours
      }
    }
    if (__ballot(!long_row && n > 0) != 0ull) {      // rows shorter than four samples: lane 0 holds them in s[]
      if (!long_row && gl == 0) {
        const unsigned s0 = (0 < n) ? (unsigned)(unsigned short)s[0] : 0x7fffu;
        const unsigned s1 = (1 < n) ? (unsigned)(unsigned short)s[1] : 0x7fffu;
        const unsigned s2 = (2 < n) ? (unsigned)(unsigned short)s[2] : 0x7fffu;
        p[0] = (s1 << 16) | s0;
        p[1] = (0x7fffu << 16) | s2;
      }
    }
  }
};

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

// branchless binary search: returns the pointer to key L (LE = false: L = #{s < x}) or key U (LE = true:
// U = #{s <= x}); `base` points at key 0.
// FULL = false: the caller knows the array holds at least one +inf pad (fewer than C real keys), so rank C cannot
// occur and the check of the last key is skipped.
template <int R, int LG, bool LE, bool FULL = true>
__device__ __forceinline__ const float* ks_search(const float* base, float x) {
  using Lay = KsLayout<R, LG>;
  const float* p = base;
  bool all = false;
  if constexpr (FULL) {
    const float last = base[Lay::LAST];
    all = LE ? (last <= x) : (last < x);                          // rank C: every key is below x
  }
#pragma unroll
  for (int hc = LG / 2; hc >= 1; hc >>= 1) {                       // h = hc * R keys: along row R - 1
    const float t = p[(R - 1) * Lay::ROW + hc - 1];
    const bool right = LE ? (t <= x) : (t < x);
    p = right ? p + hc : p;
  }
#pragma unroll
  for (int h = R / 2; h >= 1; h >>= 1) {                           // down the column
    const float t = p[(h - 1) * Lay::ROW];
    const bool right = LE ? (t <= x) : (t < x);
    p = right ? p + h * Lay::ROW : p;
  }
  return all ? base + Lay::END : p;
}

// Cooperative schedule (all 64 lanes rank one position): S is wave-uniform, so the probes of the first two search levels are
// scalars — read once per position, the two levels compare against them and take no LDS round trip (VERDICT r4 item 4:
// configs[4] KS-only 15.89 -> 15.39 ms, LDS instructions -15 %, issue utilisation 0.66 -> 0.68; -DNMOD_KS_TOPS=0 is the old loop;
// profiles/r5_ks_tops_ab.txt)
#ifndef NMOD_KS_TOPS
#define NMOD_KS_TOPS 1
#endif
struct KsTops { float hi, lo0, lo1, last; };
template <int R, int LG>
__device__ __forceinline__ KsTops ks_tops(const float* base) {
  using Lay = KsLayout<R, LG>;
  auto uni = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  KsTops t;
  t.hi = uni(base[(R - 1) * Lay::ROW + LG / 2 - 1]);
  t.lo0 = uni(base[(R - 1) * Lay::ROW + LG / 4 - 1]);
  t.lo1 = uni(base[(R - 1) * Lay::ROW + LG / 2 + LG / 4 - 1]);
  t.last = uni(base[Lay::LAST]);
  return t;
}
template <int R, int LG>
__device__ __forceinline__ const float* ks_search_tops(const float* base, float x, const KsTops& t) {   // L = #{s < x}, FULL semantics
  using Lay = KsLayout<R, LG>;
  static_assert(LG >= 4, "two row levels");
  const bool all = t.last < x;
  const bool r1 = t.hi < x;
  const float t2 = r1 ? t.lo1 : t.lo0;
  const bool r2 = t2 < x;
  const float* p = base + (r1 ? LG / 2 : 0) + (r2 ? LG / 4 : 0);
#pragma unroll
  for (int hc = LG / 8; hc >= 1; hc >>= 1) {
    const float v = p[(R - 1) * Lay::ROW + hc - 1];
    p = (v < x) ? p + hc : p;
  }
#pragma unroll
  for (int h = R / 2; h >= 1; h >>= 1) {
    const float v = p[(h - 1) * Lay::ROW];
    p = (v < x) ? p + h * Lay::ROW : p;
  }
  return all ? base + Lay::END : p;
}

template <int LG>
__device__ __forceinline__ double seg_allmax_f64(double v) {
  v = fmax(v, dpp_f64_row(v, 0)); v = fmax(v, dpp_f64_row(v, 1)); v = fmax(v, dpp_f64_row(v, 2));
  if constexpr (LG >= 16) v = fmax(v, dpp_f64_row(v, 3));
  if constexpr (LG >= 32) v = fmax(v, xor16_f64(v));
  if constexpr (LG == 64) v = wave_max_f64(v);
  return v;
}

// fl(c / n) for integers 0 <= c <= n <= 65 535, r = fl(1 / n): correctly rounded (one Newton correction of c * r;
// checked exhaustively on the host: tests/test_abi_and_host.py)
__device__ __forceinline__ double hist_exact_quot(int c, double n, double r) {
  const double dc = (double)c;
  const double q0 = __dmul_rn(dc, r);
  const double rem = __fma_rn(-q0, n, dc);
  return __fma_rn(rem, r, q0);
}

// Mean and sum of squared deviations of a group held as packed int16 keys with 32767 pads (KsRows::finish_packed), before
// the sort: exact sums S1 = sum v (32-bit integer) and S2 = sum v^2 (fp64: every partial sum is an integer < 2^53) over
// the lane's 16 halves, the pads' share taken out by their count, and n S2 - S1^2 — exact for n <= 2 048 — rounded
// once.  Milli-units: mean = S1 / n / 1000, M2 = (n S2 - S1^2) / n * 1e-6.
template <int LG>
__device__ __forceinline__ void seg_moments_packed16(const unsigned (&p)[8], int n, int gl, double& mean, double& m2, double rcp_n = 0.0) {
  int s1 = 0;
  double s2 = 0.0;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int lo = (int)(short)(p[j] & 0xffffu), hi = (int)p[j] >> 16;
    s1 += lo + hi;
    const double dl = (double)lo, dh = (double)hi;
    s2 = __fma_rn(dl, dl, s2);
    s2 = __fma_rn(dh, dh, s2);
  }
  // samples among the lane's 16 keys: chunk k holds samples k*4*LG + 4*gl .. + 3 of a row of n >= 4 (KsRows), lane 0 all of a shorter one
  int valid = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) valid += min(max(n - (k * 4 * LG + 4 * gl), 0), 4);
  valid = (n >= 4) ? valid : ((gl == 0) ? n : 0);
  const int npad = 16 - valid;
  s1 -= 32767 * npad;
  s2 -= 1073676289.0 * (double)npad;                                          // 32767^2
  unsigned v = (unsigned)s1;
  v += (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, (int)v);
  v += (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, (int)v);
  v += (unsigned)dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, (int)v);
  if constexpr (LG >= 16) v += (unsigned)dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, (int)v);
  const double S1 = (double)(int)v;                                           // |S1| <= 256 * 32 768
  const double S2 = seg_allsum_f64<LG>(s2);                                   // <= 256 * 2^30
  const double dn = (double)n;
  const double num = __fma_rn(dn, S2, -S1 * S1);                              // n * M2 in milli-units^2: every term an integer < 2^53
  const double rn = (rcp_n != 0.0) ? rcp_n : 1.0 / dn;
  mean = S1 * rn * 1e-3;
  m2 = num * rn * 1e-6;
}

// second launch-bound argument = minimum waves per SIMD: keeps every form whose LDS footprint allows
// four waves per SIMD (R <= 16; the R = 32 forms are limited to two by their LDS) at <= 128 VGPRs (the compiler otherwise spends 130-175 registers on scheduling
// freedom and occupancy drops to 2-3); no spills result
// FLAGS: also report per position whether a sample of Q tied with a key of S (args.tied).  The float64 front end
// ranks order-preserving float32 images of the samples; where no two images tie the ranks are those of the float64
// samples themselves, and the flagged positions are redone with 64-bit keys (nanomod_hip.hip: detect_f64).  For the KS
// statistic only ties BETWEEN the groups matter: two keys of S with equal images and no sample of Q on them bound a
// pooled point that lies between its neighbours' values of |F0 - F1|.
template <int R, int LG, int DTYPE, bool FLAGS = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, (R <= 16 ? 4 : 2))
void ks_rank_kernel(RankStatsArgs args) {
  static_assert(LG == 8 || LG == 16 || LG == 32 || LG == 64, "lanes per sorted group");
  static_assert(R >= 8 && R <= 32 && (R & (R - 1)) == 0, "registers per lane");
  constexpr int PW = 64 / LG;                  // positions per wave
  using Lay = KsLayout<R, LG>;
  constexpr int ROW = Lay::ROW;
  constexpr int POS_WORDS = ks_rank_pos_words(R, LG);
  constexpr int HIST_OFF = Lay::REGION;        // words from key 0 to bin 0
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);
  const int slot = lane / LG;
  float* keys = lds_all + (wave * PW + slot) * POS_WORDS;        // sorted S of this lane's position (KsLayout)
  unsigned* hist = reinterpret_cast<unsigned*>(keys + HIST_OFF);    // bin j: #{x : L(x) = j} << 16 | #{x : x = S[j], L(x) = j}, same layout
  const int e0 = gl * R;                                             // first key / bin this lane owns

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  for (int r = gl; r < R; r += LG) keys[r * ROW + Lay::END] = inf;   // the spare column: key C (and beyond) = +inf
  // fixed-stride batches: every position has the same sizes; fl(1/m) and fl(1/q) are taken once per block and parked
  // in LDS behind the positions' words
  const bool uniform = args.stride0 > 0 && args.stride1 > 0;
  double* recip = reinterpret_cast<double*>(lds_all + kWavesPerBlock * PW * POS_WORDS);
  if (uniform && threadIdx.x == 0) {
    const int64_t a0 = args.stride0 < args.stride1 ? args.stride0 : args.stride1;
    const int64_t a1 = args.stride0 < args.stride1 ? args.stride1 : args.stride0;
    recip[0] = 1.0 / (double)a0; recip[1] = 1.0 / (double)a1;
  }
  __syncthreads();

  int64_t count = args.npos;
  const int32_t* list = nullptr;
  if (args.alt_gates != nullptr && args.alt_gates[args.class_id] != 0) {     // what rank_count_wide_kernel left of the class
    count = args.alt_meta[args.class_id];
    list = args.alt_list + args.alt_meta[kClassStride + args.class_id];
  } else if (args.pos_list) {
    count = args.class_meta[args.class_id];
    list = args.pos_list + args.class_meta[kClassStride + args.class_id];
  }
  const int64_t items = (count + PW - 1) / PW;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerBlock;

  // One work item = the PW positions of a wave.  S = the smaller group (D is symmetric in the groups).
  struct Item { bool valid, swap; int m, q; int64_t pos, off_s, off_q; };
  auto describe = [&](int64_t it) {
    Item d;
    const int64_t li = it * PW + slot;
    d.valid = it < items && li < count;
    d.pos = d.valid ? (list ? (int64_t)list[li] : li) : 0;
    // (positions are < 2^31 and strides <= 65 535: one 32 x 32 -> 64-bit multiply each)
    int64_t o0 = 0, o1 = 0; int n0 = 0, n1 = 0;
    if (d.valid) {
      if (args.stride0 > 0) { o0 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride0); n0 = (int)args.stride0; }
      else { o0 = args.off0[d.pos]; n0 = (int)(args.off0[d.pos + 1] - o0); }
      if (args.stride1 > 0) { o1 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride1); n1 = (int)args.stride1; }
      else { o1 = args.off1[d.pos]; n1 = (int)(args.off1[d.pos + 1] - o1); }
    }
    d.swap = n1 < n0;
    d.m = d.swap ? n1 : n0; d.q = d.swap ? n0 : n1;
    d.off_s = d.swap ? o1 : o0; d.off_q = d.swap ? o0 : o1;
    return d;
  };
  const float big = 3.4028234663852886e38f;
  // one 16-byte (f32) / 8-byte (i16) load per lane: samples idx .. idx + 3 of a Q row when `have`.  The load is
  // issued by EVERY lane (lanes without samples read a block of FLT_MAX): a load that only some paths issue
  // makes the number of outstanding loads unknown to the compiler, and every later wait becomes vmcnt(0).
  // For the same reason the raw registers are converted only where they are used (q4_values), a round later.
  using Q4Raw = typename std::conditional<DTYPE == 0, KsF4, KsS4>::type;
  using Q1Raw = typename std::conditional<DTYPE == 0, float, int16_t>::type;
  auto load_q4 = [&](const void* sig, int64_t off, int idx, bool have) -> Q4Raw {
    using T = Q1Raw;
    const T* src = have ? reinterpret_cast<const T*>(sig) + off + idx : reinterpret_cast<const T*>(kKsBig4);
    return ks_global_load<Q4Raw>(src);
  };
  auto q4_values = [&](float (&xq)[4], const Q4Raw& t, bool have) {
    if constexpr (DTYPE == 0) {
      (void)have;                                   // the dummy block already holds FLT_MAX
      xq[0] = t.x; xq[1] = t.y; xq[2] = t.z; xq[3] = t.w;
    } else {
      xq[0] = have ? (float)t.x : big; xq[1] = have ? (float)t.y : big;
      xq[2] = have ? (float)t.z : big; xq[3] = have ? (float)t.w : big;
    }
  };
  auto load_q1 = [&](const void* sig, int64_t off, int idx, bool have) -> Q1Raw {
    using T = Q1Raw;
    return ks_global_load<T>(have ? reinterpret_cast<const T*>(sig) + off + idx : reinterpret_cast<const T*>(kKsBig4));
  };
  auto q1_value = [&](Q1Raw v, bool have) -> float {
    if constexpr (DTYPE == 0) { (void)have; return v; }
    else return have ? (float)v : big;
  };

  // Software pipeline: the S rows of the NEXT item are requested while this item's Q is ranked, and every
  // round of Q samples is requested before the previous round is ranked — otherwise each item pays five
  // dependent HBM round trips (measured: the load-only skeleton of this kernel ran at 3.9 TB/s).
  Item cur = describe(wave_global);
  constexpr bool PACKED = ks_packed_sort(R, LG, DTYPE);
  float x[R];
  unsigned pk[8];                                  // PACKED: the item's S rows as packed int16 keys (x is filled by the sort)
  {
    KsRows<R, LG, DTYPE> first;
    first.request(cur.swap ? args.sig1 : args.sig0, cur.off_s, cur.m, gl);
    if constexpr (PACKED) first.finish_packed(pk, cur.m, gl);
    else first.finish(x, cur.m, gl);
  }
  // vmcnt(0) (expcnt / lgkmcnt untouched): S rows are waited for here and at the bottom of the loop, where they
  // have long arrived — vmcnt retires in order, so a wait placed at the sort would also wait for the Q round
  // requested just before it
  __builtin_amdgcn_s_waitcnt(0x0F70);

  for (int64_t it = wave_global; it < items; it += wave_stride) {
    const bool valid = cur.valid;
    const int64_t pos = cur.pos;
    const int m = cur.m, q = cur.q;
    const void* sig_q = cur.swap ? args.sig0 : args.sig1;
    const int64_t off_q = cur.off_q;

    // ---- the ranking schedule (needs only q) and the first round of Q, requested before the sort.
    // Slots past the end of Q carry FLT_MAX: they rank at L = U = m without ever tying, so the ranking loop needs
    // no validity masks; their count is taken out of bin m below.
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
    // requests that the sort hides: the next item's S rows, then this item's first rounds of Q
    const Item nxt = describe(it + wave_stride);
    KsRows<R, LG, DTYPE> rows_next;
    rows_next.request(nxt.swap ? args.sig1 : args.sig0, nxt.off_s, nxt.m, gl);
    // (PIPE_COOP: in the coop schedule these are the first rounds of the wave's FIRST position, 64 lanes wide)
    constexpr bool PIPE_COOP = LG == 8;             // the forms ragged coverage lands in; the others keep the plain coop loop
    auto row_of_slot = [&](int sl, const void*& sg, int64_t& of, int& qn) {
      const int src = sl * LG;
      // (readlane returns int: go through unsigned, or a low word >= 2^31 would sign-extend into the high word)
      auto rl64 = [&](unsigned long long v) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src);
        return ((unsigned long long)hi << 32) | (unsigned long long)lo;
      };
      sg = reinterpret_cast<const void*>((uintptr_t)rl64((unsigned long long)(uintptr_t)sig_q));
      of = (int64_t)rl64((unsigned long long)off_q);
      qn = __builtin_amdgcn_readlane(q, src);
    };
    const void* sig_c = sig_q; int64_t off_c = off_q; int q_c = q;          // coop: the position being ranked by the wave
    int idx_a = 4 * gl, idx_t = full * (4 * LG) + gl;
    bool have_a = !coop && 0 < full, have_t = !coop && idx_t < q;
    if constexpr (PIPE_COOP) {
      if (coop) {
        row_of_slot(0, sig_c, off_c, q_c);
        idx_a = 4 * lane; idx_t = (q_c / 256) * 256 + lane; have_a = q_c >= 256; have_t = idx_t < q_c;
      }
    }
    Q4Raw ra = load_q4(sig_c, off_c, idx_a, have_a);                          // the round being ranked next (raw)
    Q1Raw rt = load_q1(sig_c, off_c, idx_t, have_t);                          // first one-per-lane round

#if !(defined(NMOD_EXP) && (NMOD_EXP & 1))
    if constexpr (PACKED) {
      // two keys per register through the network, then the sorted keys as floats; the top C - m keys of the position
      // are its pads (a sample may equal the pad value 32767: only the key INDEX tells them apart)
      seg_sort_packed16<LG>(pk, lane);
      unpack_sorted16<LG>(pk, x);
    } else {
      seg_sort_any<R, LG>(x, sel, lane);
    }
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) keys[r * ROW + gl] = x[r];
    if constexpr (PACKED) {                        // (the pads go to LDS as +inf: predicated stores, one compare per key)
      const int real = m - e0;                     // keys of this lane that are samples
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (r >= real) keys[r * ROW + gl] = inf;
    }
    // clear this lane's bins e0 .. e0 + R - 1; the last lane also clears bin C
#pragma unroll
    for (int r = 0; r < R; ++r) hist[r * ROW + gl] = 0u;
    if (gl == LG - 1) hist[Lay::END] = 0u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- rank every Q sample into S
    // a sorted group that fills its capacity exactly has no +inf pad: only then can a sample rank above every key
    const bool s_full = __ballot(m == Lay::C) != 0ull;

    // rank NV samples (xq) and count them: bin L(x) += 1 << 16, + 1 when x equals the key it landed on (key C is +inf)
    auto rank_and_count = [&](auto nv_tag, auto full_tag, const float* kbase, const float* xq) {
      constexpr int NV = decltype(nv_tag)::value;
      constexpr bool FULL = decltype(full_tag)::value;
      const float* lp[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) lp[e] = ks_search<R, LG, false, FULL>(kbase, xq[e]);
      unsigned inc[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) inc[e] = (*lp[e] == xq[e]) ? 0x10001u : 0x10000u;
#pragma unroll
      for (int e = 0; e < NV; ++e) atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(lp[e])) + HIST_OFF, inc[e]);
    };

    [[maybe_unused]] auto rank_and_count_tops = [&](auto nv_tag, const float* kbase, const KsTops& tops, const float* xq) {
      constexpr int NV = decltype(nv_tag)::value;
      const float* lp[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) lp[e] = ks_search_tops<R, LG>(kbase, xq[e], tops);
      unsigned inc[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) inc[e] = (*lp[e] == xq[e]) ? 0x10001u : 0x10000u;
#pragma unroll
      for (int e = 0; e < NV; ++e) atomicAdd(reinterpret_cast<unsigned*>(const_cast<float*>(lp[e])) + HIST_OFF, inc[e]);
    };

    // everything requested before the sort has arrived; from here the number of outstanding loads is known
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (!coop) {
#pragma unroll 1
      for (int c = 0; c < full_w; ++c) {
        const Q4Raw rb = load_q4(sig_q, off_q, (c + 1) * (4 * LG) + 4 * gl, c + 1 < full);
        float xa[4];
        q4_values(xa, ra, c < full);
#if !(defined(NMOD_EXP) && (NMOD_EXP & 2))
        if (s_full) rank_and_count(std::integral_constant<int, 4>{}, std::true_type{}, keys, xa);
        else rank_and_count(std::integral_constant<int, 4>{}, std::false_type{}, keys, xa);
#else
        if (xa[0] + xa[1] + xa[2] + xa[3] == 12345.f) atomicAdd(hist, 1u);
#endif
        ra = rb;
      }
#pragma unroll 1
      for (int c = 0; c < tail_w; ++c) {
        float xq[1] = {q1_value(rt, full * (4 * LG) + c * LG + gl < q)};
        const int idx = full * (4 * LG) + (c + 1) * LG + gl;
        rt = load_q1(sig_q, off_q, idx, idx < q);
        if (s_full) rank_and_count(std::integral_constant<int, 1>{}, std::true_type{}, keys, xq);
        else rank_and_count(std::integral_constant<int, 1>{}, std::false_type{}, keys, xq);
      }
    } else {
      const int cfull = q / 256;
      slots = (cfull * 4 + (q - cfull * 256 + 63) / 64) * 64;        // per position: what the 64 lanes will process
      if constexpr (PIPE_COOP) {
      // software pipeline over (position, round): the first rounds of the NEXT position are requested before this
      // position's are ranked, and every round before the previous one is ranked (one exposed HBM round trip per
      // round otherwise: 0.56 issue utilisation on configs[4])
#pragma unroll 1
      for (int sl = 0; sl < PW; ++sl) {
        const void* sig_n = sig_c; int64_t off_n = off_c; int q_n = 0;
        if (sl + 1 < PW) row_of_slot(sl + 1, sig_n, off_n, q_n);
        const int nfs = q_n / 256;
        const Q4Raw na = load_q4(sig_n, off_n, 4 * lane, nfs > 0);
        const Q1Raw nt = load_q1(sig_n, off_n, nfs * 256 + lane, nfs * 256 + lane < q_n);
        const float* kb = lds_all + (wave * PW + sl) * POS_WORDS;
        const int fs = q_c / 256, ts = (q_c - fs * 256 + 63) / 64;
#if NMOD_KS_TOPS
        const KsTops tops = ks_tops<R, LG>(kb);
#endif
#pragma unroll 2
        for (int c = 0; c < fs; ++c) {
          // (past the last round: the last round again — a load that is never used, from an address that exists)
          const Q4Raw rb = load_q4(sig_c, off_c, min(c + 1, fs - 1) * 256 + 4 * lane, true);
          float xq[4];
          q4_values(xq, ra, true);
#if NMOD_KS_TOPS
          rank_and_count_tops(std::integral_constant<int, 4>{}, kb, tops, xq);
#else
          rank_and_count(std::integral_constant<int, 4>{}, std::true_type{}, kb, xq);
#endif
          ra = rb;
        }
#pragma unroll 1
        for (int c = 0; c < ts; ++c) {
          const int idx_now = fs * 256 + c * 64 + lane;
          const Q1Raw r1 = load_q1(sig_c, off_c, idx_now + 64, idx_now + 64 < q_c);
          float xq[1] = {q1_value(rt, idx_now < q_c)};
#if NMOD_KS_TOPS
          rank_and_count_tops(std::integral_constant<int, 1>{}, kb, tops, xq);
#else
          rank_and_count(std::integral_constant<int, 1>{}, std::true_type{}, kb, xq);
#endif
          rt = r1;
        }
        sig_c = sig_n; off_c = off_n; q_c = q_n; ra = na; rt = nt;
      }
      } else {
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
#pragma unroll 1
        for (int c = 0; c < fs; ++c) {
          const int idx = c * 256 + 4 * lane;
          float xq[4];
          q4_values(xq, load_q4(sigs, offs, idx, true), true);
          rank_and_count(std::integral_constant<int, 4>{}, std::true_type{}, kb, xq);
        }
#pragma unroll 1
        for (int c = 0; c < ts; ++c) {
          const int idx = fs * 256 + c * 64 + lane;
          float xq[1] = {q1_value(load_q1(sigs, offs, idx, idx < qs), idx < qs)};
          rank_and_count(std::integral_constant<int, 1>{}, std::true_type{}, kb, xq);
        }
      }
      }   // !PIPE_COOP
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (gl == 0) { const unsigned extra = (unsigned)(slots - q); hist[Lay::word(m)] -= extra << 16; }   // the FLT_MAX slots: L = m, never equal to a key
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- prefix sums of the histograms and the KS numerator; lane gl owns bins e0 + 1 .. e0 + R
    unsigned h[R];
#pragma unroll
    for (int r = 0; r < R - 1; ++r) h[r] = hist[(r + 1) * ROW + gl];     // bins e0 + 1 .. e0 + R - 1
    h[R - 1] = hist[gl + 1];                                              // bin e0 + R: row 0 of the next column
    const unsigned h0 = hist[0];
    unsigned tot = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) tot += h[r];
    const unsigned cum = seg_exscan_add_u32<LG>(tot, gl) + h0;           // cumL(e0) << 16 | (ties counted up to bin e0)
    unsigned best = 0;
    // Which evaluation: the general one as soon as a sample tied with S (a low half of the wave's bins is not 0);
    // otherwise look for ties INSIDE S — only then, from the keys in LDS (the general path reloads them anyway): the
    // smallest gap between neighbours is exactly 0.  Between two +inf pads the gap is NaN, which min() drops.  (A gap
    // that underflows to 0 would only send the wave down the general path.)
    float s_own[R];
#pragma unroll
    for (int r = 0; r < R; ++r) s_own[r] = keys[r * ROW + gl];
    const float s_next = keys[gl + 1];                   // key e0 + R (or the +inf sentinel)
    const bool tie_lane = ((tot | h0) & 0xffffu) != 0u;  // a sample equal to one of this lane's keys (or to key 0)
    bool slow = __ballot(tie_lane) != 0ull;
    if (!slow) {
      float gap = inf;
#pragma unroll
      for (int r = 0; r < R; ++r) gap = fminf(gap, ((r == R - 1) ? s_next : s_own[r + 1]) - s_own[r]);
      slow = __ballot(gap == 0.0f) != 0ull;
    }
    if (!slow) {
      // no ties in this wave: cumL == cumU; D_num = max_{k < m} max(a_k, q - a_k), a_k = cumL(k)*m - k*q.
      // Q samples above every s sit in bin m; clamping k*q at (m-1)*q and the running count at
      // cumL(m-1) makes every bin >= m repeat a_{m-1}.
      // A lane owns the candidates of ITS bins k = e0+1 .. e0+R: (cumL(k), k) = a_k and (cumL(k-1), k) = a_{k-1} - q, so
      // its maximum runs over a_{e0+1} .. a_{e0+R} and its minimum over a_{e0} .. a_{e0+R-1} (lane 0 also owns
      // (cumL(0), 0) = a_0): no candidate is seen by two lanes, and the lanes that reach the position's maximum are
      // exactly the ones whose bins the float-form pass has to look at.
      const int kq_max = __mul24(m - 1, q);                              // (24-bit operands: m <= 2 048, q <= 65 535)
      int kq = min(__mul24(e0, q), kq_max);
      const int cmax = q - (int)(hist[Lay::word(m)] >> 16);              // cumL(m-1)
      int c = min((int)(cum >> 16), cmax);
      int a = __mul24(c, m) - kq;                                        // a_{e0}
      int hi = (gl == 0) ? a : -(1 << 30), lo = 1 << 30;
      // (the running counts go back to LDS in place of the bins — stores only, no extra arithmetic: the float-form pass
      // below reads cumL(k-1) and cumL(k) of the bins it examines from there)
      if (gl == 0) hist[0] = (unsigned)c;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        lo = min(lo, a);
        c = min(c + (int)(h[r] >> 16), cmax);
        if (r < R - 1) hist[(r + 1) * ROW + gl] = (unsigned)c; else hist[gl + 1] = (unsigned)c;
        kq = min(kq + q, kq_max);
        a = __mul24(c, m) - kq;
        hi = max(hi, a);
      }
      best = (unsigned)max(max(hi, q - lo), 0);
    } else {
      // general form with the run ends of S as masks
      // Pads are +inf, so "s_{k-1} != s_k" alone marks the run ends: it holds at k = m and fails for k > m.
      int cl = (int)(cum >> 16);                                    // cumL(k-1) entering bin k = e0 + 1
      // k = 0: (cumU(0), 0), cumU(0) = the samples below key 0 = cntL[0] - eq[0]
      int hi = (gl == 0) ? __mul24((int)(h0 >> 16) - (int)(h0 & 0xffffu), m) : 0, lo = 0;
      int nkq = -__mul24(e0, q);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float up = (r == R - 1) ? s_next : s_own[r + 1];
        const bool run_end = s_own[r] != up;
        nkq -= q;
        const int cand_b = __mul24(cl, m) + nkq;                   // v = the S value with upper rank k: cumL(k-1)*m - k*q
        const int clp = cl;
        cl += (int)(h[r] >> 16);                                   // cumL(k)
        const int cu = cl - (int)(h[r] & 0xffffu);                 // cumU(k) = cumL(k) - #{x = s_{k+1}}
        // (the bin's word becomes cumL(k-1) << 16 | cumU(k) for the float-form pass: two 16-bit stores, no arithmetic)
        {
          unsigned short* w16 = reinterpret_cast<unsigned short*>(hist + ((r < R - 1) ? (r + 1) * ROW + gl : gl + 1));
          w16[0] = (unsigned short)cu; w16[1] = (unsigned short)clp;
        }
        const int cand_a = __mul24(cu, m) + nkq;                   // v = the largest sample below s_{k+1}: cumU(k)*m - k*q
        const int ca = run_end ? cand_a : 0, cb = run_end ? cand_b : 0;
        hi = max(hi, max(ca, cb));
        lo = min(lo, min(ca, cb));
      }
      best = (unsigned)max(hi, -lo);
    }
    const unsigned lbest = best;
    best = seg_allmax_u32<LG>(best);
    // ---- ks_2samp forms D as max |fl(c0/n0) - fl(c1/n1)| over the pooled points.  A larger integer numerator always
    // gives a larger float value (they differ by >= 1/(n0 n1) >> ulp), so the float form is evaluated only for the
    // candidates that reach the integer maximum — the maximum of those is the reference's D bit for bit.  The lanes of a
    // position take the R bins of one such lane at a time, BPL consecutive bins each, from the table the evaluation
    // left in LDS: word k = cumL(k) when the wave saw no tie, cumL(k-1) << 16 | cumU(k) otherwise.
    double dmax = 0.0;
    if (!args.ks_rational_d) {
      constexpr int BPL = (R + LG - 1) / LG;
      const double dm = (double)m, dq = (double)q;
      double rm, rq;
      if (uniform) { rm = recip[0]; rq = recip[1]; }
      else { rm = 1.0 / dm; rq = 1.0 / dq; }
      const int seg_base = lane & ~(LG - 1);
      const unsigned long long hits = __ballot(lbest == best && best != 0u);
      using Mine = typename std::conditional<LG == 64, unsigned long long, unsigned>::type;
      Mine mine = (Mine)(hits >> seg_base);
      if constexpr (LG < 32) mine &= (Mine)((1u << LG) - 1u);
#if defined(NMOD_EXP) && (NMOD_EXP & 4)
      mine = 0;
#endif
      // the candidate (cumU(0), 0) belongs to lane 0 of the position
      if (gl == 0 && (mine & (Mine)1)) {
        const int cu0 = (int)(h0 >> 16) - (int)(h0 & 0xffffu);
        if ((unsigned)__mul24(cu0, m) == best) dmax = hist_exact_quot(cu0, dq, rq);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll 1
      while (__ballot(mine != (Mine)0) != 0ull) {
        const bool act = mine != (Mine)0;
        const int hl = act ? (__ffsll((long long)mine) - 1) : 0;              // the lane of this position whose bins are examined
        mine &= mine - (Mine)1;
#pragma unroll
        for (int j = 0; j < BPL; ++j) {
          const int rr = gl * BPL + j;                                        // bin k = hl * R + rr + 1
          const bool in = act && rr < R;
          const int k = hl * R + rr + 1;
          // key k - 1 / word k - 1 = row rr of column hl; key / word k = row rr + 1 of column hl, or row 0 of column hl + 1
          const int wp = in ? rr * ROW + hl : 0;
          const int wk = in ? ((rr + 1 < R) ? (rr + 1) * ROW + hl : hl + 1) : 0;
          const unsigned tp = hist[wp], tk = hist[wk];
          const bool run_end = keys[wp] != keys[wk];
          const int clp = slow ? (int)(tk >> 16) : (int)tp;                   // cumL(k-1)
          const int cu = slow ? (int)(tk & 0xffffu) : (int)tk;                // cumU(k)
          const int nkq = -__mul24(k, q);
          const int cand_b = __mul24(clp, m) + nkq;                           // (cumL(k-1), k)
          const int cand_a = __mul24(cu, m) + nkq;                            // (cumU(k), k)
          const bool hb = in && run_end && (unsigned)abs(cand_b) == best;
          const bool ha = in && run_end && (unsigned)abs(cand_a) == best;
          const double fk = hist_exact_quot(k, dm, rm);
          const double db = fabs(fk - hist_exact_quot(clp, dq, rq));
          const double da = fabs(fk - hist_exact_quot(cu, dq, rq));
          dmax = hb ? fmax(dmax, db) : dmax;
          dmax = ha ? fmax(dmax, da) : dmax;
        }
      }
      dmax = seg_allmax_f64<LG>(dmax);
    }
    if (valid && gl == 0) {
      args.ks_num[pos] = (m > 0 && q > 0) ? best : 0u;
      args.ks_d_ref[pos] = (m > 0 && q > 0) ? dmax : 0.0;
    }
    if constexpr (FLAGS) {
      const unsigned t = seg_allmax_u32<LG>(tie_lane ? 1u : 0u);   // a sample of Q tied with a key of S anywhere in the position
      if (valid && gl == 0) args.tied[pos] = (uint8_t)t;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if constexpr (PACKED) rows_next.finish_packed(pk, nxt.m, gl);
    else rows_next.finish(x, nxt.m, gl);
    cur = nxt;
  }
}

}  // namespace nmod
