// K1, counting form (round 5) — all-tests mode for positions whose samples are EVENT-LIKE: both groups of at most 255
// samples, every sample on the milli-unit grid of stored NanoMod events (int16 rows are by construction; float32 rows
// are tested per sample with grid_key, rank_hist.hpp), and all of the position's keys inside a window of 2 048
// milli-units.  Real events are that: norm_mean is a 3-decimal value (myRefBaseSignalAnnotation.py:1108) and the reads
// of one position sit a few tenths of a unit around the level of its k-mer — most samples of a position tie with another
// one, which is where the sorting forms (rank_hist.hpp) pay most (clean-up phases, general tie sweeps).
//
// Nothing is sorted.  With the keys as small integers u = k - kmin in [0, 2047], the empirical distribution function of a
// group IS a table: one byte per value,
//     build   count[u] += 1 for every sample of the group      (one ds_add_u32 of 1 << 8 (u & 3) per sample)
//     scan    cum[u] = #{x <= u}                                (in place; SWAR prefix sums inside a word, a carry along the
//                                                                 lane's 32 words, one 16-lane exclusive scan of the lane totals)
//     lookup  cum[u] and cum[u - 1] = #{x < u} for every sample of BOTH groups (two ds_read_u8 each)
// first for group 1 (table A), then for group 2 (table B) in the same LDS words.  From the four numbers per sample v
// (A[v], A[v-1], B[v], B[v-1]) everything getKStest needs (myDetect.py:327-343) follows exactly:
//   * KS            max over the pooled points of |A[v] n1 - B[v] n0|: every pooled point is the value of some sample, so
//                   the maximum over all samples of both groups is ks_2samp's maximum over the pooled sample; its float form
//                   max |fl(c0/n0) - fl(c1/n1)| is evaluated for the samples that reach the integer maximum (as in
//                   rank_hist.hpp: a larger numerator always gives a larger float value), bit for bit ks_2samp's D.
//   * Mann-Whitney  sum over x in group 2 of (A[x-1] + A[x]) = sum (#{a < x} + #{a <= x});  mwu_s = 2 n0 n1 - that.
//   * tie term      sum over pooled tie groups of t^3 - t = sum over ALL samples e of (t(e)^2 - 1), t(e) = the size of e's
//                   tie group = (A[v] - A[v-1]) + (B[v] - B[v-1]).
//   * Welch         exact integer moment sums (int16 rows) / shifted fp64 sums of the float32 values (float32 rows).
// 16 lanes per position, four positions per wave, up to 16 samples per lane and group; a position's table is 16 blocks of
// 36 words — lane gl scans block gl: 4 pad words + 32 words = 128 counters — so that the lanes' 16-byte reads and writes of
// the scan fall on different banks (a 32-word stride would put all 16 lanes on four banks), and byte 15 of a block's pad
// holds the block's carry-in, which makes cum[u - 1] the byte before cum[u] also at a block's first value.  A slot without
// a sample looks up two zero bytes of the first pad and adds nothing anywhere.  2 304 B per position, 36.9 KB per block of
// four waves: four blocks per CU.
// Positions that do not qualify (a group beyond 255 or below 4 samples, a float32 sample off the grid, a range beyond the
// window) are flagged in RankStatsArgs::cnt_done and taken by rank_hist_kernel right after (its AFTER_COUNT instances).
// Whether a batch is event-like at all is decided on the device by cnt_probe_kernel on a sample of positions before
// this kernel runs: continuous signals never pay for the attempt.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ks_rank.hpp"
#include "rank_hist.hpp"     // grid_key, pos_allsum_u32

namespace nmod {

constexpr int kCntLanes = 16;                                   // lanes per position
constexpr int kCntWindow = 2048;                                // values of the direct-address window
constexpr int kCntBlockWords = 36;                              // a lane's block: 4 pad words + 32 words of byte counters
constexpr int kCntTail = 16;                                    // samples far from the position's level (outliers) it may hold: one per lane
constexpr int kCntTableWords = kCntLanes * kCntBlockWords;      // 576
constexpr int kCntPosWords = kCntTableWords + kCntTail;         // + the tail list: 592 words, 37.9 KB per block of four waves: four blocks per CU
constexpr int kCntMaxN = 255;                                   // byte counters and byte prefix sums
constexpr int kCntMinN = 4;                                     // (KsRows reads shorter rows through a conditional path)
__host__ __device__ constexpr size_t rank_count_lds_bytes() { return (size_t)kWavesPerBlock * 4 * kCntPosWords * 4 + 16; }

typedef __attribute__((address_space(3))) unsigned* CntLdsU32;
typedef short CntS2 __attribute__((ext_vector_type(2)));
typedef unsigned short CntU2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const unsigned char* CntLdsU8;
// lo[h] = cum[u - 1], hi[h] = cum[u] of the two slots whose byte addresses (of cum[u - 1]) are the halves of ap0 / ap1
__device__ __forceinline__ void cnt_lookup_chunk(unsigned ap0, unsigned ap1, unsigned (&lo)[2], unsigned (&hi)[2]) {
  unsigned b[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const unsigned ap = h ? ap1 : ap0;
    const CntLdsU8 p0 = (CntLdsU8)(uintptr_t)(ap & 0xffffu), p1 = (CntLdsU8)(uintptr_t)(ap >> 16);
    b[4 * h] = p0[0]; b[4 * h + 1] = p0[1]; b[4 * h + 2] = p1[0]; b[4 * h + 3] = p1[1];
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) { lo[h] = b[4 * h] | (b[4 * h + 2] << 16); hi[h] = b[4 * h + 1] | (b[4 * h + 3] << 16); }
}

#ifndef NMOD_CNT_SKIP
#define NMOD_CNT_SKIP 0
#endif
#ifndef NMOD_CNT_WAVES
#define NMOD_CNT_WAVES 4
#endif
#ifndef NMOD_CNT_TAILS
#define NMOD_CNT_TAILS 1     // 1: the probe lets batches with a few outliers per position through (the kernel's tail path); 0: round 5's probe
#endif

// min / max over the 16 lanes of a position, in every lane
__device__ __forceinline__ int cnt_allmin_i32(int v) {
  v = min(v, dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, v));
  v = min(v, dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, v));
  v = min(v, dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, v));
  return min(v, dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, v));
}
__device__ __forceinline__ int cnt_allmax_i32(int v) {
  v = max(v, dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, v));
  v = max(v, dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, v));
  v = max(v, dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, v));
  return max(v, dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, v));
}
__device__ __forceinline__ int cnt_allsum_i32(int v) { return (int)pos_allsum_u32<16>((unsigned)v); }

// DTYPE 2: float32 keys that are whole numbers — the float64 front end's keys of a position whose samples are all k / 1000.0
// (nanomod_hip.hip: f64_encode_kernel writes (float)k).  Any whole-number keys order and tie like their integers, whatever
// their unit; the Welch moments of such a batch come from the float64 samples (f64_moments_kernel), not from here.
__device__ __forceinline__ bool cnt_int_key(float x, int& k) {
  k = (int)x;                                            // (v_cvt_i32_f32 saturates; NaN -> 0, rejected by the compare)
  return (float)k == x && __builtin_fabsf(x) <= 32767.0f;
}

// ---- is this batch event-like?  One block; wave w looks at sampled positions w, w + 16, ...: sizes, grid, range — the
// very test rank_count_kernel applies per position.  gate[0] = 1 when at least 7 of 8 sampled positions qualify.
struct CntProbeArgs {
  const void* sig0; const void* sig1; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1; int64_t npos;
  const int32_t* pos_list; const int32_t* class_meta; int32_t class_id; int32_t dtype; int32_t* gate;
};
constexpr int kCntProbeSamples = 64;
constexpr int kCntProbeWindow = 1536;                   // (the kernel's: kCntWindow = 2 048)

template <int DTYPE>
__global__ __launch_bounds__(1024) void cnt_probe_kernel(CntProbeArgs a) {
  __shared__ int fits, seen, far;
  if (threadIdx.x == 0) { fits = 0; seen = 0; far = 0; }
  __syncthreads();
  int64_t count = a.npos;
  const int32_t* list = nullptr;
  if (a.pos_list) { count = a.class_meta[a.class_id]; list = a.pos_list + a.class_meta[kClassStride + a.class_id]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nsamp = count < kCntProbeSamples ? count : kCntProbeSamples;
  for (int64_t j = wave; j < nsamp; j += 16) {
    // one position per stratum of count / nsamp list entries, at a pseudo-random place inside it (evenly strided samples alias
    // with anything periodic in the batch: round 6 found half of them on the benchmark generator's planted positions)
    const int64_t s_lo = (j * count) / nsamp, s_len = ((j + 1) * count) / nsamp - s_lo;
    const int64_t li = s_lo + (int64_t)(((uint64_t)(j + 1) * 0x9E3779B97F4A7C15ull >> 20) % (uint64_t)(s_len > 0 ? s_len : 1));
    const int64_t pos = list ? (int64_t)list[li] : li;
    bool ok = true;
    int lo = 0x7fffffff, hi = (int)0x80000000;
    long long ksum = 0; int ntot = 0;
    auto key_at = [&](int g, int64_t o, int i, int& k) -> bool {
      const void* sig = g ? a.sig1 : a.sig0;
      if constexpr (DTYPE == 0) return grid_key<true>(reinterpret_cast<const float*>(sig)[o + i], k);
      else if constexpr (DTYPE == 2) return cnt_int_key(reinterpret_cast<const float*>(sig)[o + i], k);
      else { k = (int)reinterpret_cast<const int16_t*>(sig)[o + i]; return true; }
    };
    int64_t oo[2]; int nn[2];
    for (int g = 0; g < 2; ++g) {
      const int64_t st = g ? a.stride1 : a.stride0;
      const int64_t* off = g ? a.off1 : a.off0;
      oo[g] = st > 0 ? pos * st : off[pos];
      nn[g] = st > 0 ? (int)st : (int)(off[pos + 1] - oo[g]);
      if (nn[g] < kCntMinN || nn[g] > kCntMaxN) ok = false;
      ntot += min(nn[g], kCntMaxN + 1);
      for (int i = lane; i < nn[g] && i <= kCntMaxN; i += 64) {
        int k;
        if (!key_at(g, oo[g], i, k)) ok = false;
        lo = min(lo, k); hi = max(hi, k); ksum += k;
      }
    }
    const int vmax = (int)(wave_max_u32((unsigned)hi ^ 0x80000000u) ^ 0x80000000u);
    const int vmin = (int)(~wave_max_u32(~((unsigned)lo ^ 0x80000000u)) ^ 0x80000000u);
    bool fit = __ballot(!ok) == 0ull && (unsigned)(vmax - vmin) < (unsigned)kCntWindow;
#if NMOD_CNT_TAILS
    if (!fit && __ballot(!ok) == 0ull) {
      // a few outliers: the kernel keeps the position when at most kCntTail samples lie further than 1 024 from the mean of its keys
      // (half of that here: the batch should mostly take the kernel's path without them).  The probe counts against a NARROWER
      // window, 768 either side: a mis-segmented read anywhere in +-5 units is outside both alike, but rows whose bulk fills the
      // kernel's window (sigma 0.4: ~20 samples beyond 768, 2-3 beyond 1 024 at EVERY position, the longest scans) cost the
      // counting form more than the sorting form takes — 6.37 v 6.11 ms on configs[2], measured in round 6 — and are left to it.
      const int centre = (int)__builtin_rintf((float)(long long)wave_sum_u64((unsigned long long)ksum) / (float)max(ntot, 1));
      const int base = max(-32768, min(centre - kCntProbeWindow / 2, 32768 - kCntProbeWindow));
      int tails = 0;
      for (int g = 0; g < 2; ++g)
        for (int i = lane; i < nn[g] && i <= kCntMaxN; i += 64) { int k; key_at(g, oo[g], i, k); tails += ((unsigned)(k - base) >= (unsigned)kCntProbeWindow) ? 1 : 0; }
      const int nfar = (int)wave_sum_u64((unsigned long long)tails);
      fit = nfar <= kCntTail / 2;
      if (fit && lane == 0) atomicAdd(&far, nfar);
    }
#endif
    if (lane == 0) { atomicAdd(&seen, 1); if (fit) atomicAdd(&fits, 1); }
  }
  __syncthreads();
  // (the kernel's outlier path costs ~15 % where a wave meets one such position and ~40 % at four outliers per position, measured
  // on configs[2] rows — the sorting form is 38 % slower than the clean counting form: worth it up to ~2.5 outliers per position)
  if (threadIdx.x == 0) a.gate[0] = (seen > 0 && fits * 8 >= seen * 7 && far * 2 <= seen * 5) ? 1 : 0;
}

template <int DTYPE>
__global__ __launch_bounds__(64 * kWavesPerBlock, NMOD_CNT_WAVES)
void rank_count_kernel(RankStatsArgs args) {
  constexpr int LG = kCntLanes, PW = 4, NS = 16;        // NS: sample slots per lane and group (4 chunks of 4)
  extern __shared__ __attribute__((aligned(16))) unsigned lds_cnt[];
  if (args.cnt_gate[0] == 0) return;                     // the probe found the batch not event-like: rank_hist_kernel takes all of it

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);
  const int slot = lane / LG;
  unsigned* tbl = lds_cnt + (wave * PW + slot) * kCntPosWords;
  const unsigned tb = (unsigned)(uintptr_t)(CntLdsU32)tbl;              // LDS byte address of the position's table

  const bool uniform = args.stride0 > 0 && args.stride1 > 0;
  double* recip = reinterpret_cast<double*>(lds_cnt + kWavesPerBlock * PW * kCntPosWords);
  if (uniform && threadIdx.x == 0) { recip[0] = 1.0 / (double)args.stride0; recip[1] = 1.0 / (double)args.stride1; }
  __syncthreads();

  int64_t count = args.npos;
  const int32_t* list = nullptr;
  if (args.pos_list) {
    count = args.class_meta[args.class_id];
    list = args.pos_list + args.class_meta[kClassStride + args.class_id];
  }
  const int64_t items = (count + PW - 1) / PW;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerBlock;

  struct Item { bool valid; int n0, n1; int64_t pos, o0, o1; };
  auto describe = [&](int64_t it) {
    Item d;
    const int64_t li = it * PW + slot;
    d.valid = it < items && li < count;
    d.pos = d.valid ? (list ? (int64_t)list[li] : li) : 0;
    d.o0 = 0; d.o1 = 0; d.n0 = 0; d.n1 = 0;
    if (d.valid) {
      if (args.stride0 > 0) { d.o0 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride0); d.n0 = (int)args.stride0; }
      else { d.o0 = args.off0[d.pos]; d.n0 = (int)(args.off0[d.pos + 1] - d.o0); }
      if (args.stride1 > 0) { d.o1 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride1); d.n1 = (int)args.stride1; }
      else { d.o1 = args.off1[d.pos]; d.n1 = (int)(args.off1[d.pos + 1] - d.o1); }
    }
    return d;
  };
  // rows are read through KsRows (ks_rank.hpp): unconditional 16-byte (8-byte) loads, lane gl takes samples
  // c * 64 + 4 gl .. + 3 of chunk c; a row this form cannot take (fewer than 4 or more than 255 samples) is not read
  auto rows_n = [](const Item& d, int n) { return (d.valid && d.n0 >= kCntMinN && d.n0 <= kCntMaxN && d.n1 >= kCntMinN && d.n1 <= kCntMaxN) ? n : 0; };
  constexpr int RDT = (DTYPE == 1) ? 1 : 0;               // the rows' storage type: float32 (DTYPE 0, 2) or int16
  using Rows = KsRows<NS, LG, RDT>;

  Item cur = describe(wave_global);
  Rows rw0, rw1;
  rw0.request(args.sig0, cur.o0, rows_n(cur, cur.n0), gl);
  rw1.request(args.sig1, cur.o1, rows_n(cur, cur.n1), gl);

  for (int64_t it = wave_global; it < items; it += wave_stride) {
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this item's rows
    const bool valid = cur.valid;
    const int64_t pos = cur.pos;
    const int n0 = cur.n0, n1 = cur.n1;
    bool fit = rows_n(cur, 1) != 0;

    // ---- keys, two per register (int16 halves).  Chunk c of group g = registers KP[8 g + 2 c], KP[8 g + 2 c + 1]: components
    // (0, 1) and (2, 3), i.e. samples 64 c + 4 gl + 0 .. 3.  A chunk that holds the end of a row was read as the row's LAST four
    // samples (KsRows): its components j >= 4 - t are this lane's, t = samples left at the chunk's start.  Per chunk, wave-wide:
    // `full` — every lane of the wave holds four samples (no selects anywhere); `any` — some lane holds a sample (a chunk without
    // any is skipped in every phase).  A slot without a sample carries the group's first key.
    unsigned KP[NS];                                     // [8 g + 2 c + h]
    unsigned vmask = 0u;                                 // bit 16 g + 4 c + j: the slot holds a sample (chunks that are not full)
    unsigned fullm = 0u, anym = 0u;                      // wave-uniform: bit 4 g + c
    bool bad = false;                                    // float32: a sample of this lane is off the grid
    double f1[2] = {0.0, 0.0}, f2[2] = {0.0, 0.0};       // float32: shifted moment sums of the values
    int i1[2] = {0, 0}, i2[2] = {0, 0};                  // int16: exact moment sums of k - first key
    int kfirst[2] = {0, 0};
    float xfirst2[2] = {0.0f, 0.0f};
    CntS2 MN = {32767, 32767}, MX = {-32768, -32768};
    const CntS2 ones = {1, 1};
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const Rows& rw = g ? rw1 : rw0;
      const int n = rows_n(cur, g ? n1 : n0);
      // the group's first sample (lane 0, chunk 0, component 0: there for every row of at least 4 samples) stands in the empty slots
      float xfirst = 0.0f; double K = 0.0; int kf = 0;
      if constexpr (DTYPE != 1) {
        xfirst = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, __float_as_int((float)rw.v[0].x)));
        K = (double)xfirst;
        xfirst2[g] = xfirst;
      } else {
        kf = __builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, (int)rw.v[0].x);
      }
      kfirst[g] = kf;
      const unsigned kf2 = ((unsigned)kf & 0xffffu) | ((unsigned)kf << 16);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int t = n - (c * 64 + 4 * gl);
        const bool full = __ballot(t < 4) == 0ull;
        const bool any = __ballot(t > 0) != 0ull;
        fullm |= full ? (1u << (4 * g + c)) : 0u;
        anym |= any ? (1u << (4 * g + c)) : 0u;
        if (!any) continue;
        const int first = (t >= 4) ? 0 : ((t <= 0) ? 4 : 4 - t);          // components first .. 3 are samples of this lane
        if (!full) vmask |= (0xfu & (0xfu << first)) << (16 * g + 4 * c);
        if constexpr (DTYPE != 1) {
          auto four = [&](auto full_tag) {
            int kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float xr = (j == 0) ? rw.v[c].x : (j == 1) ? rw.v[c].y : (j == 2) ? rw.v[c].z : rw.v[c].w;
              const float x = (decltype(full_tag)::value || j >= first) ? xr : xfirst;
              const bool ok = (DTYPE == 2) ? cnt_int_key(x, kk[j]) : grid_key<true>(x, kk[j]);   // (|k| <= 32 767: the key fits its half)
              bad = bad || !ok;
              if constexpr (DTYPE == 0) {
                const double d = (double)x - K;
                f1[g] += d;
                f2[g] = __fma_rn(d, d, f2[g]);
              }
            }
            KP[8 * g + 2 * c] = ((unsigned)kk[0] & 0xffffu) | ((unsigned)kk[1] << 16);
            KP[8 * g + 2 * c + 1] = ((unsigned)kk[2] & 0xffffu) | ((unsigned)kk[3] << 16);
          };
          if (full) four(std::true_type{}); else four(std::false_type{});
        } else {
          unsigned p0 = ((unsigned)(unsigned short)rw.v[c].x) | ((unsigned)(unsigned short)rw.v[c].y << 16);
          unsigned p1 = ((unsigned)(unsigned short)rw.v[c].z) | ((unsigned)(unsigned short)rw.v[c].w << 16);
          if (!full) {                                                      // the components below `first`: the group's first key
            const unsigned m0 = (first <= 0 ? 0xffffu : 0u) | (first <= 1 ? 0xffff0000u : 0u);
            const unsigned m1 = (first <= 2 ? 0xffffu : 0u) | (first <= 3 ? 0xffff0000u : 0u);
            p0 = (p0 & m0) | (kf2 & ~m0);
            p1 = (p1 & m1) | (kf2 & ~m1);
          }
          KP[8 * g + 2 * c] = p0; KP[8 * g + 2 * c + 1] = p1;
#pragma unroll
          for (int h = 0; h < 2; ++h) {                                     // exact moment sums of k - first key (|d| <= 2 047 where the position fits)
            const CntS2 d = __builtin_bit_cast(CntS2, h ? p1 : p0) - __builtin_bit_cast(CntS2, kf2);
            i1[g] = __builtin_amdgcn_sdot2(d, ones, i1[g], false);
            i2[g] = __builtin_amdgcn_sdot2(d, d, i2[g], false);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const CntS2 kp = __builtin_bit_cast(CntS2, KP[8 * g + 2 * c + h]);
          MN = __builtin_elementwise_min(MN, kp); MX = __builtin_elementwise_max(MX, kp);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- the window: the position's smallest key is value 0 (moved up: the outlier path below needs the range)
    int kmin = cnt_allmin_i32(min((int)MN.x, (int)MN.y)), kmax = cnt_allmax_i32(max((int)MX.x, (int)MX.y));
    // ---- outliers (round 6).  A mis-segmented read sits anywhere in the +-5 unit clip range (myRefBaseSignalAnnotation.py:251-259)
    // and pushes the position's key range past the window.  Where a position of the wave is that wide, its samples further than
    // 1 024 milli-units from the mean of its keys — TAIL samples, at most kCntTail = 16 — leave their slots (which become empty
    // slots) for a list in LDS; the rank statistics depend on the ORDER of the keys only, so the tail samples come back as one
    // extra sample slot per lane with REMAPPED keys: the distinct tail values below the window take the values just below the
    // smallest remaining key, those above it the values just above the largest, in order, ties kept (dense ranks from an all-pairs
    // pass over the list).  Tables, scans, look-ups and sums then run as for any position; only the Welch moments use the true keys.
    int xk = 0; unsigned xg = 0u; bool xv = false;       // the lane's extra slot: remapped key, group, in use
    int xd[2] = {0, 0}; double xdd[2] = {0.0, 0.0};      // ... its true moment terms about the reference key (int16 rows)
    bool have_x = false;                                  // (wave-uniform) some lane of the wave holds an extra slot
    {
      const bool wide = fit && (unsigned)(kmax - kmin) >= (unsigned)kCntWindow;
      if (__ballot(wide) != 0ull) {                       // (wave-uniform; never on rows without outliers)
        // the mean of the position's keys: empty slots carry their group's first key, taken out again by their count
        int ksum = 0, kf[2], nv[2];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          kf[g] = (int)(short)(__builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, (int)KP[8 * g]) & 0xffff);
          const int n = rows_n(cur, g ? n1 : n0);
          nv[g] = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) nv[g] += min(max(n - (c * 64 + 4 * gl), 0), 4);
#pragma unroll
          for (int i = 0; i < 8; ++i) if ((anym >> (4 * g + (i >> 1))) & 1u) ksum = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, KP[8 * g + i]), ones, ksum, false);
          int held = 0;                                   // slots of the chunks that exist: 4 per chunk
#pragma unroll
          for (int c = 0; c < 4; ++c) held += ((anym >> (4 * g + c)) & 1u) ? 4 : 0;
          ksum -= kf[g] * (held - nv[g]);
        }
        const int ntot = max(rows_n(cur, n0) + rows_n(cur, n1), 1);
        const int centre = (int)__builtin_rintf((float)cnt_allsum_i32(ksum) * __builtin_amdgcn_rcpf((float)ntot));
        const int base = max(-32768, min(centre - kCntWindow / 2, 32768 - kCntWindow));
        const int cref = base + kCntWindow / 2;           // the reference key of the moments; |k - cref| <= 1 024 for what stays
        const unsigned base2 = ((unsigned)base & 0xffffu) * 0x10001u;
        const CntU2 win2 = {(unsigned short)kCntWindow, (unsigned short)kCntWindow};
        unsigned* tlist = tbl + kCntTableWords;           // the position's tail list: key | group << 16
        int listed = 0;                                   // (the same in the 16 lanes of a position)
        unsigned tmask = 0u, cht = 0u;                    // bit s: slot s of this lane is a tail sample; (wave-uniform) bit ch: chunk ch holds one
        auto push = [&](bool t, int key, unsigned grp) {
          const unsigned long long mk = __ballot(t);
          if (mk == 0ull) return false;
          const unsigned seg = (unsigned)(mk >> (lane & 48)) & 0xffffu;
          const int idx = listed + (int)__popc(seg & ((1u << gl) - 1u));
          if (t && idx < kCntTail) tlist[idx] = ((unsigned)key & 0xffffu) | (grp << 16);
          listed += (int)__popc(seg);
          return true;
        };
        // one cheap pass over the 16 registers: u = min(k - base, 2 048) per slot (2 048: outside the window), the smallest and the
        // largest u below 2 048, (int16 rows) the sums of u - 1 024 and its square over ALL slots — the outside and the empty slots'
        // constant shares are taken out below by their counts; only a register in which some lane of the wave holds an outside slot
        // pays for the validity tests and the appends
        CntU2 UMN = {65535, 65535}, UMX = {65535, 65535};   // min u; min of 2 047 - u as unsigned (an outside slot gives 65 535: ignored)
        int j1[2] = {0, 0}, j2[2] = {0, 0}, ntl[2] = {0, 0};
        const CntU2 top2 = {(unsigned short)(kCntWindow - 1), (unsigned short)(kCntWindow - 1)};
        const CntS2 half2 = {(short)(kCntWindow / 2), (short)(kCntWindow / 2)};
#pragma unroll
        for (int i = 0; i < NS; ++i) {
          const int ch = i >> 1;                          // chunk 4 g + c; the register holds its slots 2 (i & 1), 2 (i & 1) + 1
          if (!((anym >> ch) & 1u)) continue;
          const int grp = ch >> 2;
          const CntU2 u2 = __builtin_elementwise_min(__builtin_bit_cast(CntU2, __builtin_bit_cast(CntS2, KP[i]) - __builtin_bit_cast(CntS2, base2)), win2);
          UMN = __builtin_elementwise_min(UMN, u2); UMX = __builtin_elementwise_min(UMX, top2 - u2);
          if constexpr (DTYPE == 1) {
            const CntS2 d = __builtin_bit_cast(CntS2, u2) - half2;
            j1[grp] = __builtin_amdgcn_sdot2(d, ones, j1[grp], false);
            j2[grp] = __builtin_amdgcn_sdot2(d, d, j2[grp], false);
          }
          const unsigned ub = __builtin_bit_cast(unsigned, u2);
          if (__ballot(wide && (ub & 0x08000800u) != 0u) != 0ull) {   // (wave-uniform) some lane: a slot outside the window
            const bool fullc = ((fullm >> ch) & 1u) != 0u;
            const bool v0 = fullc || ((vmask >> (2 * i)) & 1u) != 0u, v1 = fullc || ((vmask >> (2 * i + 1)) & 1u) != 0u;
            const bool t0 = wide && v0 && (ub & 0x0800u) != 0u, t1 = wide && v1 && ub >= 0x08000000u;
            const bool a0 = push(t0, (int)(short)(KP[i] & 0xffffu), (unsigned)grp), a1 = push(t1, (int)(short)(KP[i] >> 16), (unsigned)grp);
            if (a0 || a1) cht |= 1u << ch;
            tmask |= (t0 ? 1u << (2 * i) : 0u) | (t1 ? 1u << (2 * i + 1) : 0u);
            ntl[grp] += (t0 ? 1 : 0) + (t1 ? 1 : 0);
          }
        }
        if constexpr (DTYPE == 1) {
          // a tail slot entered the sums as 1 024 (u = 2 048), an empty slot as its group's first key does
#pragma unroll
          for (int g = 0; g < 2; ++g) {
            int held = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) held += ((anym >> (4 * g + c)) & 1u) ? 4 : 0;
            const int de = (int)min((unsigned)(kf[g] - base) , (unsigned)kCntWindow) - kCntWindow / 2, ne = held - nv[g];
            j1[g] -= (kCntWindow / 2) * ntl[g] + de * ne;
            j2[g] -= (kCntWindow / 2) * (kCntWindow / 2) * ntl[g] + de * de * ne;
          }
        }
        // the chunks that lost a sample are not "full" any more: their slots get validity bits, the tail slots lose theirs
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) if (((cht >> ch) & 1u) && ((fullm >> ch) & 1u)) vmask |= 0xfu << (4 * ch);
        fullm &= ~cht;
        vmask &= ~tmask;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (wide) {
          if constexpr (DTYPE == 1) { i1[0] = j1[0]; i1[1] = j1[1]; i2[0] = j2[0]; i2[1] = j2[1]; kfirst[0] = cref; kfirst[1] = cref; }
          fit = listed <= kCntTail;
        }
        // lane gl < listed holds tail sample gl; every one against every other: earlier copies, distinct values between it and the window
        const bool tv = wide && fit && gl < listed;
        const unsigned te = tv ? tlist[gl] : 0u;
        const int tk = (int)(short)(te & 0xffffu);
        const bool tlow = tk < base;
        int tp = 0, dr = 0;
        const int maxn = max(max(__builtin_amdgcn_readlane(wide && fit ? listed : 0, 0), __builtin_amdgcn_readlane(wide && fit ? listed : 0, 16)),
                             max(__builtin_amdgcn_readlane(wide && fit ? listed : 0, 32), __builtin_amdgcn_readlane(wide && fit ? listed : 0, 48)));
#pragma unroll 1
        for (int j = 0; j < maxn; ++j) {
          const int mine = (tk & 0xffff) | (tp == 0 ? 0x10000 : 0);                   // (lane j's count of earlier copies is complete by round j)
          const int wj = __builtin_amdgcn_ds_bpermute(((lane & ~(LG - 1)) + j) << 2, mine);
          const int kj = (int)(short)(wj & 0xffff);
          const bool vj = tv && j < listed, firstj = vj && (wj & 0x10000) != 0, lowj = kj < base;
          tp += (vj && kj == tk && j < gl) ? 1 : 0;
          dr += (firstj && lowj == tlow && (tlow ? kj > tk : kj < tk)) ? 1 : 0;
        }
        // the distinct tail values on either side of the window (first copies), per position
        const unsigned fl = (unsigned)(__ballot(tv && tp == 0 && tlow) >> (lane & 48)) & 0xffffu, fh = (unsigned)(__ballot(tv && tp == 0 && !tlow) >> (lane & 48)) & 0xffffu;
        const int ndl = (int)__popc(fl), ndh = (int)__popc(fh);
        if (wide && fit) {
          // what stayed spans [in_min, in_max] (inside the window); the distinct tail values sit just outside it, in order
          const int umin = cnt_allmin_i32((int)min(UMN.x, UMN.y));
          const bool none = umin >= kCntWindow;                                         // (no sample inside the window at all: the tail keys sit around cref)
          const int in_min = none ? cref : base + umin, in_max = none ? cref - 1 : base + (kCntWindow - 1) - cnt_allmin_i32((int)min(UMX.x, UMX.y));
          kmin = in_min - ndl; kmax = in_max + ndh;
          fit = kmin >= -32768 && kmax <= 32767;                                       // (the remapped keys are int16 keys; the range is tested below)
          xk = tlow ? in_min - 1 - dr : in_max + 1 + dr;
          xg = (te >> 16) & 1u;
          xv = tv && fit;
          if constexpr (DTYPE == 1) {                      // the sample's true terms about cref (|d| < 2^16: d^2 as a double)
            const int d = tk - cref;
            xd[0] = (xv && xg == 0u) ? d : 0; xd[1] = (xv && xg != 0u) ? d : 0;
            xdd[0] = (double)xd[0] * (double)xd[0]; xdd[1] = (double)xd[1] * (double)xd[1];
          }
        }
        have_x = __ballot(xv) != 0ull;
      }
    }
    // ---- the moments are final here: written at once (a position that turns out not to fit is written again by rank_hist_kernel)
    if constexpr (DTYPE != 2) {
      double mean[2], m2[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const double dn = (double)(g ? n1 : n0);
        const double rn = uniform ? recip[g] : 1.0 / dn;
        if constexpr (DTYPE == 0) {
          const double s1 = seg_allsum_f64<LG>(f1[g]), s2 = seg_allsum_f64<LG>(f2[g]);
          const double K = (double)xfirst2[g];
          mean[g] = K + s1 * rn; m2[g] = s2 - s1 * s1 * rn;
        } else {
          double S1 = (double)cnt_allsum_i32(i1[g] + xd[g]), S2 = (double)cnt_allsum_i32(i2[g]);   // exact: |S1| < 2^21, S2 < 2^31 (fitting positions)
          if (have_x) S2 += seg_allsum_f64<LG>(xdd[g]);                                          // (wave-uniform) + the tail samples' squares
          mean[g] = ((double)kfirst[g] + S1 * rn) * 1e-3;
          m2[g] = __fma_rn(dn, S2, -S1 * S1) * rn * 1e-6;                                       // (n S2 - S1^2: exact integers)
        }
      }
      if (fit && gl == 0) {
        double* mo = args.moments + pos * 4;
        mo[0] = mean[0]; mo[1] = m2[0]; mo[2] = mean[1]; mo[3] = m2[1];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- the lane blocks are as long as the widest position of the wave needs
    {
      const unsigned long long bm = __ballot(bad);
      const bool pos_bad = ((unsigned)(bm >> (lane & 48)) & 0xffffu) != 0u;
      fit = fit && !pos_bad && (unsigned)(kmax - kmin) < (unsigned)kCntWindow;
    }
    // 16-byte chunks per lane block: 2, 4, 6 or 8 (block = 32 .. 128 values; an even count keeps the padded block stride an odd
    // number of 16-byte units: the lanes' 16-byte accesses of the scan stay on different banks)
    int nc;
    {
      const int need = fit ? 2 * (((kmax - kmin) >> 9) + 1) : 2;
      nc = max(max(__builtin_amdgcn_readlane(need, 0), __builtin_amdgcn_readlane(need, 16)),
               max(__builtin_amdgcn_readlane(need, 32), __builtin_amdgcn_readlane(need, 48)));
    }
    const unsigned bdiv = (nc == 2) ? 131072u : (nc == 4) ? 65536u : (nc == 6) ? 43691u : 32768u;   // ceil(2^22 / (16 nc)): u * bdiv >> 22 = u / (16 nc), u < 2048
    const int bstride = 16 * nc + 16;                    // bytes from a lane's block to the next (its 16-byte pad first)

    // Only group 1's samples are looked up (in both tables); group 2's are only counted.  Every pooled point that can carry the
    // KS maximum is a point (A[v], B[v]) or (A[v-1], B[v-1]) at a value v of group 1 (between two values of group 1 F_A is
    // constant and F_B rises: the extremes sit at the ends); mwu_s = sum over a in group 1 of (B[a-1] + B[a]); and with a(v), b(v)
    // the two groups' counts of value v, sum_v (a + b)^3 = sum_{e in group 1} (a^2 + 3 a b + 3 b^2)(e) + sum_v b^3, the last term
    // from the arrival numbers the counting adds of group 2 return: sum_{e in group 2} (3 p^2 + 3 p + 1), p = earlier copies.
    unsigned mws = 0u, saa = 0u, sab = 0u, sbb = 0u, b3 = 0u;
    unsigned AX = 0u, xA = 0u, xA1 = 0u, xnum0 = 0u, xnum1 = 0u;   // the extra slot: its table address, A[v], A[v-1], its two KS candidates
    CntU2 BEST = {0, 0};
    double dmax = 0.0;
    constexpr int NP = NS / 2;                           // pairs of group 1's slots
    unsigned CAp[NP], CA1p[NP];                          // per pair: A[u], A[u-1]
    unsigned N0p[NP], N1p[NP];                           // per pair: |A[u] n1 - B[u] n0|, |A[u-1] n1 - B[u-1] n0|
    Item nxt;
    Rows nx0, nx1;
    auto request_next = [&](int part) {                 // part 0: describe + group 1's rows; 1: group 2's rows; 2: both
      if (part != 1) { nxt = describe(it + wave_stride); nx0.request(args.sig0, nxt.o0, rows_n(nxt, nxt.n0), gl); }
      if (part != 0) nx1.request(args.sig1, nxt.o1, rows_n(nxt, nxt.n1), gl);
    };
    if (fit) {
      // byte address of cum[u - 1]: tb + 15 + u + 16 (u / block)  (block b starts 16 (b + 1) bytes later than in a flat table),
      // two per register; a slot without a sample: the first two bytes of the first pad, always zero
      const unsigned kmin2 = ((unsigned)kmin & 0xffffu) | ((unsigned)kmin << 16);
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        if (!((anym >> ch) & 1u)) continue;
        auto addr = [&](auto full_tag) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const CntU2 u2 = __builtin_bit_cast(CntU2, __builtin_bit_cast(CntS2, KP[2 * ch + h]) - __builtin_bit_cast(CntS2, kmin2));
            unsigned a[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const unsigned u = e ? (unsigned)u2.y : (unsigned)u2.x;
              a[e] = tb + 15u + u + ((__umul24(u, bdiv) >> 22) << 4);
              if constexpr (!decltype(full_tag)::value) a[e] = ((vmask >> (4 * ch + 2 * h + e)) & 1u) ? a[e] : tb;
            }
            KP[2 * ch + h] = a[0] | (a[1] << 16);
          }
        };
        if ((fullm >> ch) & 1u) addr(std::true_type{}); else addr(std::false_type{});
      }
      if (have_x) {                                        // (wave-uniform) the extra slot: a tail sample under its remapped key
        const unsigned u = (unsigned)(xk - kmin) & 0xffffu;
        AX = xv ? tb + 15u + u + ((__umul24(u, bdiv) >> 22) << 4) : tb;
      }
      // (made opaque after every phase: otherwise the compiler keeps unpacked copies alive beside the packed registers)
#pragma unroll
      for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(KP[s]));
      __builtin_amdgcn_sched_barrier(0);
      uint4* blk = reinterpret_cast<uint4*>(__builtin_assume_aligned(reinterpret_cast<char*>(tbl) + gl * bstride + 16, 16));   // the lane's block (behind its pad)
      auto clear_table = [&]() {
        unsigned z = 0u;
        asm volatile("" : "+v"(z));                       // (a fresh zero per call: hoisted out of the item loop, the zero quad is spilled and reloaded per store)
#pragma unroll
        for (int i = 0; i < 9; ++i) if (i <= nc) blk[i - 1] = make_uint4(z, z, z, z);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto build = [&](int g) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int ch = 4 * g + c;
          if (!((anym >> ch) & 1u)) continue;
          auto add4 = [&](auto full_tag) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const unsigned ap = KP[2 * ch + (j >> 1)];
              const unsigned a1 = ((j & 1) ? (ap >> 16) : (ap & 0xffffu)) + 1u;   // byte of the sample's counter (empty slot: tb + 1, adds 0)
              const unsigned one = 1u << ((a1 & 3u) * 8u);
              const unsigned val = (decltype(full_tag)::value || ((vmask >> (4 * ch + j)) & 1u)) ? one : 0u;
              const unsigned old = __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)(a1 & ~3u), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
              if (g == 1) {                                                        // earlier copies of the value: p (an empty slot reads the zero pad)
                const unsigned pcnt = (old >> ((a1 & 3u) * 8u)) & 0xffu;
                b3 += __umul24(pcnt, pcnt) + pcnt;
              }
            }
          };
          if ((fullm >> ch) & 1u) add4(std::true_type{}); else add4(std::false_type{});
        }
        if (have_x) {                                      // (wave-uniform) the tail samples of this group, one per lane
          const unsigned a1 = AX + 1u;
          const unsigned old = __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)(a1 & ~3u), (xv && xg == (unsigned)g) ? 1u << ((a1 & 3u) * 8u) : 0u,
                                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          if (g == 1) {
            const unsigned pcnt = (xv && xg != 0u) ? (old >> ((a1 & 3u) * 8u)) & 0xffu : 0u;
            b3 += __umul24(pcnt, pcnt) + pcnt;
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      // counts -> inclusive prefix sums, in place.  Lane gl owns block gl: nc chunks of 16 counters behind its pad.
      auto scan = [&]() {
        // the lane's total first (its words are read again below: registers for all of them would not fit beside the samples)
        unsigned tot = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (i < nc) {
            const uint4 q = blk[i];
            tot = __builtin_amdgcn_sad_u8(q.x, 0u, tot); tot = __builtin_amdgcn_sad_u8(q.y, 0u, tot);
            tot = __builtin_amdgcn_sad_u8(q.z, 0u, tot); tot = __builtin_amdgcn_sad_u8(q.w, 0u, tot);
          }
        }
        const unsigned base = seg_exscan_add_u32<LG>(tot, gl);                  // samples below the lane's first value (< 256)
        unsigned carry = base | (base << 8);
        carry |= carry << 16;
        blk[-1] = make_uint4(0u, 0u, 0u, base << 24);                            // byte 15 of the pad: cum just below the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (2 * h < nc) {
            unsigned w[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) { const uint4 q = blk[2 * h + i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              // (inline asm: written as v + (v << 8) the compiler multiplies by 0x01010101 with a 64-bit v_mad_u64_u32 per word)
              unsigned v = w[i], t;
              asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(t) : "v"(v));
              asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(v) : "v"(t));
              v += carry;
              carry = __builtin_amdgcn_perm(v, v, 0x03030303u);                  // the word's last byte in all four
              w[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) blk[2 * h + i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      // cum[u - 1] and cum[u] of the two slots of every pair of a chunk: four byte reads, joined to two registers of 16-bit halves
      // (an unaligned 16-bit read costs the LDS pipe a pass per lane; D16 loads into register halves do not keep the other
      // half on this part: SRAM-ECC)
      // ---- table A: group 1
      clear_table();
      build(0);
      scan();
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (!((anym >> ch) & 1u)) continue;
        unsigned c1[2], c0[2];
        cnt_lookup_chunk(KP[2 * ch], KP[2 * ch + 1], c1, c0);
#pragma unroll
        for (int h = 0; h < 2; ++h) { CAp[2 * ch + h] = c0[h]; CA1p[2 * ch + h] = c1[h]; }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (have_x) { const CntLdsU8 px = (CntLdsU8)(uintptr_t)AX; xA1 = px[0]; xA = px[1]; }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);

      // ---- table B: group 2
      clear_table();
      build(1);
      scan();
    }
    // the next item's rows, used at the top of the next iteration: int16 rows (16 registers) are requested here, before the last
    // lookups; float32 rows (32 registers) behind them, the second group's behind the float-form pass
    if constexpr (DTYPE != 0) request_next(2);
    if (fit) {
      const CntU2 n0x2 = {(unsigned short)n0, (unsigned short)n0}, n1x2 = {(unsigned short)n1, (unsigned short)n1};
      const CntU2 one2 = {1, 1};
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        if (!((anym >> ch) & 1u)) continue;
        unsigned c1[2], c0[2];
        cnt_lookup_chunk(KP[2 * ch], KP[2 * ch + 1], c1, c0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int p = 2 * ch + h;
          const CntU2 cB = __builtin_bit_cast(CntU2, c0[h]), cB1 = __builtin_bit_cast(CntU2, c1[h]);
          const CntU2 cA = __builtin_bit_cast(CntU2, CAp[p]), cA1 = __builtin_bit_cast(CntU2, CA1p[p]);
          const CntU2 ta = cA - cA1, tb = cB - cB1;                               // copies of the sample's value in either group (0, 0: empty slot)
          saa = __builtin_amdgcn_udot2(ta, ta, saa, false);
          sab = __builtin_amdgcn_udot2(ta, tb, sab, false);
          sbb = __builtin_amdgcn_udot2(tb, tb, sbb, false);
          mws = __builtin_amdgcn_udot2(cB + cB1, one2, mws, false);               // #{b < a} + #{b <= a}
          const CntU2 x0 = cA * n1x2, y0 = cB * n0x2, x1 = cA1 * n1x2, y1 = cB1 * n0x2;   // <= 255 * 255
          const CntU2 num0 = __builtin_elementwise_max(x0, y0) - __builtin_elementwise_min(x0, y0);   // |A n1 - B n0| at v
          const CntU2 num1 = __builtin_elementwise_max(x1, y1) - __builtin_elementwise_min(x1, y1);   // ... just below v
          BEST = __builtin_elementwise_max(BEST, __builtin_elementwise_max(num0, num1));
          N0p[p] = __builtin_bit_cast(unsigned, num0); N1p[p] = __builtin_bit_cast(unsigned, num1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (have_x) {                                        // (wave-uniform) group 1's tail samples: the same terms, unpacked
        const CntLdsU8 px = (CntLdsU8)(uintptr_t)AX;
        const bool g1 = xv && xg == 0u;
        const unsigned cB1 = g1 ? px[0] : 0u, cB = g1 ? px[1] : 0u, cA1 = g1 ? xA1 : 0u, cA = g1 ? xA : 0u;
        const unsigned ta = cA - cA1, tbq = cB - cB1;
        saa += ta * ta; sab += ta * tbq; sbb += tbq * tbq; mws += cB + cB1;
        const int x0 = (int)(cA * (unsigned)n1) - (int)(cB * (unsigned)n0), x1 = (int)(cA1 * (unsigned)n1) - (int)(cB1 * (unsigned)n0);
        xnum0 = (unsigned)(x0 < 0 ? -x0 : x0); xnum1 = (unsigned)(x1 < 0 ? -x1 : x1);
        const unsigned short xb = (unsigned short)max(xnum0, xnum1);
        BEST = __builtin_elementwise_max(BEST, CntU2{xb, xb});
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (DTYPE == 0) request_next(0);
    if (__ballot(fit) != 0ull) {
      unsigned best = fit ? seg_allmax_u32<LG>(max((unsigned)BEST.x, (unsigned)BEST.y)) : 0u;
      // ---- the float form of D at the samples that reach the integer maximum
      const double dn0 = (double)n0, dn1 = (double)n1;
      double r0, r1;
      if (uniform) { r0 = recip[0]; r1 = recip[1]; } else { r0 = 1.0 / dn0; r1 = 1.0 / dn1; }
#if !(NMOD_CNT_SKIP & 1)
#pragma unroll
      for (int s = 0; s < 2 * NS; ++s) {                                          // candidate s: slot s >> 1 of group 1, at its value (even) / just below it (odd)
        const int sl = s >> 1, p = sl >> 1;
        if (!((anym >> (sl >> 2)) & 1u)) continue;
        const unsigned nn = (s & 1) ? N1p[p] : N0p[p];
        const unsigned num = (sl & 1) ? (nn >> 16) : (nn & 0xffffu);
        const bool hit = fit && num == best && best != 0u;
        if (__ballot(hit) != 0ull) {
          // (B's count is read again — table B stands until the next item clears it; keeping all of them would cost 16 registers)
          const unsigned ca = (s & 1) ? CA1p[p] : CAp[p];
          const int cA = (int)((sl & 1) ? (ca >> 16) : (ca & 0xffffu));
          const unsigned ab = (sl & 1) ? (KP[p] >> 16) : (KP[p] & 0xffffu);
          const int cB = (int)((CntLdsU8)(uintptr_t)ab)[(s & 1) ? 0 : 1];
          const double d = fabs(hist_exact_quot(cA, dn0, r0) - hist_exact_quot(cB, dn1, r1));
          dmax = hit ? fmax(dmax, d) : dmax;
        }
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
      if (have_x) {                                        // (wave-uniform) ... and at group 1's tail samples
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const bool hit = fit && xv && xg == 0u && (e ? xnum1 : xnum0) == best && best != 0u;
          if (__ballot(hit) != 0ull) {
            const int cA = (int)(e ? xA1 : xA);
            const int cB = (int)((CntLdsU8)(uintptr_t)AX)[e ? 0 : 1];
            const double d = fabs(hist_exact_quot(cA, dn0, r0) - hist_exact_quot(cB, dn1, r1));
            dmax = hit ? fmax(dmax, d) : dmax;
          }
        }
      }
#endif
      dmax = seg_allmax_f64<LG>(dmax);
      const unsigned MWS = pos_allsum_u32<LG>(mws);
      // sum_v (a + b)^3 (< 2^28): group 1's samples give a^3 + 3 a^2 b + 3 a b^2, group 2's arrival numbers b^3
      const unsigned CUBES = pos_allsum_u32<LG>(saa + 3u * (sab + sbb) + 3u * b3) + (unsigned)n1;
      if (fit && gl == 0) {
        args.ks_num[pos] = best;
        args.ks_d_ref[pos] = dmax;
        args.mwu_s[pos] = (unsigned long long)MWS;
        args.tie[pos] = (unsigned long long)(CUBES - (unsigned)(n0 + n1));
        if (args.tied) args.tied[pos] = (CUBES != (unsigned)(n0 + n1)) ? 1 : 0;
      }
    }
    if constexpr (DTYPE == 0) request_next(1);
    {
      // the item's four flag bytes as one dword (1: nothing left to do for the position — produced here, or no position at all)
      const unsigned long long fm = __ballot(fit || !valid);
      const unsigned word = (unsigned)(fm & 1ull) | ((unsigned)((fm >> 16) & 1ull) << 8) | ((unsigned)((fm >> 32) & 1ull) << 16) |
                            ((unsigned)((fm >> 48) & 1ull) << 24);
      if (lane == 0) reinterpret_cast<unsigned*>(args.cnt_done)[it] = word;
    }
    rw0 = nx0; rw1 = nx1;
    cur = nxt;
  }
}

}  // namespace nmod
