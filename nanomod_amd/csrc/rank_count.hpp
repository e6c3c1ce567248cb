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
constexpr int kCntPosWords = kCntLanes * kCntBlockWords;        // 576
constexpr int kCntMaxN = 255;                                   // byte counters and byte prefix sums
constexpr int kCntMinN = 4;                                     // (KsRows reads shorter rows through a conditional path)
__host__ __device__ constexpr size_t rank_count_lds_bytes() { return (size_t)kWavesPerBlock * 4 * kCntPosWords * 4 + 16; }

typedef __attribute__((address_space(3))) const unsigned char* CntLdsU8;
typedef __attribute__((address_space(3))) unsigned* CntLdsU32;

#ifndef NMOD_CNT_SKIP
#define NMOD_CNT_SKIP 0
#endif
#ifndef NMOD_CNT_WAVES
#define NMOD_CNT_WAVES 4
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

// ---- is this batch event-like?  One block; wave w looks at sampled positions w, w + 16, ...: sizes, grid, range — the
// very test rank_count_kernel applies per position.  gate[0] = 1 when at least 7 of 8 sampled positions qualify.
struct CntProbeArgs {
  const void* sig0; const void* sig1; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1; int64_t npos;
  const int32_t* pos_list; const int32_t* class_meta; int32_t class_id; int32_t dtype; int32_t* gate;
};
constexpr int kCntProbeSamples = 64;

template <int DTYPE>
__global__ __launch_bounds__(1024) void cnt_probe_kernel(CntProbeArgs a) {
  __shared__ int fits, seen;
  if (threadIdx.x == 0) { fits = 0; seen = 0; }
  __syncthreads();
  int64_t count = a.npos;
  const int32_t* list = nullptr;
  if (a.pos_list) { count = a.class_meta[a.class_id]; list = a.pos_list + a.class_meta[kClassStride + a.class_id]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nsamp = count < kCntProbeSamples ? count : kCntProbeSamples;
  for (int64_t j = wave; j < nsamp; j += 16) {
    const int64_t li = (j * count) / nsamp;
    const int64_t pos = list ? (int64_t)list[li] : li;
    bool ok = true;
    int lo = 0x7fffffff, hi = (int)0x80000000;
    for (int g = 0; g < 2; ++g) {
      const int64_t st = g ? a.stride1 : a.stride0;
      const int64_t* off = g ? a.off1 : a.off0;
      const int64_t o = st > 0 ? pos * st : off[pos];
      const int n = st > 0 ? (int)st : (int)(off[pos + 1] - o);
      if (n < kCntMinN || n > kCntMaxN) ok = false;
      const void* sig = g ? a.sig1 : a.sig0;
      for (int i = lane; i < n && i <= kCntMaxN; i += 64) {
        int k;
        if constexpr (DTYPE == 0) { if (!grid_key<true>(reinterpret_cast<const float*>(sig)[o + i], k)) ok = false; }
        else k = (int)reinterpret_cast<const int16_t*>(sig)[o + i];
        lo = min(lo, k); hi = max(hi, k);
      }
    }
    const int vmax = (int)(wave_max_u32((unsigned)hi ^ 0x80000000u) ^ 0x80000000u);
    const int vmin = (int)(~wave_max_u32(~((unsigned)lo ^ 0x80000000u)) ^ 0x80000000u);
    const bool fit = __ballot(!ok) == 0ull && (unsigned)(vmax - vmin) < (unsigned)kCntWindow;
    if (lane == 0) { atomicAdd(&seen, 1); if (fit) atomicAdd(&fits, 1); }
  }
  __syncthreads();
  if (threadIdx.x == 0) a.gate[0] = (seen > 0 && fits * 8 >= seen * 7) ? 1 : 0;
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
  using Rows = KsRows<NS, LG, DTYPE>;

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

    // ---- keys.  Slot s = 4 c + j of group g: component j of chunk c.  A chunk that holds the end of a row was read as the
    // row's LAST four samples (KsRows): its components j >= 4 - t are this lane's, t = samples left at the chunk's start.
    int k[2 * NS];                                       // integer keys (milli-units); a slot without a sample: the group's first key
    unsigned vmask = 0u;                                 // bit s of group g at 16 g + s: the slot holds a sample
    bool bad = false;                                    // float32: a sample of this lane is off the grid
    double f1[2] = {0.0, 0.0}, f2[2] = {0.0, 0.0};       // float32: shifted moment sums of the values
    int i1[2] = {0, 0}, i2[2] = {0, 0};                  // int16: exact moment sums of k - first key
    int kfirst[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const Rows& rw = g ? rw1 : rw0;
      const int n = rows_n(cur, g ? n1 : n0);
      float xf[NS];
      int ki[NS];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if constexpr (DTYPE == 0) { xf[4 * c] = rw.v[c].x; xf[4 * c + 1] = rw.v[c].y; xf[4 * c + 2] = rw.v[c].z; xf[4 * c + 3] = rw.v[c].w; }
        else { ki[4 * c] = (int)rw.v[c].x; ki[4 * c + 1] = (int)rw.v[c].y; ki[4 * c + 2] = (int)rw.v[c].z; ki[4 * c + 3] = (int)rw.v[c].w; }
        const int t = n - (c * 64 + 4 * gl);
        const int first = (t >= 4) ? 0 : ((t <= 0) ? 4 : 4 - t);        // components first .. 3 are samples of this lane
#pragma unroll
        for (int j = 0; j < 4; ++j) vmask |= (j >= first) ? (1u << (16 * g + 4 * c + j)) : 0u;
      }
      // the group's first sample (lane 0, chunk 0, component 0: there for every row of at least 4 samples) stands in the empty slots
      if constexpr (DTYPE == 0) {
        const float xfirst = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, __float_as_int(xf[0])));
        const double K = (double)xfirst;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bool have = (vmask >> (16 * g + s)) & 1u;
          const float x = have ? xf[s] : xfirst;
          int kk;
          const bool ok = grid_key<true>(x, kk);
          bad = bad || !ok;
          k[NS * g + s] = kk;
          const double d = (double)x - K;
          f1[g] += d;
          f2[g] = __fma_rn(d, d, f2[g]);
          if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        kfirst[g] = 0;
        f1[g] = seg_allsum_f64<LG>(f1[g]);
        f2[g] = seg_allsum_f64<LG>(f2[g]);
        const double dn = (double)(g ? n1 : n0);
        const double rn = uniform ? recip[g] : 1.0 / dn;
        const double mu = K + f1[g] * rn, qq = f2[g] - f1[g] * f1[g] * rn;
        f1[g] = mu; f2[g] = qq;
        __builtin_amdgcn_sched_barrier(0);
      } else {
        const int kf = __builtin_amdgcn_ds_bpermute((lane & ~(LG - 1)) << 2, ki[0]);
        kfirst[g] = kf;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const bool have = (vmask >> (16 * g + s)) & 1u;
          const int kk = have ? ki[s] : kf;
          k[NS * g + s] = kk;
          const int d = kk - kf;                         // (|d| <= 65 535 for any int16 row; <= 2 047 for one that fits the window)
          i1[g] += d;
          i2[g] = (int)((unsigned)i2[g] + (unsigned)__mul24(d, d));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- the moments are final here: written at once (a position that turns out not to fit is written again by rank_hist_kernel)
    {
      double mean[2], m2[2];
      if constexpr (DTYPE == 0) {
        mean[0] = f1[0]; m2[0] = f2[0]; mean[1] = f1[1]; m2[1] = f2[1];
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const double S1 = (double)cnt_allsum_i32(i1[g]), S2 = (double)cnt_allsum_i32(i2[g]);   // exact: |S1| < 2^20, S2 < 2^31 (fitting positions)
          const double dn = (double)(g ? n1 : n0);
          const double rn = uniform ? recip[g] : 1.0 / dn;
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
    // ---- the window: the position's smallest key is value 0
    int kmin = k[0], kmax = k[0];
#pragma unroll
    for (int s = 1; s < 2 * NS; ++s) { kmin = min(kmin, k[s]); kmax = max(kmax, k[s]); }
    kmin = cnt_allmin_i32(kmin); kmax = cnt_allmax_i32(kmax);
    {
      const unsigned long long bm = __ballot(bad);
      const bool pos_bad = ((unsigned)(bm >> (lane & 48)) & 0xffffu) != 0u;
      fit = fit && !pos_bad && (unsigned)(kmax - kmin) < (unsigned)kCntWindow;
    }
    // (int16 moments: sums of squares of a position that does not fit may have wrapped; they are not used then)

    unsigned best = 0u, slu = 0u, tsq = 0u;
    double dmax = 0.0;
    unsigned reg[2 * NS];
    if (fit) {
      // byte address of cum[u - 1]: tb + 15 + u + 16 (u >> 7)  (block u >> 7 starts 16 pad bytes later than a flat table would);
      // a slot without a sample: the first two bytes of the first pad, always zero
#pragma unroll
      for (int s = 0; s < 2 * NS; ++s) {
        const unsigned u = (unsigned)(k[s] - kmin);
        const unsigned a = tb + 15u + u + ((u >> 7) << 4);
        reg[s] = ((vmask >> s) & 1u) ? a : tb;
      }
      // (the packed registers are made opaque after every phase: otherwise the compiler keeps the unpacked addresses — 32
      // more registers — alive beside them for the later phases)
#pragma unroll
      for (int s = 0; s < 2 * NS; ++s) asm volatile("" : "+v"(reg[s]));
      __builtin_amdgcn_sched_barrier(0);
      auto clear_table = [&]() {
#pragma unroll
        for (int i = 0; i < 9; ++i) reinterpret_cast<uint4*>(tbl)[gl + 16 * i] = make_uint4(0u, 0u, 0u, 0u);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      auto build = [&](int g) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const unsigned a1 = (reg[NS * g + s] & 0xffffu) + 1u;               // byte of the sample's counter (empty slot: tb + 1, adds 0)
          const unsigned val = ((vmask >> (NS * g + s)) & 1u) ? (1u << ((a1 & 3u) * 8u)) : 0u;
          __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)(a1 & ~3u), val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };
      // counts -> inclusive prefix sums, in place.  Lane gl owns block gl (words 36 gl + 4 .. + 35).
      auto scan = [&]() {
        uint4* blk = reinterpret_cast<uint4*>(__builtin_assume_aligned(tbl + kCntBlockWords * gl + 4, 16));
        // the lane's total first (its words are read again below: 32 registers for them would not fit beside the samples)
        unsigned tot = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const uint4 q = blk[i];
          tot = __builtin_amdgcn_sad_u8(q.x, 0u, tot); tot = __builtin_amdgcn_sad_u8(q.y, 0u, tot);
          tot = __builtin_amdgcn_sad_u8(q.z, 0u, tot); tot = __builtin_amdgcn_sad_u8(q.w, 0u, tot);
        }
        const unsigned base = seg_exscan_add_u32<LG>(tot, gl);                  // samples below the lane's first value (< 256)
        unsigned carry = base | (base << 8);
        carry |= carry << 16;
        blk[-1] = make_uint4(0u, 0u, 0u, base << 24);                            // byte 15 of the pad: cum just below the block
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          unsigned w[16];
#pragma unroll
          for (int i = 0; i < 4; ++i) { const uint4 q = blk[4 * h + i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            // (inline asm: written as v + (v << 8) the compiler multiplies by 0x01010101 with a 64-bit v_mad_u64_u32 per word)
            unsigned v = w[i], t;
            asm("v_lshl_add_u32 %0, %1, 8, %1" : "=v"(t) : "v"(v));
            asm("v_lshl_add_u32 %0, %1, 16, %1" : "=v"(v) : "v"(t));
            v += carry;
            carry = __builtin_amdgcn_perm(v, v, 0x03030303u);                    // the word's last byte in all four
            w[i] = v;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) blk[4 * h + i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
          __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_sched_barrier(0);
      };

      // ---- table A: group 1
      clear_table();
      build(0);
      scan();
#pragma unroll
      for (int s0 = 0; s0 < 2 * NS; s0 += 8) {
        unsigned v[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                             // A[u-1], A[u]: two byte reads (an unaligned 16-bit read costs the LDS pipe a pass per lane)
          const CntLdsU8 p = (CntLdsU8)(uintptr_t)(reg[s0 + e] & 0xffffu);
          v1[e] = p[0]; v[e] = p[1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s = s0 + e;
          const unsigned cA = v[e], cA1 = v1[e];
          if (s >= NS) slu += cA + cA1;                                           // group 2: #{a < x} + #{a <= x}
          reg[s] = (reg[s] & 0xffffu) | (cA << 16) | ((cA - cA1) << 24);
          asm volatile("" : "+v"(reg[s]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);

      // ---- table B: group 2
      clear_table();
      build(1);
      scan();
#pragma unroll
      for (int s0 = 0; s0 < 2 * NS; s0 += 8) {
        unsigned v[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {                                             // B[u-1], B[u]
          const CntLdsU8 p = (CntLdsU8)(uintptr_t)(reg[s0 + e] & 0xffffu);
          v1[e] = p[0]; v[e] = p[1];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int s = s0 + e;
          const unsigned cB = v[e], cB1 = v1[e];
          const unsigned cA = (reg[s] >> 16) & 0xffu, ta = reg[s] >> 24;
          const unsigned t = ta + (cB - cB1);                                      // size of the sample's pooled tie group (0: empty slot)
          tsq += __umul24(t, t);
          unsigned num;                                                             // |A n1 - B n0| <= 65 025
          asm("v_sad_u32 %0, %1, %2, 0" : "=v"(num) : "v"(__umul24(cA, (unsigned)n1)), "v"(__umul24(cB, (unsigned)n0)));
          best = max(best, num);
          reg[s] = (num << 16) | (cA << 8) | cB;
          asm volatile("" : "+v"(reg[s]));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }

    // the next item's rows: requested here, used at the top of the next iteration
    const Item nxt = describe(it + wave_stride);
    Rows nx0, nx1;
    nx0.request(args.sig0, nxt.o0, rows_n(nxt, nxt.n0), gl);
    nx1.request(args.sig1, nxt.o1, rows_n(nxt, nxt.n1), gl);

    if (__ballot(fit) != 0ull) {
      best = fit ? seg_allmax_u32<LG>(best) : 0u;
      // ---- the float form of D at the samples that reach the integer maximum
      const double dn0 = (double)n0, dn1 = (double)n1;
      double r0, r1;
      if (uniform) { r0 = recip[0]; r1 = recip[1]; } else { r0 = 1.0 / dn0; r1 = 1.0 / dn1; }
#if !(NMOD_CNT_SKIP & 1)
#pragma unroll
      for (int s = 0; s < 2 * NS; ++s) {
        const bool hit = fit && (reg[s] >> 16) == best && best != 0u;
        if (__ballot(hit) != 0ull) {
          const double d = fabs(hist_exact_quot((int)((reg[s] >> 8) & 0xffu), dn0, r0) - hist_exact_quot((int)(reg[s] & 0xffu), dn1, r1));
          dmax = hit ? fmax(dmax, d) : dmax;
        }
        if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
#endif
      dmax = seg_allmax_f64<LG>(dmax);
      const unsigned SLU = pos_allsum_u32<LG>(slu);
      const unsigned TSQ = pos_allsum_u32<LG>(tsq);
      if (fit && gl == 0) {
        args.ks_num[pos] = best;
        args.ks_d_ref[pos] = dmax;
        args.mwu_s[pos] = 2ull * (unsigned long long)n0 * (unsigned long long)n1 - (unsigned long long)SLU;
        args.tie[pos] = (unsigned long long)TSQ - (unsigned long long)(n0 + n1);
        if (args.tied) args.tied[pos] = (TSQ != (unsigned)(n0 + n1)) ? 1 : 0;
      }
    }
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
