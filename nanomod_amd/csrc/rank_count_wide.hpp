// K1, counting form for any coverage (round 5) — EVENT-LIKE positions of every shape the reference sees, all tests or KS only:
// one position per wave, the smaller group S of up to 64 RS samples (RS = 1 ... 16 registers per lane: 64 ... 1 024 samples), the
// larger group Q of up to 4 095.  rank_count.hpp is the same idea for four 200 v 200-like positions per wave with byte tables; here
// ONE table serves both groups: a 32-bit word per value of a 2 048-value window centred on S,
//   count     ds_add_rtn_u32 of 1 (a sample of S) or 0x10000 (a sample of Q) at the word of its value; the word that comes back
//             holds the copies of the value counted before it, in both groups: the sample's arrival number p
//   scan      prefix sums in place, both halves at once (S's half stays below 2^16): word v = A[v] | B[v] << 16
//   look up   every sample of S reads the words just below and at its value (Q — most of the samples at skewed coverage,
//             configs[4]: ~1 131 v ~57 — is streamed once and never looked up)
// and from those (rank_count.hpp has the derivations):
//   KS        max over S's samples of |A[v] nQ - B[v] nS| and |A[v-1] nQ - B[v-1] nS| (one v_dot2_i32_i16 against (nQ, -nS)); the float
//             form of D at the candidates that reach the maximum
//   MWU       sum over S of (B[v-1] + B[v]) = sum (#{q < s} + #{q <= s}); mwu_s is that sum when S is group 1, else 2 n0 n1 - it
//   ties      sum_v t^3 - n = 3 sum over all arrivals of (p^2 + p)   (sum_{p < t} (3 p^2 + 3 p + 1) = t^3): no limit on the copies
// Per streamed int16 sample 12.5 instructions (pairs: v_pk_sub_i16, v_pk_min_u16 against the window size, v_dot2 moment sums about the
// window's centre, v_mad_u32_u16 with op_sel for the address) against 27 for the sorting form's binary search alone; float32 adds
// the sample's key (grid_key, 9) and its fp64 moment sums (4).  KS only: the adds return nothing, no moments, no MWU sums.
// Outliers (round 6).  Stored events are clipped at +-5 units (myRefBaseSignalAnnotation.py:251-259) and a mis-segmented read sits
// anywhere in that range: one such sample in ~1 200 must not cost the position its form.  The window is centred on a robust centre
// of S (its mean, then the mean of the samples within 1 024 of that); a sample of either group outside the window — a TAIL sample,
// at most kCwTail = 64 per position — goes to a list in LDS instead of the table: key | group << 16, appended under an exec mask on
// the rare path (int16 rows: one v_or per pair of samples keeps watch, the samples of a chunk group are looked at again only
// when the wave saw one).  The samples below the window enter the scan as its carry-in, so every in-window look-up is the count
// over the WHOLE group; the tail samples — one per lane — are finished exactly by an all-pairs pass over the list (readlane
// broadcast): their counts at and just below their values (KS candidates and MWU terms of S's tail samples), their copies among
// the tail (tie term), their true moment terms.
// A position is left to the sorting forms (flag byte per list entry, compacted into a work list by cnt_compact_kernel; the sorting
// kernel of the class walks that list where the class's gate is set: RankStatsArgs::alt_*) when a group is out of range, a float32
// sample is off the milli-unit grid, or more than kCwTail samples fall outside the window.  Whether a class of a batch is
// event-like at all is decided by cnt_wide_probe_kernel on a sample of its positions (gate): continuous signals pay the probe
// and two empty launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rank_count.hpp"
#include "rank_stats_launch.hpp"

#ifndef NMOD_CW_OR3
#define NMOD_CW_OR3 1        // 1: one v_or3_b32 per two pairs of int16 samples keeps watch for samples outside the window
#endif

namespace nmod {

constexpr int kCwMaxQ = 4095;                                  // Q's half of a word, and |A nQ - B nS| through 16-bit dot products
constexpr int kCwWindow = 2048;                                // values the table covers: 64 lane blocks of 32 entries
constexpr int kCwTail = 64;                                    // tail samples (outside the window) a position may have: one per lane
constexpr int kCwTableWords = 64 * 36 + 8;                     // a lane's block: 4 pad words + up to 32 entries; + the dump entry
constexpr int kCwWaveWords = kCwTableWords + kCwTail;          // + the tail list: 9 504 B per wave, four blocks of four waves per CU
__host__ __device__ constexpr size_t rank_count_wide_lds_bytes() { return (size_t)kWavesPerBlock * kCwWaveWords * 4; }

struct CntWideArgs {
  RankStatsArgs rs;                                            // rows, class lists, outputs (cnt_gate / cnt_done unused here)
  const int32_t* gates;                                        // [class] the probe's verdict (cnt_wide_probe_kernel)
  const int32_t* segs;                                         // [0] number of classes to try, [1 + i] their ids
  int32_t* work_list; int32_t* work_meta;                      // what the form hands on to the class's sorting form: the layout of pos_list / class_meta
                                                               // (work_meta[c] = count — zeroed by the probe —, [kClassStride + c] = the list's offset)
};

typedef unsigned CntWU2 __attribute__((ext_vector_type(2)));

// ---- probe: 64 sampled positions per class (one block per class of the batch that can take the form); event-like = sizes in
// range, every sample on the grid, all keys of the position within 2 048 milli-units.
struct CntWideProbeArgs {
  const void* sig0; const void* sig1; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1; int64_t npos;
  const int32_t* pos_list; const int32_t* class_meta;          // null / null: one class, all of [0, npos)
  int32_t nclasses; int32_t cls[kClassStride]; int32_t max_s[kClassStride];      // block b probes class cls[b]
  int32_t* gate;                                                 // [kClassStride]: gate[class]
  int32_t* segs;                                                 // [0] nclasses, [1 + b] cls[b]: the list the later kernels walk
  int32_t* work_meta;                                            // [class] = 0, [kClassStride + class] = the class list's offset: the work lists start empty
  int32_t min_q;                                                 // KS-only mode: the form takes positions whose larger group holds at least this many (0: all)
};
template <int DTYPE, bool KS>                                  // (KS: only to give each translation unit's instance its own name)
__global__ __launch_bounds__(1024) void cnt_wide_probe_kernel(CntWideProbeArgs a) {
  __shared__ int fits, seen, looked, far, tot;
  const int cid = a.cls[blockIdx.x];
  if (threadIdx.x == 0) { fits = 0; seen = 0; looked = 0; far = 0; tot = 0; }
  __syncthreads();
  int64_t count = a.npos;
  const int32_t* list = nullptr;
  if (a.pos_list) { count = a.class_meta[cid]; list = a.pos_list + a.class_meta[kClassStride + cid]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t nsamp = count < kCntProbeSamples ? count : kCntProbeSamples;
  for (int64_t j = wave; j < nsamp; j += 16) {
    // one position per stratum of count / nsamp list entries, at a pseudo-random place inside it (evenly strided samples alias
    // with anything periodic in the batch: round 6 found half of them on the benchmark generator's planted positions)
    const int64_t s_lo = (j * count) / nsamp, s_len = ((j + 1) * count) / nsamp - s_lo;
    const int64_t li = s_lo + (int64_t)(((uint64_t)(j + 1) * 0x9E3779B97F4A7C15ull >> 20) % (uint64_t)(s_len > 0 ? s_len : 1));
    const int64_t pos = list ? (int64_t)list[li] : li;
    bool ok = true;
    int nn[2]; int64_t oo[2];
    for (int g = 0; g < 2; ++g) {
      const int64_t st = g ? a.stride1 : a.stride0;
      const int64_t* off = g ? a.off1 : a.off0;
      oo[g] = st > 0 ? pos * st : off[pos];
      nn[g] = st > 0 ? (int)st : (int)(off[pos + 1] - oo[g]);
    }
    auto key_at = [&](int g, int i, int& k) -> bool {
      const void* sig = g ? a.sig1 : a.sig0;
      if constexpr (DTYPE == 0) return grid_key<true>(reinterpret_cast<const float*>(sig)[oo[g] + i], k);
      else if constexpr (DTYPE == 2) return cnt_int_key(reinterpret_cast<const float*>(sig)[oo[g] + i], k);
      else { k = (int)reinterpret_cast<const int16_t*>(sig)[oo[g] + i]; return true; }
    };
    const int gs = nn[1] < nn[0] ? 1 : 0;                  // S = the smaller group (ties: group 1), as in cw_segment
    const int m = nn[gs], q = nn[1 - gs];
    if (m < 1 || m > a.max_s[blockIdx.x] || q > kCwMaxQ) ok = false;
    // the window the kernel would pick: the mean of S, then the mean of S's samples within 1 024 of it
    int tails = 0;
    if (ok) {
      long long s1 = 0;
      for (int i = lane; i < m; i += 64) { int k; if (!key_at(gs, i, k)) ok = false; s1 += k; }
      s1 = (long long)wave_sum_u64((unsigned long long)s1);
      const int c0 = (int)__builtin_rintf((float)s1 / (float)m);
      long long s2 = 0; unsigned c2 = 0;
      for (int i = lane; i < m; i += 64) { int k; key_at(gs, i, k); const int d = k - c0; if ((unsigned)(d + 1024) < 2048u) { s2 += d; ++c2; } }
      s2 = (long long)wave_sum_u64((unsigned long long)s2); c2 = (unsigned)wave_sum_u64((unsigned long long)c2);
      const int c = c2 ? c0 + (int)__builtin_rintf((float)s2 / (float)c2) : c0;
      const int base = max(-32768, min(c - 1024, 32768 - 2048));
      for (int g = 0; g < 2; ++g)
        for (int i = lane; i < nn[g] && i <= kCwMaxQ; i += 64) { int k; if (!key_at(g, i, k)) ok = false; tails += ((unsigned)(k - base) >= 2048u) ? 1 : 0; }
      tails = (int)wave_sum_u64((unsigned long long)tails);
    }
    // (half the list: the probe's window is only close to the kernel's; the class whose groups both exceed 1 024 samples — max_s 2 048,
    // rank_count_value.hpp — lists 128)
    const bool fit = __ballot(!ok) == 0ull && tails <= (a.max_s[blockIdx.x] > 1024 ? 96 : kCwTail / 2);
    if (lane == 0) {
      atomicAdd(&looked, 1);
      if (q >= a.min_q) { atomicAdd(&seen, 1); if (fit) { atomicAdd(&fits, 1); atomicAdd(&far, tails); atomicAdd(&tot, m + q); } }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // The share of tail samples the form still wins at, against the sorting form of the class (tools/outlier_sweep.sh, outlier_sweep_ks.sh on
    // configs[4]'s sizes and 500 v 500; 4 of 5 outliers over +-5 units are tail samples): all tests — ahead at 10 per mille outliers, level at 30;
    // KS only, where the form's lead on clean rows is +24 % (int16) / +4 .. +24 % (float32) — level at ~8 / ~4 per mille.
    constexpr int kTailShare = KS ? (DTYPE == 1 ? 6 : 4) : 20;                // per mille of the sampled positions' samples
    // (KS-only: at least half of the class must be of the form's sizes — it walks the whole class list)
    // (the class whose groups both exceed 1 024 samples: its sorting forms are 2 - 10 x behind the value-domain form at any share it can hold)
    const int share = a.max_s[blockIdx.x] > 1024 ? 1000 : kTailShare;
    a.gate[cid] = (seen > 0 && fits * 8 >= seen * 7 && seen * 2 >= looked && (long long)far * 1000 <= (long long)tot * share) ? 1 : 0;
    a.segs[1 + blockIdx.x] = cid;
    a.work_meta[cid] = 0; a.work_meta[kClassStride + cid] = a.pos_list ? a.class_meta[kClassStride + cid] : 0;
    if (blockIdx.x == 0) a.segs[0] = a.nclasses;
  }
}

// ---- the work list of the sorting form that follows, for a class whose gate is set: the positions rank_count_wide_kernel hands on.
// The waves append them themselves: the positions whose SIZES are not the form's where the headers are loaded, 64 at a time (one
// atomicAdd per batch of headers), a position whose samples turn out not to fit by one atomicAdd of its own (the probe saw 7 of 8
// fit; usually far fewer are handed on).  (Queues of 64 rejected positions per wave, in a register or in LDS, flushed inside the
// loop made the float32 instances spill 1.2 KB per lane.)  (Round 5 wrote a flag byte per position and ran
// cnt_compact_kernel over them — a serial loop over the classes, 0.43 ms per 10 M positions, 4 % of the ragged KS pass.)  The
// list's order is whatever order the waves flush in; every position's numbers are its own.

// ---- wave-wide helpers of this form: four DPP steps inside each 16-lane row, two row broadcasts into lane 63, one v_readlane
// (wave_ops.hpp's reductions read four row results and combine them with scalar instructions: eleven operations instead of seven)
__device__ __forceinline__ unsigned cw_wave_sum_u32(unsigned v) {
  v += (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2)>(0, (int)v);
  v += (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1)>(0, (int)v);
  v += (unsigned)dpp_i<kDppRowHalfMirror>(0, (int)v);
  v += (unsigned)dpp_i<kDppRowMirror>(0, (int)v);
  v += (unsigned)dpp_i<kDppRowBcast15, 0xA>(0, (int)v);     // rows 1, 3 += rows 0, 2
  v += (unsigned)dpp_i<kDppRowBcast31, 0xC>(0, (int)v);     // rows 2, 3 += row 1
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned cw_wave_max_u32(unsigned v) {
  v = max(v, (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2)>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1)>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowHalfMirror>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowMirror>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowBcast15, 0xA>((int)v, (int)v));
  v = max(v, (unsigned)dpp_i<kDppRowBcast31, 0xC>((int)v, (int)v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned cw_wave_max_pk_u16(unsigned v) {          // both 16-bit halves at once
  auto mx = [](unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(CntU2, a), __builtin_bit_cast(CntU2, b))); };
  v = mx(v, (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2)>((int)v, (int)v));
  v = mx(v, (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1)>((int)v, (int)v));
  v = mx(v, (unsigned)dpp_i<kDppRowHalfMirror>((int)v, (int)v));
  v = mx(v, (unsigned)dpp_i<kDppRowMirror>((int)v, (int)v));
  v = mx(v, (unsigned)dpp_i<kDppRowBcast15, 0xA>((int)v, (int)v));
  v = mx(v, (unsigned)dpp_i<kDppRowBcast31, 0xC>((int)v, (int)v));
  return __builtin_amdgcn_readlane(v, 63);
}

// 1 / n, correctly rounded, n <= 4 095 (hist_exact_quot's r): one scalar load instead of a float64 division per position
struct CwRcpTable {
  double v[4096];
  constexpr CwRcpTable() : v{} { for (int i = 1; i < 4096; ++i) v[i] = 1.0 / (double)i; }
};
static __device__ const CwRcpTable kCwRcp = CwRcpTable();

// one class of the batch (a segment of the class lists): positions start, start + stride, ... of its list
template <int DTYPE, int RS, bool KS>
__device__ __forceinline__ void cw_segment(const RankStatsArgs& args, int32_t* work_list, int32_t* work_cnt, unsigned* tbl, int64_t count, int64_t loff, const int32_t* list,
                                           int64_t start, int64_t wave_stride, int lane) {
  constexpr int RDT = (DTYPE == 1) ? 1 : 0;
  using Q4Raw = typename std::conditional<RDT == 0, KsF4, KsS4>::type;
  using Q1Raw = typename std::conditional<RDT == 0, float, int16_t>::type;
  const unsigned tb = (unsigned)(uintptr_t)(CntLdsU32)tbl;
  const unsigned tbE = tb + 16u;                           // entry 0 of block 0 (after its pad)

  auto key_of = [&](float x, int& k) -> bool {            // float32 rows: the integer key and whether the sample has one
    if constexpr (DTYPE == 2) return cnt_int_key(x, k);
    else return grid_key<true>(x, k);
  };
  const CntU2 one2 = {1, 1};
  const CntS2 sone2 = {1, 1};
  const CntU2 hi1 = {0, 1};

  // Headers: 64 positions at a time, one per lane, by vector loads — list entry, row offsets, sizes — and everything that follows
  // from them computed there, 64 positions per instruction: which group is S, whether the sizes are this form's, the rows' addresses.
  // A position's header is read out of its lane (six v_readlane).  (As scalar loads per position the entries were two dependent
  // memory latencies on the LDS's counter; as scalar arithmetic per position the rest was 50 of its 275 scalar instructions.)
  struct Hdr { int pos; int m, q; bool swap, fit; const Q1Raw* row_s; const Q1Raw* row_q; };
  int hb_pos = 0; unsigned hb_mq = 0u; uint64_t hb_rs = 0ull, hb_rq = 0ull;
  auto load_batch = [&](int64_t it0) {                     // lane j: position it0 + j * wave_stride of the list
    const int64_t itj = it0 + (int64_t)lane * wave_stride;
    int64_t p = 0, o0 = 0, o1 = 0; int n0 = 0, n1 = 0;
    if (itj < count) {
      p = list ? (int64_t)list[itj] : itj;
      if (args.stride0 > 0) { o0 = p * args.stride0; n0 = (int)args.stride0; } else { o0 = args.off0[p]; n0 = (int)(args.off0[p + 1] - o0); }
      if (args.stride1 > 0) { o1 = p * args.stride1; n1 = (int)args.stride1; } else { o1 = args.off1[p]; n1 = (int)(args.off1[p + 1] - o1); }
    }
    const bool sw = n1 < n0;                               // S = the smaller group (ties: group 1)
    const int mr = sw ? n1 : n0, qr = sw ? n0 : n1;
    const bool ok = mr >= 1 && mr <= 64 * RS && qr >= 4 && qr <= kCwMaxQ && (!KS || qr >= kCwKsMinQ);   // (otherwise: left to the sorting form)
    const Q1Raw* dummy = reinterpret_cast<const Q1Raw*>(kKsBig4);                                       // (read as a row of one / four samples)
    hb_pos = (int)p;
    {
      // positions whose sizes are not this form's (KS only: the larger group below kCwKsMinQ — up to half of a class) go to the class's
      // work list here, 64 headers at a time: one atomicAdd reserves the slots of the batch
      const unsigned long long mk = __ballot(itj < count && !ok);
      if (mk != 0ull) {                                      // (wave-uniform)
        int at = 0;
        if (lane == 0) at = atomicAdd(work_cnt, (int)__popcll(mk));
        at = __builtin_amdgcn_readfirstlane(at);
        if (itj < count && !ok) work_list[loff + at + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u))] = (int)p;
      }
    }
    hb_mq = (unsigned)(ok ? mr : 1) | ((unsigned)(ok ? qr : 4) << 12) | (sw ? 1u << 24 : 0u) | (ok ? 1u << 25 : 0u);
    hb_rs = (uint64_t)(uintptr_t)(ok ? reinterpret_cast<const Q1Raw*>(sw ? args.sig1 : args.sig0) + (sw ? o1 : o0) : dummy);
    hb_rq = (uint64_t)(uintptr_t)(ok ? reinterpret_cast<const Q1Raw*>(sw ? args.sig0 : args.sig1) + (sw ? o0 : o1) : dummy);
  };
  auto rl64 = [&](uint64_t v, int j) -> uint64_t {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, j);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), j);
    return ((uint64_t)hi << 32) | lo;
  };
  int hb_j = 0;                                            // lane of the header read next
  auto load_hdr = [&](int64_t it) -> Hdr {                 // called with it = start, start + wave_stride, ... in order
    if (hb_j == 0) load_batch(it);
    Hdr h;
    h.pos = __builtin_amdgcn_readlane(hb_pos, hb_j);
    const unsigned mq = (unsigned)__builtin_amdgcn_readlane((int)hb_mq, hb_j);
    h.m = (int)(mq & 0xfffu); h.q = (int)((mq >> 12) & 0xfffu); h.swap = ((mq >> 24) & 1u) != 0u; h.fit = ((mq >> 25) & 1u) != 0u;
    h.row_s = reinterpret_cast<const Q1Raw*>((uintptr_t)rl64(hb_rs, hb_j)); h.row_q = reinterpret_cast<const Q1Raw*>((uintptr_t)rl64(hb_rq, hb_j));
    hb_j = (hb_j + 1) & 63;
    return h;
  };
  // Rows: requested a position ahead — after this position's Q has been streamed, while its table is scanned and its sums are
  // reduced: up to PF chunks of 256 samples of Q (four per lane), the <= 255 that remain one per lane, S one per lane and register.
  // (With one chunk in flight and the requests at the top the waves waited a memory latency per 50 instructions.)  A sample that
  // does not exist is read as the row's last one and masked.
  constexpr int PF = (RS == 16) ? (RDT == 0 ? 1 : 2) : 4;
  constexpr bool PS = RS <= 8;                             // S a position ahead as well (RS = 16: sixteen more live registers spill)
  struct Rows { Q1Raw s[RS]; Q4Raw qa[PF]; Q1Raw rt[4]; Q1Raw q0; };
  auto load_at = [&](const void* row, unsigned byte_off, auto tag) { return ks_global_load<decltype(tag)>(reinterpret_cast<const char*>(row) + byte_off); };
  // (every load is issued whatever the position looks like — a chunk or a position that does not exist reads the 16-byte dummy:
  // a conditional load makes the loaded registers phi nodes, and the copies the compiler places for them wait for the data
  // right where the request was meant to run ahead)
  auto request = [&](const Hdr& h, Rows& R, bool with_q, bool with_s) {      // (a position past the list's end: the dummy rows, load_batch)
    const Q1Raw* row_s = h.row_s; const Q1Raw* row_q = h.row_q;
    const int m_ = h.m, q_ = h.q;
    const int full_ = q_ / 256;
    if (with_q) {
      // (a chunk the row does not have reads the row's last four samples: no pointer selects)
#pragma unroll
      for (int j = 0; j < PF; ++j) R.qa[j] = load_at(row_q, (unsigned)min(j * 256 + 4 * lane, q_ - 4) * (unsigned)sizeof(Q1Raw), Q4Raw());
#pragma unroll
      for (int j = 0; j < 4; ++j) R.rt[j] = load_at(row_q, (unsigned)min(full_ * 256 + j * 64 + lane, q_ - 1) * (unsigned)sizeof(Q1Raw), Q1Raw());
      if constexpr (RDT == 0) R.q0 = load_at(row_q, 0u, Q1Raw());
    }
    if (with_s) {
#pragma unroll
      for (int r = 0; r < RS; ++r) R.s[r] = load_at(row_s, (unsigned)min(r * 64 + lane, m_ - 1) * (unsigned)sizeof(Q1Raw), Q1Raw());
    }
  };
  Hdr nxt = load_hdr(start);
  Rows rows;
#pragma unroll
  for (int r = 0; r < RS; ++r) rows.s[r] = Q1Raw(0);
#pragma unroll
  for (int j = 0; j < 4; ++j) rows.rt[j] = Q1Raw(0);
  rows.q0 = Q1Raw(0);
  request(nxt, rows, true, PS);
  for (int64_t it = start; it < count; it += wave_stride) {
    const Hdr cur = nxt;
    nxt = load_hdr(it + wave_stride);
    const int64_t pos = cur.pos;
    const bool swap = cur.swap;
    const int m = cur.m, q = cur.q;                        // (a position that is not this form's: 1 and 4, fit = false)
    const Q1Raw* rowq = cur.row_q;
    bool fit = cur.fit;                                    // (wave-uniform)
    const int mm = fit ? m : 0, qq = fit ? q : 0;
    const double rcp_m = kCwRcp.v[mm], rcp_q = kCwRcp.v[qq];   // (requested here, used at the end: scalar loads)
    const int full = qq / 256;
    const int tail = (qq - full * 256 + 63) / 64;
    float xq0 = 0.0f;
    if constexpr (RDT == 0) xq0 = (float)rows.q0;
    if constexpr (!PS) request(cur, rows, false, true);

    // ---- S: sample r * 64 + lane in register r.  Keys, float32 moments, range.
    int ks[RS];
    bool bad = false;
    double ms1 = 0.0, ms2 = 0.0;                           // float32 rows: shifted moment sums
    float xs0 = 0.0f;
    {
      Q1Raw (&raw)[RS] = rows.s;
      if constexpr (RDT == 0) {
        xs0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)raw[0])));
        const double K = (double)xs0;
#pragma unroll
        for (int r = 0; r < RS; ++r) {
          const bool have = r * 64 + lane < mm;
          const float x = have ? (float)raw[r] : xs0;
          int k;
          const bool ok = key_of(x, k);
          bad = bad || !ok;
          ks[r] = k;
          if constexpr (DTYPE == 0 && !KS) { const double d = (double)x - K; ms1 += d; ms2 = __fma_rn(d, d, ms2); }
        }
      } else {
        const int ks0 = __builtin_amdgcn_readfirstlane((int)raw[0]);
#pragma unroll
        for (int r = 0; r < RS; ++r) ks[r] = (r * 64 + lane < mm) ? (int)raw[r] : ks0;
      }
    }
    fit = fit && __ballot(bad) == 0ull;
    // the window: 2 048 values around a robust centre of S — its mid-range where S spans less than three quarters of a window (no
    // far outlier in S: the common case), else its mean, then the mean of its samples within 1 024 of that (one mis-segmented read among ~57 moves
    // the mean by ~90 milli-units and not the second estimate); samples of either group outside it are the position's tail
    // (header comment).  E = 1 << lgE entries per lane block.  (A window of 512 / 1 024 values
    // where S's range allowed it — a shorter clear and scan — was measured in round 5: event-like rows at sigma >= 0.1 never took
    // it, and its wave-uniform branches cost every position 4-8 %.)
    constexpr int lgE = 5;
    constexpr int W = 64 << lgE;
    int centre;
    bool s_narrow;
    {
      int kmx = ks[0], kmn = ks[0];
#pragma unroll
      for (int r = 1; r < RS; ++r) { kmx = max(kmx, ks[r]); kmn = min(kmn, ks[r]); }
      const unsigned P = cw_wave_max_pk_u16(((unsigned)(kmx + 32768) & 0xffffu) | ((unsigned)(32767 - kmn) << 16));   // (a float32 key out of range: `bad`)
      const int smax = (int)(P & 0xffffu) - 32768, smin = 32767 - (int)(P >> 16);
      s_narrow = (smax - smin) < (kCwWindow * 3) / 4;
      centre = (smin + smax) >> 1;                         // S within 1 536 values: the window on its mid-range holds all of S with >= 256 values to spare on either side
    }
    if (!s_narrow) {                                       // (wave-uniform) an outlier in S: its mean, then the mean of the samples within 1 024 of that
      int s1 = 0;
#pragma unroll
      for (int r = 0; r < RS; ++r) s1 += (r * 64 + lane < mm) ? ks[r] : 0;
      const float rm_f = (float)rcp_m;
      const int c0 = (int)__builtin_rintf((float)(int)cw_wave_sum_u32((unsigned)s1) * rm_f);
      unsigned pk = 0u;                                    // count << 21 | sum of (k - c0 + 1 024) over the samples within 1 024 of c0
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        const unsigned dd = (unsigned)(ks[r] - c0 + 1024);
        pk += (r * 64 + lane < mm && dd < 2048u) ? dd + (1u << 21) : 0u;
      }
      const unsigned P2 = cw_wave_sum_u32(pk);
      const int cnt2 = (int)(P2 >> 21), sd2 = (int)(P2 & 0x1fffffu) - 1024 * cnt2;
      centre = cnt2 > 0 ? c0 + (int)__builtin_rintf((float)sd2 * __builtin_amdgcn_rcpf((float)cnt2)) : c0;
    }
    int base = centre - (W >> 1);
    base = max(-32768, min(base, 32768 - W));             // (inside the int16 domain: k - base mod 2^16 cannot alias into the window)
    const int c = base + (W >> 1);                         // moments are taken about the centre: |k - c| <= 1 024 inside the window

    unsigned sp = 0u, sp2 = 0u;                            // over the arrivals of both groups: p = earlier copies of the sample's value
    int is1 = 0, iq1 = 0; unsigned is2 = 0u, iq2 = 0u;     // int16 rows: exact sums of k - c
    double mq1 = 0.0, mq2 = 0.0;
    unsigned mws = 0u; int vmax = 0, vmin = 0; unsigned best = 0u;
    double dmax = 0.0;
    constexpr bool KEEPA = RS <= 8;                        // the samples' table addresses stay in registers (RS = 16: recomputed from the keys)
    unsigned addr[KEEPA ? RS : 1];
    unsigned inmask = 0u;                                  // bit r: sample r of S exists and lies inside the window
    int nt = 0, nt_q_pol = 0, nt_s_ = 0;                   // tail samples of the position (wave-uniform), those of Q, those of S; lane i < nt holds sample i:
    bool t_val = false; int t_key = 0; unsigned t_grp = 0u, t_le = 0u, t_lt = 0u, t_p = 0u;
    int ts_d = 0, tq_d = 0; unsigned tq_dw2 = 0u; unsigned long long ts_dd = 0ull, tq_dd = 0ull;   // the tail's moment terms (wave-uniform scalars, int16 rows)
    const unsigned tail_b = tb + (unsigned)kCwTableWords * 4u;
    int listed = 0;                                        // samples in the tail list so far (wave-uniform: a scalar register, not a counter in LDS)
    // the lanes whose sample lies outside the window append key | group << 16 to the tail list: called by the whole wave; the
    // compare's lane mask is the ballot, a lane's slot is the list's length + the tail lanes below it (v_mbcnt) — no atomic, no
    // round trip to LDS on the way (one ds_add_rtn per appended sample, each waiting for the LDS queue to drain, was 600 wave
    // cycles per tail sample)
    auto tail_push = [&](bool is_tail, int k, unsigned grp) {
      const unsigned long long mk = __ballot(is_tail);
      if (mk != 0ull) {                                    // (wave-uniform)
        const unsigned idx = (unsigned)listed + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (is_tail && idx < (unsigned)kCwTail) *(CntLdsU32)(uintptr_t)(tail_b + (idx << 2)) = ((unsigned)k & 0xffffu) | (grp << 16);
        listed += (int)__popcll(mk);
      }
    };
    const CntS2 qm = {(short)q, (short)-m};
    constexpr int nch = 1 << (lgE - 2);                    // 16-byte chunks of a lane block
    uint4* blk = reinterpret_cast<uint4*>(__builtin_assume_aligned(reinterpret_cast<char*>(tbl) + lane * ((4 << lgE) + 16) + 16, 16));
    auto entry = [&](unsigned u) -> unsigned { return tbE + (u << 2) + ((u >> lgE) << 4); };
    auto below_of = [&](int r, bool have) -> unsigned {      // LDS address of the word just below sample r's own (no sample in the window: the two zero words at the head of block 0's pad)
      if constexpr (KEEPA) return addr[r] - 4u;              // (tb + 4 for a sample that is not in the window)
      else return have ? entry((unsigned)(ks[r] - base)) - 4u : tb;
    };
    auto in_window = [&](int r) -> bool { if constexpr (KEEPA) return true; else return ((inmask >> r) & 1u) != 0u; };   // (RS = 16: the addresses are not kept)
    if (fit) {
      {                                                    // ---- clear
        unsigned z = 0u;
        asm volatile("" : "+v"(z));
#pragma unroll
        for (int i = 0; i < 9; ++i) if (i <= nch) blk[i - 1] = make_uint4(z, z, z, z);
        *(CntLdsU32)(uintptr_t)entry((unsigned)W) = z;     // the dump entry
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      // count one sample; the word that comes back holds the copies of its value counted before it (both groups): its arrival
      // number p.  A sample that does not exist, or lies outside the window, adds nothing: to the dump entry (float32 rows, S) —
      // which then stays 0, p = 0 — or as 0x10000 to it (int16 rows of Q, where the add is unconditional: the arrival numbers the
      // dump entry hands back are 0, 1, 2, ... and their share of the tie sums is taken out again below).
      // (Using the word four arrivals later, when it has long returned, changed nothing: measured, profiles/r5_count_wide_ab.txt.)
      auto arrive = [&](unsigned a, unsigned inc, bool have) {
        if constexpr (KS && DTYPE != 2) {                  // (KS only: no tie term — the add returns nothing)
          __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)a, have ? inc : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else if constexpr (KS) {                         // (keys of float64 samples: whether any value occurs twice is reported, RankStatsArgs::tied)
          sp |= __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)a, have ? inc : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        } else {
          const unsigned old = __hip_atomic_fetch_add((CntLdsU32)(uintptr_t)a, have ? inc : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          const unsigned p = __builtin_amdgcn_udot2(__builtin_bit_cast(CntU2, old), one2, 0u, false);
          sp2 = __umul24(p, p) + sp2; sp += p;
        }
      };
      // ---- S
      bool s_tail = false;
#pragma unroll
      for (int r = 0; r < RS; ++r) {
        const bool have = r * 64 + lane < m;
        const unsigned u = (unsigned)(ks[r] - base);
        const bool in = have && u < (unsigned)W;
        inmask |= in ? (1u << r) : 0u;
        s_tail = s_tail || (have && !in);
        const unsigned a_ = in ? entry(u) : tb + 4u;             // (no sample in the window: a zero word of block 0's pad; nothing is added, nothing comes back)
        if constexpr (KEEPA) addr[r] = a_;
        arrive(a_, 1u, in);
        if constexpr (RDT == 1 && !KS) { const int d = in ? ks[r] - c : 0; is1 += d; is2 += (unsigned)__mul24(d, d); }
      }
      if (__ballot(s_tail) != 0ull) {                        // (rare; wave-uniform) samples of S outside the window
#pragma unroll
        for (int r = 0; r < RS; ++r) tail_push(r * 64 + lane < m && (unsigned)(ks[r] - base) >= (unsigned)W, ks[r], 0u);
      }

      // ---- Q, streamed once
      const double KQ = (double)xq0;
      const unsigned cc = ((unsigned)c & 0xffffu) * 0x10001u, hw2 = (unsigned)(W >> 1) * 0x10001u, w2 = (unsigned)W * 0x10001u;
      auto q_f32 = [&](float x, bool have) {
        int k;
        const bool ok = key_of(x, k);
        bad = bad || !ok;
        if constexpr (DTYPE == 0 && !KS) { const double d = (double)x - KQ; mq1 += d; mq2 = __fma_rn(d, d, mq2); }
        const unsigned u = min((unsigned)(k - base), (unsigned)W);
        const bool in = have && u < (unsigned)W;
        arrive(entry(in ? u : (unsigned)W), 0x10000u, in);
        tail_push(have && !in, k, 1u);
      };
      unsigned watch = 0u;                      // OR of the pairs' clamped values: bit 11 of a half <=> a sample outside the window
      auto q_pair16 = [&](unsigned kk) {        // two int16 samples of a full chunk
        const CntS2 d2 = __builtin_bit_cast(CntS2, kk) - __builtin_bit_cast(CntS2, cc);
        if constexpr (!KS) {
          iq1 = __builtin_amdgcn_sdot2(d2, sone2, iq1, false);
          iq2 = (unsigned)__builtin_amdgcn_sdot2(d2, d2, (int)iq2, false);
        }
        const unsigned uu = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(CntU2, d2) + __builtin_bit_cast(CntU2, hw2), __builtin_bit_cast(CntU2, w2)));
#if !NMOD_CW_OR3
        watch |= uu;
#endif
        const unsigned x0 = ((uu & 0xffffu) >> lgE << 4) + tbE, x1 = ((uu >> 16) >> lgE << 4) + tbE;
        unsigned a0, a1;
        asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a0) : "v"(uu), "v"(x0));
        asm("v_mad_u32_u16 %0, %1, 4, %2 op_sel:[1,0,0,0]" : "=v"(a1) : "v"(uu), "v"(x1));
        arrive(a0, 0x10000u, true); arrive(a1, 0x10000u, true);
        return uu;
      };
      auto tail_of_pair16 = [&](unsigned kk) {   // the rare path: which of the pair's samples lie outside the window
        const CntS2 d2 = __builtin_bit_cast(CntS2, kk) - __builtin_bit_cast(CntS2, cc);
        const unsigned uu = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(CntU2, d2) + __builtin_bit_cast(CntU2, hw2), __builtin_bit_cast(CntU2, w2)));
        tail_push((uu & 0x00000800u) != 0u, (int)(short)(kk & 0xffffu), 1u);
        tail_push(uu >= 0x08000000u, (int)(short)(kk >> 16), 1u);
      };
#pragma unroll 1
      for (int ch0 = 0; ch0 < full; ch0 += PF) {
        Q4Raw qb[PF];
#pragma unroll
        for (int j = 0; j < PF; ++j) qb[j] = rows.qa[j];
        if (ch0 + PF < full) {                             // (wave-uniform)
#pragma unroll
          for (int j = 0; j < PF; ++j) if (ch0 + PF + j < full) rows.qa[j] = load_at(rowq, (unsigned)((ch0 + PF + j) * 256 + 4 * lane) * (unsigned)sizeof(Q1Raw), Q4Raw());
        }
#pragma unroll
        for (int j = 0; j < PF; ++j) {
          if (ch0 + j < full) {                            // (wave-uniform)
            if constexpr (RDT == 0) {
              q_f32(qb[j].x, true); q_f32(qb[j].y, true); q_f32(qb[j].z, true); q_f32(qb[j].w, true);
            } else {
              const CntWU2 two = __builtin_bit_cast(CntWU2, qb[j]);
              const unsigned ua = q_pair16(two.x), ub = q_pair16(two.y);
#if NMOD_CW_OR3
              asm("v_or3_b32 %0, %1, %2, %3" : "=v"(watch) : "v"(watch), "v"(ua), "v"(ub));
#else
              (void)ua; (void)ub;
#endif
            }
          }
        }
        if constexpr (RDT == 1) {
          if (__ballot((watch & 0x08000800u) != 0u) != 0ull) {   // (wave-uniform) a sample of these chunks fell outside the window
#pragma unroll
            for (int j = 0; j < PF; ++j) {
              if (ch0 + j < full) {
                const CntWU2 two = __builtin_bit_cast(CntWU2, qb[j]);
                tail_of_pair16(two.x); tail_of_pair16(two.y);
              }
            }
            watch = 0u;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (j < tail) {                                    // (wave-uniform)
          const bool have = full * 256 + j * 64 + lane < q;
          const Q1Raw cur1 = rows.rt[j];
          if constexpr (RDT == 0) {
            q_f32(have ? (float)cur1 : xq0, have);
          } else {
            // (the sample and, as its pair, the centre: d = 0 there and only the low half is counted)
            const unsigned kk = have ? (((unsigned)(int)cur1 & 0xffffu) | (cc & 0xffff0000u)) : cc;
            const CntS2 d2 = __builtin_bit_cast(CntS2, kk) - __builtin_bit_cast(CntS2, cc);
            if constexpr (!KS) {
              iq1 = __builtin_amdgcn_sdot2(d2, sone2, iq1, false);
              iq2 = (unsigned)__builtin_amdgcn_sdot2(d2, d2, (int)iq2, false);
            }
            const unsigned uu = __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(CntU2, d2) + __builtin_bit_cast(CntU2, hw2), __builtin_bit_cast(CntU2, w2)));
            const unsigned x0 = ((uu & 0xffffu) >> lgE << 4) + tbE;
            unsigned a0;
            asm("v_mad_u32_u16 %0, %1, 4, %2" : "=v"(a0) : "v"(uu), "v"(x0));
            // (a sample that does not exist: a zero word of block 0's pad — the dump entry's arrival numbers are the tail samples')
            arrive(have ? a0 : tb, 0x10000u, have);
            tail_push(have && (uu & 0xffffu) == (unsigned)W, (int)cur1, 1u);
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    request(nxt, rows, true, PS);       // the next position's rows: in flight while this one's table is scanned
    if (fit) {
      // ---- prefix sums in place (both halves at once: the S half stays below 2^16); the pad's last word = the sum below the block
      unsigned tot = 0u;
#pragma unroll
      for (int i = 0; i < 8; ++i) if (i < nch) { const uint4 v = blk[i]; tot += (v.x + v.y) + (v.z + v.w); }
      unsigned below = seg_exscan_add_u32<64>(tot, lane);
      const unsigned all_in = __builtin_amdgcn_readlane(below + tot, 63);     // samples inside the window: S | Q << 16
      // ---- the tail: what is missing from the window's total sits in the list (more than it holds: the position goes to the sorting form)
      const int nt_s = m - (int)(all_in & 0xffffu), nt_q = q - (int)(all_in >> 16);
      nt = nt_s + nt_q; nt_q_pol = nt_q; nt_s_ = nt_s;
      fit = __ballot(bad) == 0ull && nt <= kCwTail && listed == nt;
      if (fit && nt > 0) {                                   // (wave-uniform) lane i < nt holds tail sample i
        const unsigned e = *(CntLdsU32)(uintptr_t)(tail_b + ((unsigned)min(lane, kCwTail - 1) << 2));
        t_val = lane < nt;
        t_key = (int)(short)(e & 0xffffu); t_grp = (e >> 16) & 1u;
        // every tail sample against every other: the tail samples at or below / strictly below it (S | Q << 16), its earlier copies.
        // What depends on sample j alone is scalar arithmetic on the broadcast key (the SALU runs beside the vector work): how many
        // lie below the window, and the samples' own moment terms about the window's centre — S's sums left its tail samples out,
        // Q's int16 stream saw theirs through 16-bit arithmetic, (k - c) mod 2^16: taken out again as that, put in as the true
        // distance (|d| < 2^16: d^2 < 2^32)
        unsigned low_cnt = 0u;
#pragma unroll 1
        for (int j = 0; j < nt; ++j) {
          const int kj = __builtin_amdgcn_readlane(t_key, j);
          const bool qj = __builtin_amdgcn_readlane((int)t_grp, j) != 0;
          const unsigned incj = qj ? 0x10000u : 1u;
          t_le += (kj <= t_key) ? incj : 0u;
          t_lt += (kj < t_key) ? incj : 0u;
          t_p += (kj == t_key && j < lane) ? 1u : 0u;
          low_cnt += kj < base ? incj : 0u;
          if constexpr (RDT == 1 && !KS) {
            const int d = kj - c, dw = (int)(short)d;
            const unsigned dd = (unsigned)d * (unsigned)d;
            if (!qj) { ts_d += d; ts_dd += (unsigned long long)dd; }
            else { tq_d += d - dw; tq_dw2 += (unsigned)(dw * dw); tq_dd += (unsigned long long)dd; }
          }
        }
        below += low_cnt;                                    // the samples below the window: every in-window count includes them
        // ... and everything inside or below the window for a sample above it
        const unsigned under = (t_val && t_key >= base) ? all_in : 0u;
        t_le += under; t_lt += under;
      }
      if (fit) {
        blk[-1] = make_uint4(0u, 0u, 0u, below);
        unsigned carry = below;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          if (2 * h < nch) {
            unsigned w[8];
#pragma unroll
            for (int i = 0; i < 2; ++i) { const uint4 v = blk[2 * h + i]; w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w; }
#pragma unroll
            for (int i = 0; i < 8; ++i) { carry += w[i]; w[i] = carry; }
#pragma unroll
            for (int i = 0; i < 2; ++i) blk[2 * h + i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- every sample of S: (A | B << 16) just below and at its value
        constexpr bool KEEP = RS <= 8;                   // the candidates stay in registers (RS = 16: recomputed from the table)
        int xs[KEEP ? 2 * RS : 2];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
          const bool have = in_window(r);
          const CntLdsU32 pw = (CntLdsU32)(uintptr_t)below_of(r, have);
          const unsigned w0 = pw[0], w1 = pw[1];
          if constexpr (!KS) {
            mws = __builtin_amdgcn_udot2(__builtin_bit_cast(CntU2, w0), hi1, mws, false);    // #{q < s} + #{q <= s}
            mws = __builtin_amdgcn_udot2(__builtin_bit_cast(CntU2, w1), hi1, mws, false);
          }
          const int x0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, w0), qm, 0, false);   // A nQ - B nS just below the value
          const int x1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, w1), qm, 0, false);   // ... at it
          if constexpr (KEEP) { xs[2 * r] = x0; xs[2 * r + 1] = x1; }
          vmax = max(vmax, max(x0, x1)); vmin = min(vmin, min(x0, x1));
        }
        // the tail samples of S: the same two candidates and MWU terms from the all-pairs counts (a word per candidate, A | B << 16)
        int tx0 = 0, tx1 = 0;
        if (nt_s_ > 0) {                                     // (wave-uniform)
          const bool ts = t_val && t_grp == 0u;
          const unsigned w0 = ts ? t_lt : 0u, w1 = ts ? t_le : 0u;
          if constexpr (!KS) {
            mws = __builtin_amdgcn_udot2(__builtin_bit_cast(CntU2, w0), hi1, mws, false);
            mws = __builtin_amdgcn_udot2(__builtin_bit_cast(CntU2, w1), hi1, mws, false);
          }
          tx0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, w0), qm, 0, false);
          tx1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, w1), qm, 0, false);
          vmax = max(vmax, max(tx0, tx1)); vmin = min(vmin, min(tx0, tx1));
        }
        best = cw_wave_max_u32((unsigned)max(vmax, -vmin));
        // D in the float form at the candidates that reach the maximum (few: their table words are read again)
        if (best != 0u && !(KS && args.ks_rational_d)) {
          const double dm_ = (double)m, dq_ = (double)q;
          const double rm_ = rcp_m, rq_ = rcp_q;
#pragma unroll
          for (int r = 0; r < RS; ++r) {
            int xr[2];
            if constexpr (KEEP) { xr[0] = xs[2 * r]; xr[1] = xs[2 * r + 1]; }
            else {
              const CntLdsU32 pw = (CntLdsU32)(uintptr_t)below_of(r, in_window(r));
              xr[0] = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, pw[0]), qm, 0, false);
              xr[1] = __builtin_amdgcn_sdot2(__builtin_bit_cast(CntS2, pw[1]), qm, 0, false);
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const bool hit = xr[e] == (int)best || xr[e] == -(int)best;
              if (__ballot(hit) != 0ull) {
                const unsigned w = ((CntLdsU32)(uintptr_t)below_of(r, in_window(r)))[e];
                const double d = fabs(hist_exact_quot((int)(w & 0xffffu), dm_, rm_) - hist_exact_quot((int)(w >> 16), dq_, rq_));
                dmax = hit ? fmax(dmax, d) : dmax;
              }
            }
          }
          if (nt_s_ > 0) {                                   // (wave-uniform) the tail samples of S that reach the maximum
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const int xe = e ? tx1 : tx0;
              const bool hit = t_val && t_grp == 0u && (xe == (int)best || xe == -(int)best);
              if (__ballot(hit) != 0ull) {
                const unsigned w = e ? t_le : t_lt;
                const double d = fabs(hist_exact_quot((int)(w & 0xffffu), dm_, rm_) - hist_exact_quot((int)(w >> 16), dq_, rq_));
                dmax = hit ? fmax(dmax, d) : dmax;
              }
            }
          }
        }
      }
    }

    if (fit) {                                             // (wave-uniform)
      dmax = wave_max_f64(dmax);
      if constexpr (KS) {
        const bool any_tie = DTYPE == 2 && __ballot(sp != 0u || (nt > 0 && t_val && t_p != 0u)) != 0ull;
        if (lane == 0) {
          args.ks_num[pos] = best;
          args.ks_d_ref[pos] = dmax;
          if (args.tied) args.tied[pos] = any_tie ? 1 : 0;
        }
      } else {
        const double dm = (double)m, dq = (double)q;
        const double rm = rcp_m, rq = rcp_q;
        const unsigned MWS = cw_wave_sum_u32(mws);           // <= 2 m q
        // sum_v t^3 - n = 3 sum (p^2 + p) over the arrivals: the table's, + the tail samples' earlier copies among the tail, - what
        // the dump entry handed the tail samples of an int16 Q as arrival numbers (0, 1, ..., nt_q - 1: sum of p^2 + p = (n-1) n (n+1) / 3)
        if (nt > 0) { const unsigned tp = t_val ? t_p : 0u; sp += tp; sp2 += tp * tp; }
        unsigned long long arrivals = wave_sum_u64((unsigned long long)(sp2 + sp));
        if constexpr (RDT == 1) arrivals -= (unsigned long long)(nt_q_pol > 0 ? nt_q_pol - 1 : 0) * (unsigned long long)nt_q_pol * (unsigned long long)(nt_q_pol + 1) / 3ull;
        const unsigned long long TIE = 3ull * arrivals;
        double mean_s = 0.0, m2_s = 0.0, mean_q = 0.0, m2_q = 0.0;
        if constexpr (DTYPE == 0) {
          const double s1 = wave_sum_f64(ms1), s2 = wave_sum_f64(ms2), t1 = wave_sum_f64(mq1), t2 = wave_sum_f64(mq2);
          mean_s = (double)xs0 + s1 * rm; m2_s = s2 - s1 * s1 * rm;
          mean_q = (double)xq0 + t1 * rq; m2_q = t2 - t1 * t1 * rq;
        } else if constexpr (DTYPE == 1) {
          // exact integer sums about the centre: |sum d| <= 4 095 * 1 024, sum d^2 <= 4 095 * 2^20 < 2^32
          double s1 = (double)(int)cw_wave_sum_u32((unsigned)is1), s2 = (double)cw_wave_sum_u32(is2);
          unsigned t1u = cw_wave_sum_u32((unsigned)iq1), t2u = cw_wave_sum_u32(iq2);
          double t2x = 0.0;
          if (nt > 0) {                                      // (wave-uniform) the tail samples' own terms, summed beside the all-pairs pass
            s1 += (double)ts_d; s2 += (double)ts_dd;
            t1u += (unsigned)tq_d; t2u -= tq_dw2;
            t2x = (double)tq_dd;
          }
          const double t1 = (double)(int)t1u, t2 = (double)t2u + t2x;
          mean_s = ((double)c + s1 * rm) * 1e-3; m2_s = __fma_rn(dm, s2, -s1 * s1) * rm * 1e-6;
          mean_q = ((double)c + t1 * rq) * 1e-3; m2_q = __fma_rn(dq, t2, -t1 * t1) * rq * 1e-6;
        }
        if (lane == 0) {
          args.ks_num[pos] = best;
          args.ks_d_ref[pos] = dmax;
          args.mwu_s[pos] = swap ? 2ull * (unsigned long long)m * (unsigned long long)q - (unsigned long long)MWS : (unsigned long long)MWS;
          args.tie[pos] = TIE;
          if constexpr (DTYPE != 2) {
            double* mo = args.moments + pos * 4;
            mo[swap ? 2 : 0] = mean_s; mo[swap ? 3 : 1] = m2_s; mo[swap ? 0 : 2] = mean_q; mo[swap ? 1 : 3] = m2_q;
          }
          if (args.tied) args.tied[pos] = TIE != 0ull ? 1 : 0;
        }
      }
    }
    if (!fit && cur.fit) {                                 // (wave-uniform) its sizes were this form's, its samples are not: left to the class's sorting form
      if (lane == 0) work_list[loff + atomicAdd(work_cnt, 1)] = (int)pos;
    }
  }
}

// every class the probe accepted, in one launch: a wave takes positions w, w + (waves), ... of the classes' lists laid end to end
// (no tail per class: 15 classes of a ragged batch in 15 launches left the chip half idle)
template <int DTYPE, bool KS>
__global__ __launch_bounds__(64 * kWavesPerBlock, 4)
void rank_count_wide_kernel(CntWideArgs cw) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds_cw[];
  const RankStatsArgs& args = cw.rs;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* tbl = lds_cw + wave * kCwWaveWords;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
  const int nseg = cw.segs[0];
  int64_t rot = 0;                                         // positions handed out so far, mod the number of waves
  for (int sg = 0; sg < nseg; ++sg) {
    const int cls = cw.segs[1 + sg];
    if (cw.gates[cls] == 0) continue;
    int64_t count = args.npos, loff = 0;
    const int32_t* list = nullptr;
    if (args.pos_list) { count = args.class_meta[cls]; loff = args.class_meta[kClassStride + cls]; list = args.pos_list + loff; }
    int64_t start = wave_global - rot;
    if (start < 0) start += nw;
    rot = (rot + count) % nw;
    switch (count_wide_rs_index(cls)) {
      case 0: cw_segment<DTYPE, 1, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane); break;
      case 1: cw_segment<DTYPE, 2, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane); break;
      case 2: cw_segment<DTYPE, 4, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane); break;
      case 3: cw_segment<DTYPE, 8, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane); break;
      case 4: cw_segment<DTYPE, 16, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane); break;
      default: break;                                      // (index 5 — both groups above 1 024 samples: rank_count_value_kernel, rank_count_value.hpp)
    }
  }
}

}  // namespace nmod
