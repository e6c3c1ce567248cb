// K1, all-tests form for positions whose two groups fall in the same capacity class (+-1)
// (tests mask = MWU | Welch | KS: what getKStest computes for every position, myDetect.py:327-343).
//
// Same skeleton as the KS-only kernel (ks_rank.hpp): the smaller group S (m samples) is sorted in registers and
// written to wave-private LDS, every sample x of the other group Q (q samples) finds L = #{s < x} by binary search
// and looks whether the key it lands on equals it, and one LDS atomic per sample builds the histograms
// cntL[j] = #{x : L(x) = j} and eq[j] = #{x : x = S[j], the first key of its run}.  With cumL the prefix sum of cntL
// everything the three tests need follows without sorting Q against S:
//   * KS          exact integer max |c0 n1 - c1 n0| over the pooled points from the two candidates per run end
//                 k of S, (cumL(k-1), k) and (cumU(k), k), cumU(k) = cumL(k) - eq[k]  (derivation: ks_rank.hpp).
//                 ks_2samp forms D as max |fl(c0/n0) - fl(c1/n1)|; a larger integer numerator always gives a larger
//                 float value (they differ by >= 1/(n0 n1) >> ulp), so the float form is evaluated ONLY for the
//                 candidates that reach the integer maximum — the maximum of those is the reference's D bit for bit.
//   * Mann-Whitney sum_{x in Q} (L(x) + U(x)), U = #{s <= x}:  sum L = sum_{j<C} (q - cumL(j))  [Abel summation] and
//                 U(x) - L(x) = the length a of the run of S that x ties with, so sum (L + U) = 2 sum L + sum_runs a b,
//                 b = eq[start of the run] = the samples of Q equal to it.  This is
//                 sum_{a in group 1} (#{b < a} + #{b <= a}) when Q is group 1, and 2mq minus it when Q is group 2.
//   * tie term    sum over pooled tie groups of t^3 - t = 3 pp(S) + 3 pp(Q) + 3 sum_{runs of S tied with Q} a b (a + b),
//                 pp(X) = sum over the elements of X of p (p - 1), p = place of the element in its run of equal keys.
//   * Welch       fp64 shifted one-pass moments of both groups on the way (S from the registers before the sort,
//                 Q as its samples stream through the ranking rounds).
// pp(S) comes from the sorted registers.  pp(Q) needs equal samples of Q next to each other, but not a second
// sort: S acts as the splitter set of a sample sort.  Samples with different L are already ordered, so
// Q is scattered to index cumL(L - 1) + (arrival number within its bin, returned by the histogram atomic) —
// into the LDS words of S's keys, which are dead by then — read back R consecutive keys per lane, and finished
// by as many odd-even transposition phases as the fullest bin holds samples (~1 sample per bin: 8-10 phases of
// 2 instructions per key against the 44 of a full network).
// Against round 1's rank_all_kernel (both groups sorted by one network, 2 positions per wave, a 72-instruction
// ranking step per sample of group 2 with four exact fp64 quotients): 4 positions per wave and 740 instead of 1003
// VALU instructions per 200 v 200 position.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ks_rank.hpp"
#include "rank_all.hpp"      // pad_run_pp
#include "rank_stats_launch.hpp"   // kNumSizeClasses

namespace nmod {

// sum over the lane's elements of p (p - 1), p = 1-based place of the element in its run of equal keys; y = the
// sorted keys of a group in the blocked layout (element gl * R + r in register r of lane gl), +inf pads at the end
// (not counted; y is clobbered)
template <int R, int LG>
__device__ __forceinline__ unsigned seg_tie_pp(float (&y)[R], int gl, int lane) {
  constexpr int N = R * LG;
  // LG == 8: two groups share one DPP row: the second group's scan values are biased by N, so whatever leaks in
  // from the first group (< N) can never win a max
  const int bias = (LG == 8 && (lane & 8)) ? N : 0;
  const float nanv = __builtin_nanf("");
  const float inf = __builtin_inff();
  float prev_last = lane_prev(y[R - 1], nanv);
  prev_last = (gl == 0) ? nanv : prev_last;
  // Usual case (continuous signals, and the occasional tie of 3-dp rounded ones): no key of the wave equals BOTH of its
  // two predecessors, i.e. every run of equal keys is a pair and adds p (p - 1) = 2: count the keys that equal their
  // predecessor.  The comparison results are wave masks in scalar registers; "both predecessors" is a scalar AND per
  // key (across a lane boundary: the mask of the lane's first key against the last key's mask moved up one lane).
  {
    unsigned long long eq_prev = 0ull, triple = 0ull, eq_first = 0ull;
    unsigned cnt = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const bool eq = (y[r] == ((r == 0) ? prev_last : y[r - 1])) && (y[r] != inf);   // (the +inf pads are not a tie)
      cnt += eq ? 1u : 0u;
      const unsigned long long e = __ballot(eq);
      if (r == 0) eq_first = e; else triple |= e & eq_prev;
      eq_prev = e;
    }
    triple |= eq_first & (eq_prev << 1);                 // key 0 of a lane, key R - 1 and key R - 2.. of the lane below:
    // (eq_prev is now the mask of key R - 1 == key R - 2; a first key that equals the previous lane's last key, which
    // itself equals its predecessor, closes a run of three; first keys of a group never compare equal: prev_last is NaN)
    if (triple == 0ull) return 2u * cnt;
  }
  // general case: pads become NaN (never equal to anything), then every key gets its place in its run
#pragma unroll
  for (int r = 0; r < R; ++r) y[r] = (y[r] != inf) ? y[r] : nanv;
  prev_last = lane_prev(y[R - 1], nanv);
  prev_last = (gl == 0) ? nanv : prev_last;
  int run = bias;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float p = (r == 0) ? prev_last : y[r - 1];
    run = (y[r] != p) ? (gl * R + r + bias) : run;
  }
  int carry = lane_prev_i(seg_scan_max_i32<LG>(run), 0);
  carry = (gl == 0) ? bias : carry;
#pragma unroll
  for (int r = 0; r < R; ++r) asm volatile("" : "+v"(y[r]));     // (keeps the first sweep's comparison masks out of SGPRs)
  run = carry;
  unsigned acc = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float p = (r == 0) ? prev_last : y[r - 1];
    const int e = gl * R + r + bias;
    run = (y[r] != p) ? e : run;
    const unsigned pm1 = (unsigned)(e - run);
    acc += __umul24(pm1, pm1) + pm1;
  }
  return acc;
}

// compare-exchange without the canonicalising v_max x, x that fminf / fmaxf put in front of values of unknown origin
// (here: keys read back from LDS and carried around a loop): the keys are ordinary numbers or +-inf
__device__ __forceinline__ void ce_raw(float& lo, float& hi) {
  float a, b;
  asm("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3" : "=&v"(a), "=v"(b) : "v"(lo), "v"(hi));
  lo = a; hi = b;
}

// Is the float32 sample x on the milli-unit grid of real events (myRefBaseSignalAnnotation.py:1108 rounds them to 3 decimals),
// i.e. x == RN32(k / 1000) for an integer |k| <= 32 767?  k = rint(1000 x); the quotient by Markstein's sequence with the
// correctly rounded reciprocal (q0 = k r, rem = fma(-q0, 1000, k), q = fma(rem, r, q0)): equal to the float32 division
// k / 1000.0f for every |k| <= 32 767 (exhaustive host test, tests/test_grid_key.py).  Equal keys <=> equal samples among
// the samples that pass: the integer keys order and tie exactly as the floats do.
// RANGE = false leaves the |k| <= 32 767 test out (one compare per sample of the streamed group): an accepted sample is a function
// of its key whatever the range (x == q(t)), so equal keys still mean equal samples; keys outside the counters' window wait on the
// tail list (compared as 32-bit keys) or, past its 64 entries, send the position to the recount, which checks the range once per
// position (a key beyond +-2^31 saturates and fails the test above unless it is that one float: no two samples share it).
template <bool RANGE = true>
__device__ __forceinline__ bool grid_key(float x, int& k) {
  const float t = __builtin_rintf(__fmul_rn(x, 1000.0f));
  const float r = 1.0e-3f;
  const float q0 = __fmul_rn(t, r);
  const float rem = __fmaf_rn(-q0, 1000.0f, t);
  const float q = __fmaf_rn(rem, r, q0);
  k = (int)t;                                          // (v_cvt_i32_f32 saturates; a value off the grid is never used as a key)
  if constexpr (RANGE) return q == x && __builtin_fabsf(t) <= 32767.0f;
  else return q == x;
}

// `phases` odd-even transposition phases over the R x LG keys of every group of the wave (blocked layout): enough to
// sort a sequence whose elements are at most phases - 1 places from home
template <int R, int LG>
__device__ __forceinline__ void seg_oddeven_phases(float (&y)[R], int gl, int phases) {
  const float inf = __builtin_inff();
#pragma unroll 1
  for (int ph = 0; ph < phases; ph += 2) {
#pragma unroll
    for (int r = 0; r + 1 < R; r += 2) ce_raw(y[r], y[r + 1]);
#pragma unroll
    for (int r = 1; r + 1 < R; r += 2) ce_raw(y[r], y[r + 1]);
    float nx, pv;                                       // key 0 of the next lane / key R - 1 of the previous one
    if constexpr (LG <= 16) {                           // (a group lies inside one DPP row)
      nx = dpp_f<kDppRowShl + 1, 0xf, 0xf, true>(0.0f, y[0]);
      pv = dpp_f<kDppRowShr + 1, 0xf, 0xf, true>(0.0f, y[R - 1]);
    } else {
      nx = lane_next(y[0], inf);
      pv = lane_prev(y[R - 1], -inf);
    }
    nx = (gl == LG - 1) ? inf : nx;                     // (group boundaries)
    pv = (gl == 0) ? -inf : pv;
    float lo, hi;
    asm("v_max_f32 %0, %2, %3\n\tv_min_f32 %1, %4, %5" : "=&v"(lo), "=&v"(hi) : "v"(y[0]), "v"(pv), "v"(y[R - 1]), "v"(nx));
    y[0] = lo; y[R - 1] = hi;
  }
}

template <int LG>
__device__ __forceinline__ unsigned pos_allsum_u32(unsigned v) {     // sum over the LG lanes of a group, in every lane
  v += (unsigned)dpp_i<NMOD_QP(1, 0, 3, 2), 0xf, 0xf, true>(0, (int)v);
  v += (unsigned)dpp_i<NMOD_QP(2, 3, 0, 1), 0xf, 0xf, true>(0, (int)v);
  v += (unsigned)dpp_i<kDppRowHalfMirror, 0xf, 0xf, true>(0, (int)v);
  if constexpr (LG >= 16) v += (unsigned)dpp_i<kDppRowMirror, 0xf, 0xf, true>(0, (int)v);
  if constexpr (LG >= 32) v += (unsigned)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);
  if constexpr (LG == 64) v = (unsigned)__builtin_amdgcn_readlane((int)v, 0) + (unsigned)__builtin_amdgcn_readlane((int)v, 32);
  return v;
}

#ifndef NMOD_SKIP
#define NMOD_SKIP 0
#endif
#ifndef NMOD_HIST_WAVES
#define NMOD_HIST_WAVES 4
#endif
// WIDE (LG = 64, R = 1, 2 or 4; one position per wave): positions whose groups fall in DIFFERENT capacity classes, the
// smaller one S of at most 256 samples, the other Q of up to 4 096 (config 5: ~1000 v ~50 reads).  S is sorted and
// ranked into exactly as above; Q streams through the ranking rounds in a loop (nothing of it is kept), and the ties
// inside Q — the scatter + clean-up needs Q to fit the words of S — are counted in a per-wave table in LDS behind the
// bins (wide_table_words: 2 048 words = 8 KB per wave for every class, four blocks per CU).  int16, and float32 positions on
// the milli-unit grid of real events (grid_key): one 8-bit counter per VALUE of a window of 8 192 milli-units, a returning add per
// sample gives its place p in its run.  float32 otherwise: two bitmaps — a sample sets its bit in the first and, when the bit was
// already set, marks it in the second; a second pass over Q sends only the samples on marked bits through an exact multiset
// table (open addressing: an arrival passes all earlier copies of its key and adds p (p - 1)); a position with more of them than
// that table takes goes on a redo list for wide_redo_kernel (big_rank.hpp).  See the streaming section below.
constexpr unsigned kWideEmpty = 0xffffffffu;
constexpr int kWideList = 128;                     // words of the list in front of the exact table: < 64 waiting + <= 64 of one flush
constexpr int kWideTail = 64;                      // counters: samples of Q outside the window wait here for an all-pairs tie count (round 6)
#ifndef NMOD_WIDE_TAILS
#define NMOD_WIDE_TAILS 1                          // 0: round 5's form — any sample outside the window sends the position to recount16
#endif

// AFTER (R = LG = 16 only): the launch follows rank_count_kernel (rank_count.hpp) over the same work list.  When the probe's
// gate is set, that kernel has left one flag byte per position, four per work item (cnt_done as dwords): a wave reads the
// dwords of its next 64 items with one load and walks only the items that still hold a position (a 64-bit mask), and
// inside such an item only those positions are valid.  Gate clear: every item, as without AFTER.
template <int R, int LG, int DTYPE, bool WIDE = false, bool AFTER = false>
__global__ __launch_bounds__(64 * kWavesPerBlock, (WIDE ? 2 : (R <= 16 ? NMOD_HIST_WAVES : 2)))
void rank_hist_kernel(RankStatsArgs args) {
  static_assert(!AFTER || (!WIDE && 64 / LG == 4), "the counting form works on items of four positions");
  // around the counting form two instances are launched and the probe's gate picks one: continuous rows run the plain instance
  // at its own register budget, event-like rows the AFTER instance over what rank_count_kernel left
  if (args.cnt_mode != 0 && (args.cnt_gate[0] != 0) != (args.cnt_mode == 2)) return;
  static_assert(WIDE ? (LG == 64 && (R == 1 || R == 2 || R == 4)) : (LG == 8 || LG == 16 || LG == 32 || LG == 64), "lanes per sorted group");
  static_assert(WIDE || (R >= 8 && R <= 32 && (R & (R - 1)) == 0), "registers per lane");
  static_assert(R * LG <= 1024, "32-bit tie sums and 15-bit counts need sorted groups of at most 1024 samples");   // (WIDE: Q <= 4096 < 2^15)
  constexpr int PW = 64 / LG;                  // positions per wave
  constexpr int C = R * LG;
  constexpr int LOG_R = (R == 1) ? 0 : (R == 2) ? 1 : (R == 4) ? 2 : (R == 8) ? 3 : (R == 16) ? 4 : 5;
  using Lay = KsLayout<R, LG>;
  constexpr int ROW = Lay::ROW;
  constexpr int BIN_WORDS = WIDE ? ((ks_rank_pos_words(R, LG) + 3) & ~3) : ks_rank_pos_words(R, LG);   // keys + bins of a position
  const int wslots = WIDE ? wide_table_words(args.class_id, DTYPE) : 0;           // words of the wave's tie table
  const int POS_WORDS = BIN_WORDS + wslots + ((WIDE && DTYPE == 0) ? kWideList : 0);   // WIDE: the table (float32: and the deferred list) behind them, 16-byte aligned
  constexpr int HIST_OFF = Lay::REGION;        // words from key 0 to bin 0
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);
  const int slot = lane / LG;
  float* keys = lds_all + (wave * PW + slot) * POS_WORDS;        // sorted S of this lane's position (KsLayout)
  unsigned* hist = reinterpret_cast<unsigned*>(keys + HIST_OFF);    // bin j: cntL[j] << 16 | eq[j]; later the prefix table

  const float inf = __builtin_inff();
  const float big = 3.4028234663852886e38f;
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  for (int r = gl; r < R; r += LG) keys[r * ROW + Lay::END] = inf;   // the spare column: key C (and beyond) = +inf

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

  // AFTER: the items this wave still has to do among its next 64 (item `win_base + j * wave_stride` <-> bit j / lane j's dword)
  bool after_count = false;
  int64_t win_base = wave_global;
  unsigned long long win_todo = 0ull;
  unsigned win_flags = 0u;                       // lane j: the four flag bytes of item win_base + j * wave_stride (1 = done)
  unsigned cur_flags = 0u;                       // ... of the item being described
  if constexpr (AFTER) after_count = true;       // (cnt_mode = 2: this instance runs only behind rank_count_kernel)
  [[maybe_unused]] auto load_window = [&]() {
    const int64_t mine = win_base + (int64_t)(threadIdx.x & 63) * wave_stride;
    win_flags = 0x01010101u;
    if (mine < items) win_flags = after_count ? reinterpret_cast<const unsigned*>(args.cnt_done)[mine] : 0u;
    win_todo = __ballot(win_flags != 0x01010101u);
  };
  // the first item at or after the window's start that still holds work (>= items: none)
  [[maybe_unused]] auto next_in_windows = [&]() -> int64_t {
    while (win_todo == 0ull) {
      win_base += 64 * wave_stride;
      if (win_base >= items) return items;
      load_window();
    }
    const int j = __ffsll((long long)win_todo) - 1;
    win_todo &= win_todo - 1ull;
    cur_flags = (unsigned)__builtin_amdgcn_readlane((int)win_flags, j);
    return win_base + (int64_t)j * wave_stride;
  };

  // One work item = the PW positions of a wave.  S = the smaller group (ties: group 1).
  struct Item { bool valid, swap; int m, q; int64_t pos, off_s, off_q; };
  auto describe = [&](int64_t it) {
    Item d;
    const int64_t li = it * PW + slot;
    d.valid = it < items && li < count;
    if constexpr (AFTER) d.valid = d.valid && ((cur_flags >> (8 * slot)) & 0xffu) == 0u;
    d.pos = d.valid ? (list ? (int64_t)list[li] : li) : 0;
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
  using Q4Raw = typename std::conditional<DTYPE == 0, KsF4, KsS4>::type;
  using Q1Raw = typename std::conditional<DTYPE == 0, float, int16_t>::type;
  // unconditional loads (see ks_rank_kernel): lanes without samples read a block of FLT_MAX
  auto load_q4 = [&](const void* sig, int64_t off, int idx, bool have) -> Q4Raw {
    const Q1Raw* src = have ? reinterpret_cast<const Q1Raw*>(sig) + off + idx : reinterpret_cast<const Q1Raw*>(kKsBig4);
    return ks_global_load<Q4Raw>(src);
  };
  auto load_q1 = [&](const void* sig, int64_t off, int idx, bool have) -> Q1Raw {
    return ks_global_load<Q1Raw>(have ? reinterpret_cast<const Q1Raw*>(sig) + off + idx : reinterpret_cast<const Q1Raw*>(kKsBig4));
  };

  // fixed-stride batches: every position has the same sizes; fl(1/m) and fl(1/q) are taken once per block and parked
  // in LDS behind the positions' words (four vector registers for the whole kernel otherwise)
  const bool uniform = args.stride0 > 0 && args.stride1 > 0;
  double* recip = reinterpret_cast<double*>(lds_all + kWavesPerBlock * PW * POS_WORDS);
  if (uniform && threadIdx.x == 0) {
    const int64_t a0 = args.stride0 < args.stride1 ? args.stride0 : args.stride1;
    const int64_t a1 = args.stride0 < args.stride1 ? args.stride1 : args.stride0;
    recip[0] = 1.0 / (double)a0; recip[1] = 1.0 / (double)a1;
  }
  __syncthreads();

  int64_t it_first = wave_global;
  if constexpr (AFTER) {
    if (win_base < items) { load_window(); it_first = next_in_windows(); } else it_first = items;
  }
  Item cur = describe(it_first);
  constexpr bool PACKED = !WIDE && ks_packed_sort(R, LG, DTYPE);           // int16 rows sorted two keys per register
  float x[R];
  unsigned pk[8];                                  // PACKED: the item's S rows as packed int16 keys (x is filled by the sort)
  if constexpr (WIDE) {
    load_group<R, DTYPE>(x, cur.swap ? args.sig1 : args.sig0, cur.off_s, cur.m, lane);
  } else {
    KsRows<R, LG, DTYPE> first;
    first.request(cur.swap ? args.sig1 : args.sig0, cur.off_s, cur.m, gl);
    if constexpr (PACKED) first.finish_packed(pk, cur.m, gl);
    else first.finish(x, cur.m, gl);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)

  const int lane_k = lane;
  int64_t it_next = 0;
  for (int64_t it = it_first; it < items; it = it_next) {
    // everything derived from the lane number is re-derived per item from an opaque copy: hoisted out of the loop
    // the index constants of the unrolled sweeps (gl * R + r, ...) would occupy ~40 registers for the whole kernel
    int lane = lane_k;
    asm volatile("" : "+v"(lane));
    const int gl = lane & (LG - 1);
    const int seg_base = lane & ~(LG - 1);
    const bool valid = cur.valid;
    const int64_t pos = cur.pos;
    const int m = cur.m, q = cur.q;
    const bool swap = cur.swap;
    const void* sig_q = swap ? args.sig0 : args.sig1;
    const int64_t off_q = cur.off_q;

    // ---- ranking schedule: full rounds of one 16-byte load per lane, then the remaining < 4*LG samples one per lane
    const int full = q / (4 * LG);
    const int tail = (q - full * (4 * LG) + LG - 1) / LG;
    int full_w = 0, tail_w = 0;
#pragma unroll
    for (int s = 0; s < PW; ++s) {
      full_w = max(full_w, __builtin_amdgcn_readlane(full, s * LG));
      tail_w = max(tail_w, __builtin_amdgcn_readlane(tail, s * LG));
    }
    // requests that the sort hides: this item's first rounds of Q and its first sample
    Q4Raw ra = load_q4(sig_q, off_q, 4 * gl, 0 < full);
    Q1Raw rt = load_q1(sig_q, off_q, full * (4 * LG) + gl, full * (4 * LG) + gl < q);
    const Q1Raw rk = load_q1(sig_q, off_q, 0, q > 0);                       // the shift of Q's moments

    // ---- S: moments, sort, keys to LDS, ties inside S
#if !(NMOD_SKIP & 4)
    // WIDE: fl(1/m) and fl(1/q) once per position, for both groups' moments and the float form of D (this form has the
    // registers to keep them; the packed form re-derives them where needed)
    double rm_w = 0.0, rq_w = 0.0;
    if constexpr (WIDE) {
      if (uniform) { rm_w = recip[0]; rq_w = recip[1]; }
      else { rm_w = 1.0 / (double)m; rq_w = 1.0 / (double)q; }
    }
    {
      double mean, m2;
      const double rcp_m = WIDE ? rm_w : (uniform ? recip[0] : 0.0);   // fl(1/m) of a fixed-stride batch (parked in LDS)
      if constexpr (PACKED) seg_moments_packed16<LG>(pk, m, gl, mean, m2, rcp_m);
      else seg_moments<R, LG, DTYPE>(x, m, mean, m2, rcp_m);
      if (valid && gl == 0) {
        double* mo = args.moments + pos * 4 + (swap ? 2 : 0);
        mo[0] = mean; mo[1] = m2;
      }
    }
#endif
    // WIDE, float32: are S's samples on the milli-unit grid?  Then Q's ties are counted by value (direct-address counters, as
    // for int16 input) instead of through the multiset hash, as long as Q's samples are on the grid too (checked as they stream)
    bool s_grid = false;
    if constexpr (WIDE && DTYPE == 0) {
      bool okl = true;
#pragma unroll
      for (int r = 0; r < R; ++r) { int k; const bool ok = grid_key(x[r], k); okl = okl && (ok || r * 64 + lane >= m); }
      s_grid = __ballot(!okl) == 0ull && m > 0;
#if defined(NMOD_NO_GRID)
      s_grid = false;
#endif
    }
#if !(NMOD_SKIP & 16)
    if constexpr (PACKED) {
      // (packed_sort_i16.hpp; the top C - m keys of the position are its pads: a sample may equal the pad value 32767,
      // only the key index tells them apart)
      seg_sort_packed16<LG>(pk, lane);
      unpack_sorted16<LG>(pk, x);
      const int real = m - gl * R;
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = (r < real) ? x[r] : inf;
    } else {
      seg_sort_any<R, LG>(x, sel, lane);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) keys[r * ROW + gl] = x[r];
#endif
#pragma unroll
    for (int r = 0; r < R; ++r) hist[r * ROW + gl] = 0u;
    if (gl == LG - 1) hist[Lay::END] = 0u;                               // bin C
#if (NMOD_SKIP & 8)
    unsigned pp = 0;
#else
    unsigned pp = seg_tie_pp<R, LG>(x, gl, lane);
#endif
    asm volatile("" : "+v"(pp));                  // (done here, while S is in registers: not sunk below the ranking rounds)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- rank every Q sample into S and count it in the histograms; keep the sample and the LDS byte offset of its
    // L-bin inside the position's words (16 bits, two per register) for the scatter
    constexpr int NXS = WIDE ? 1 : R, NLA = WIDE ? 1 : R / 2;
    float xs[NXS];                                 // the samples this lane ranked (FLT_MAX where it had none)
    unsigned la[NLA];                              // byte offsets of their L-bins from `keys`
#pragma unroll
    for (int r = 0; r < NXS; ++r) xs[r] = big;
#pragma unroll
    for (int r = 0; r < NLA; ++r) la[r] = 0u;

    // rank NV samples: all the searches first (independent chains of LDS reads that the scheduler interleaves), then
    // the histogram updates: bin L(x) += 1 << 16, + 1 when x equals the key it landed on (ks_rank.hpp) — no second
    // search and no branch on ties
#ifndef NMOD_WIDE_TOPS
#define NMOD_WIDE_TOPS 0
#endif
    // WIDE: one position per wave, so S is wave-uniform: the probes of the first two search levels as scalars (ks_rank.hpp)
    [[maybe_unused]] KsTops tops;
    if constexpr (WIDE && NMOD_WIDE_TOPS) tops = ks_tops<R, LG>(keys);
    auto rank_many = [&](auto nv_tag, const float* xq, bool have, unsigned* ad) {
      constexpr int NV = decltype(nv_tag)::value;
      const float* lp[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        if constexpr (WIDE && NMOD_WIDE_TOPS) lp[e] = ks_search_tops<R, LG>(keys, xq[e], tops);
        else lp[e] = ks_search<R, LG, false, true>(keys, xq[e]);
      }
      unsigned inc[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) inc[e] = (*lp[e] == xq[e]) ? 0x10001u : 0x10000u;
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        unsigned* bin = reinterpret_cast<unsigned*>(const_cast<float*>(lp[e])) + HIST_OFF;
#if !(NMOD_SKIP & 1024)
        if (have) atomicAdd(bin, inc[e]);
#else
        asm volatile("" :: "v"(bin), "v"(inc[e]));
        if (have) atomicAdd(hist + (threadIdx.x & 63), inc[e]);          // (timing experiment: a conflict-free address)
#endif
        ad[e] = (unsigned)(uintptr_t)bin - (unsigned)(uintptr_t)keys;      // byte offset inside the position's LDS (< 64 KB)
      }
    };

    unsigned ppq = 0;                              // WIDE: ties inside Q (counters / exact table of the bitmap form)
    bool redo_flag = false;                        // WIDE float32: the position went on the redo list (wide_redo_kernel adds Q's ties)
    double s1w = 0.0, s2w = 0.0;                   // WIDE: Q's shifted moment sums
    // WIDE: the direct-address tie counters (int16 input, and float32 input on the milli-unit grid).  The wave's table holds
    // one 8-bit counter per VALUE of a window of the milli-unit domain (four per 32-bit word): one returning LDS add per
    // sample gives the number of earlier copies of its value, p - 1 — no hashing, no walks, one round trip.
    unsigned long long redo = 0ull;                                          // lanes that saw a counter at 255 (round 5: or a sample outside)
    // Round 6: a sample of Q OUTSIDE the window — a mis-segmented read anywhere in the +-5 unit clip range
    // (myRefBaseSignalAnnotation.py:251-259) — no longer voids the counts.  It cannot tie with a counted value, so it waits on a
    // list of kWideTail words (ballot + mbcnt compaction; int16: the table's last 64 words, the window is 7 936 values; float32
    // on the grid: the deferred list's words, unused on this path) and tail_ties() counts the ties among the listed samples by
    // all pairs after the pass.  More than kWideTail of them: recount16 as before.
    int wtails = 0;                                                          // (wave-uniform) samples put on the tail list
    const int wcount = WIDE ? wslots - ((NMOD_WIDE_TAILS && DTYPE == 1) ? kWideTail : 0) : 0;   // words of 8-bit counters
    [[maybe_unused]] unsigned* const wtail = reinterpret_cast<unsigned*>(keys) + BIN_WORDS + wcount;   // (float32: wcount = wslots, the deferred list)
    [[maybe_unused]] auto count_many = [&](auto cb_tag, auto nv_tag, const int* iv, const bool* have, int base) {
      constexpr int BITS = decltype(cb_tag)::value, NV = decltype(nv_tag)::value;
      constexpr int PW_LOG = (BITS == 16) ? 1 : 2;
      unsigned* ht = reinterpret_cast<unsigned*>(keys) + BIN_WORDS;
      unsigned old[NV], sh[NV];
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const unsigned u = (unsigned)(iv[e] - base);
        const bool in = u < ((unsigned)((BITS == 8) ? wcount : wslots) << PW_LOG);
        sh[e] = (u & ((1u << PW_LOG) - 1u)) * (unsigned)BITS;
        old[e] = 0u;
#if !(NMOD_SKIP & 128)
        if (have[e] && in) old[e] = atomicAdd(&ht[u >> PW_LOG], 1u << sh[e]);
#endif
        if constexpr (BITS == 8) {
#if NMOD_WIDE_TAILS
          const unsigned long long mk = __ballot(have[e] && !in);
          if (mk != 0ull) {                                                  // (wave-uniform; never on rows without outliers)
            const int idx = wtails + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            if (have[e] && !in && idx < kWideTail) wtail[idx] = (unsigned)iv[e];
            wtails += (int)__popcll(mk);
          }
#else
          redo |= __ballot(have[e] && !in);
#endif
        }
      }
#pragma unroll
      for (int e = 0; e < NV; ++e) {
        const unsigned c = (old[e] >> sh[e]) & ((1u << BITS) - 1u);        // earlier copies of the value: p - 1
        ppq += __umul24(c, c) + c;                                          // p (p - 1)
        if constexpr (BITS == 8) redo |= __ballot(c == 255u);               // the add wrapped the counter into its neighbour
      }
    };
    [[maybe_unused]] auto clear_table = [&]() {
      unsigned* ht = reinterpret_cast<unsigned*>(keys) + BIN_WORDS;
      __builtin_amdgcn_wave_barrier();
      for (int i = lane; i < wslots / 4; i += 64) reinterpret_cast<uint4*>(ht)[i] = make_uint4(0u, 0u, 0u, 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    };
    // the ties among the listed samples: lane i holds sample i and meets every earlier one — c copies before it add c (c + 1)
    [[maybe_unused]] auto tail_ties = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const bool tv = lane < wtails;
      const int v = tv ? (int)wtail[lane] : 0;
      unsigned c = 0u;
#pragma unroll 1
      for (int j = 0; j + 1 < wtails; ++j) {
        const int vj = __builtin_amdgcn_readlane(v, j);
        c += (tv && j < lane && vj == v) ? 1u : 0u;
      }
      ppq += __umul24(c, c) + c;
    };
    // The exceptions — more than kWideTail samples outside the window (groups far apart, a range beyond 8 units), or 256 samples of one
    // value (a constant stretch of signal): the counts are void.  The ties of the position are counted again the plain
    // way: min and max of Q, then one pass per window of [min, max] with 16-bit counters, one sample per lane and trip.
    // key_at(i, have): the integer key of sample i of Q.
    [[maybe_unused]] auto recount16 = [&](auto key_at) -> bool {
      ppq = 0u;
      int lo = 0x7fffffff, hi = (int)0x80000000;
#pragma unroll 1
      for (int i0 = 0; i0 < q; i0 += 64) {
        const int v = key_at(min(i0 + lane, q - 1), true);                   // (a sample read twice changes neither)
        lo = min(lo, v); hi = max(hi, v);
      }
      // (int16 input: keys within +-32 768.  float32 on the grid: any int, saturated beyond — such a range is refused)
      const unsigned bmax = wave_max_u32((unsigned)hi ^ 0x80000000u), bmin = ~wave_max_u32(~((unsigned)lo ^ 0x80000000u));
      if (bmax - bmin > 65535u) return false;
      const int vmax = (int)(bmax ^ 0x80000000u), vmin = (int)(bmin ^ 0x80000000u);
      const int np16 = (vmax - vmin) / (2 * wslots) + 1;                      // two 16-bit counters per word
#pragma unroll 1
      for (int pass = 0; pass < np16; ++pass) {
        const int base = vmin + pass * (2 * wslots);
        clear_table();
#pragma unroll 1
        for (int i0 = 0; i0 < q; i0 += 64) {
          const bool hv[1] = {i0 + lane < q};
          const int iv1[1] = {key_at(i0 + lane, hv[0])};
          count_many(std::integral_constant<int, 16>{}, std::integral_constant<int, 1>{}, iv1, hv, base);
        }
      }
      return true;
    };
    if constexpr (WIDE && DTYPE == 1) {
      // int16 samples.  The window is centred on the median of S (the two groups are reads of one position: real events
      // spread a few hundred milli-units around their level).  Samples of Q outside the window: the tail list (count_many).  A position
      // with more than kWideTail of them, or with a value that occurs 256 times (the counter wraps into its neighbour), is counted
      // again after the pass (recount16).
      const int wb = (int)keys[Lay::word(m > 0 ? (m >> 1) : 0)] - 2 * wcount;   // four values per word (m = 0: key 0 is the +inf pad -> any window)
      const int kq = (q > 0) ? (int)rk : 0;                                 // shift of Q's moment sums
      int s1i = 0; long long s2i = 0;                                        // sum (x - kq), sum (x - kq)^2: exact integers
      clear_table();
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
      for (int c = 0; c < full_w; ++c) {
        const Q4Raw rb = load_q4(sig_q, off_q, (c + 1) * (4 * LG) + 4 * gl, c + 1 < full);
        const int iv[4] = {(int)ra.x, (int)ra.y, (int)ra.z, (int)ra.w};
        const bool hv[4] = {true, true, true, true};
        float xa[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) xa[e] = (float)iv[e];
        unsigned ad[4];
#if !(NMOD_SKIP & 512)
        rank_many(std::integral_constant<int, 4>{}, xa, true, ad);
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int d = iv[e] - kq; s1i += d; s2i += (long long)d * (long long)d; }
        count_many(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}, iv, hv, wb);
        ra = rb;
      }
#pragma unroll 1
      for (int c = 0; c < tail_w; ++c) {
        const int idx_now = full * (4 * LG) + c * LG + gl;
        const bool have = idx_now < q;
        const int iv1[1] = {(int)rt};
        const int idx = full * (4 * LG) + (c + 1) * LG + gl;
        rt = load_q1(sig_q, off_q, idx, idx < q);
        const bool hv[1] = {have};
        const float xq1[1] = {have ? (float)iv1[0] : big};
        unsigned a1[1];
#if !(NMOD_SKIP & 512)
        rank_many(std::integral_constant<int, 1>{}, xq1, have, a1);
#endif
        const int d = have ? iv1[0] - kq : 0;
        s1i += d; s2i += (long long)d * (long long)d;
        count_many(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, iv1, hv, wb);
      }
      if (redo != 0ull || wtails > kWideTail) recount16([&](int i, bool have) { return (int)load_q1(sig_q, off_q, i, have); });
      else if (wtails > 1) tail_ties();
      // exact sums to doubles (|s1| < 2^23 and s2 < 2^39 per lane): the common moments code below reduces them
      s1w = (double)s1i; s2w = (double)s2i;
    } else if constexpr (WIDE) {
      unsigned* ht = reinterpret_cast<unsigned*>(keys) + BIN_WORDS;
      const float kqf = (q > 0) ? (float)rk : 0.0f;
      const double KQ = (double)kqf;
      // The exact multiset table of the bitmap form (below): a sample walks its value's double-hashing sequence (start and odd
      // step from the value alone, so every copy of a value walks the same slots) past all earlier copies to the first empty
      // slot; `dup` = the copies it passed, it is the (dup + 1)-th of its value: p (p - 1) = dup (dup + 1).  Only the few samples
      // on shared bitmap bits come here, from a list in LDS, 64 at a time, one per lane.  (Until round 4 EVERY sample of Q went
      // through such a table, two compare-and-swaps each: half of this form's time.)
      constexpr int W1 = 1024;                       // words of B1 (the table's words in pass 2) and of B2 behind it: 32 768 bits each
      constexpr unsigned SH = 17u;                   // bit index = the top 15 bits of the multiplicative hash
      constexpr unsigned tslots = (unsigned)wide_table_slots(W1);     // 1 021, a prime: any step visits every slot
      static_assert(2 * W1 <= wide_table_words(0, 0) && wide_table_words(0, 0) * 4 >= 8192, "two bitmaps; a counter window of 8 192 values");
      unsigned* B2 = ht + W1;
      unsigned* lst = ht + wslots;                   // the samples on shared bits that wait for their walk (kWideList words)
      int lcnt = 0;                                  // how many (the same in every lane)
      auto walk_to_end = [&](unsigned bits, bool act) {
        const unsigned hsh = bits * 2654435761u;
        unsigned hh = ((hsh >> 13) * tslots) >> 19;         // 19 hash bits x 1 021 slots: below 2^32
        const unsigned st = 1u + (hsh & 511u);
        unsigned dup = 0u;
        while (__ballot(act) != 0ull) {
          const unsigned old = atomicCAS(&ht[hh], kWideEmpty, act ? bits : kWideEmpty);   // (idle lanes: empty -> empty)
          dup += (act && old == bits) ? 1u : 0u;
          act = act && old != kWideEmpty;
          hh += st;
          hh = min(hh, hh - tslots);                          // (below tslots the difference wraps to a huge value)
        }
        ppq += dup * (dup + 1u);
      };
      auto drain = [&](int n) {                      // the last n <= 64 entries of the list
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        lcnt -= n;
        const bool act = lane < n;
        const unsigned bits = lst[lcnt + (act ? lane : 0)];
        walk_to_end(bits, act);
      };
      // ---- Samples off the milli-unit grid (continuous signals): ties inside Q are rare.  Pass 1 (with the ranking and the
      // moments when `rank` is set): every sample sets one bit of a bitmap B1 by a returning OR; a sample that finds its bit set
      // marks the same bit in a second bitmap B2 — the bits that two or more samples share.  Equal samples share a bit, so every
      // tied sample of Q sits on a B2 bit.  Pass 2 streams Q again (from L2), reads each sample's B2 bit, and only the samples on
      // marked bits (~n^2 / bits: a few per cent) walk the exact table, which takes B1's words.  No B2 bit: no ties inside Q.
      // More marked samples than that table takes at load 0.75 (thousands of equal samples off the grid): the position goes on
      // the redo list and wide_redo_kernel counts its ties by sorting Q; the kernel here adds nothing for them.
      auto bitmap_passes = [&](bool rank) {
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < (2 * W1) / 4; i += 64) reinterpret_cast<uint4*>(ht)[i] = make_uint4(0u, 0u, 0u, 0u);
        if (!rank) {                                 // (Q streams again: the requests of the item's head have been consumed)
          ra = load_q4(sig_q, off_q, 4 * gl, 0 < full);
          rt = load_q1(sig_q, off_q, full * (4 * LG) + gl, full * (4 * LG) + gl < q);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0x0F70);
        unsigned hits = 0u;                                        // samples of this lane that found their bit set
        // (the NV returning ORs of a round are issued together, then the marks: one LDS round trip per round, not one per sample)
        auto mark_many = [&](auto nv_tag, const float* xv, const bool* have) {
          constexpr int NV = decltype(nv_tag)::value;
          unsigned idx[NV], bit[NV], old[NV];
#pragma unroll
          for (int e = 0; e < NV; ++e) {
            const unsigned hsh = __float_as_uint(xv[e] + 0.0f) * 2654435761u;   // (-0.0 -> +0.0: one key per value)
            idx[e] = hsh >> SH;
            bit[e] = 1u << (idx[e] & 31u);
          }
#if !(NMOD_SKIP & 4096)
#pragma unroll
          for (int e = 0; e < NV; ++e) { old[e] = 0u; if (have[e]) old[e] = atomicOr(&ht[idx[e] >> 5], bit[e]); }
#else
#pragma unroll
          for (int e = 0; e < NV; ++e) old[e] = 0u;               // (timing experiment: no bitmap atomics)
#endif
          bool hit[NV], any = false;
#pragma unroll
          for (int e = 0; e < NV; ++e) { hit[e] = (old[e] & bit[e]) != 0u; any = any || hit[e]; hits += hit[e] ? 1u : 0u; }
          if (__ballot(any) != 0ull) {
#pragma unroll
            for (int e = 0; e < NV; ++e) if (hit[e]) atomicOr(&B2[idx[e] >> 5], bit[e]);
          }
        };
#pragma unroll 1
        for (int c = 0; c < full_w; ++c) {
          const Q4Raw rb = load_q4(sig_q, off_q, (c + 1) * (4 * LG) + 4 * gl, c + 1 < full);
          const float xa[4] = {ra.x, ra.y, ra.z, ra.w};
          if (rank) {
            unsigned ad[4];
#if !(NMOD_SKIP & 512)
            rank_many(std::integral_constant<int, 4>{}, xa, true, ad);
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) { const double d = (double)xa[e] - KQ; s1w += d; s2w = __fma_rn(d, d, s2w); }
          }
          { const bool hv[4] = {true, true, true, true}; mark_many(std::integral_constant<int, 4>{}, xa, hv); }
          ra = rb;
        }
        // the first rounds of pass 2 are requested HERE, before the tail rounds of pass 1 (a position is only ~4 full rounds long)
        constexpr int PF = 4;
        Q4Raw buf[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) buf[i] = load_q4(sig_q, off_q, i * (4 * LG) + 4 * gl, i < full);
        Q1Raw rt2 = load_q1(sig_q, off_q, full * (4 * LG) + gl, full * (4 * LG) + gl < q);
#pragma unroll 1
        for (int c = 0; c < tail_w; ++c) {
          const int idx_now = full * (4 * LG) + c * LG + gl;
          const bool have = idx_now < q;
          const float xq1[1] = {have ? (float)rt : big};
          const int idx = full * (4 * LG) + (c + 1) * LG + gl;
          rt = load_q1(sig_q, off_q, idx, idx < q);
          if (rank) {
            unsigned a1[1];
#if !(NMOD_SKIP & 512)
            rank_many(std::integral_constant<int, 1>{}, xq1, have, a1);
#endif
            const double d = (double)(have ? xq1[0] : kqf) - KQ;
            s1w += d; s2w = __fma_rn(d, d, s2w);
          }
          { const bool hv[1] = {have}; mark_many(std::integral_constant<int, 1>{}, xq1, hv); }
        }
        const unsigned total_hits = pos_allsum_u32<64>(hits);
#if (NMOD_SKIP & 2048)
        return;                                                    // (timing experiment: no second pass)
#endif
        if (total_hits == 0u) return;                              // no two samples on one bit: no ties inside Q
        if (2u * total_hits + 64u > (3u * tslots) / 4u) {          // (a bit shared by c samples: c - 1 hits, c <= 2 (c - 1) samples to walk)
          if (valid && lane == 0) args.redo_list[atomicAdd(args.redo_count, 1)] = (int32_t)pos;
          redo_flag = true;
          return;
        }
        // pass 2: B1's words become the exact table; B2 stays
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < W1 / 4; i += 64) reinterpret_cast<uint4*>(ht)[i] = make_uint4(kWideEmpty, kWideEmpty, kWideEmpty, kWideEmpty);
        rt = rt2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // a sample on a marked bit waits in a register of its lane (`pend`); the waiting samples of all lanes go to the list
        // together when some lane gets a second one (a ballot + branch per sample slot instead of a compaction per slot)
        unsigned pend = 0u;
        bool full_l = false;
        auto flush = [&]() {
          const unsigned long long mk = __ballot(full_l);
          if (mk != 0ull) {
            const int at = lcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            if (full_l) lst[at] = pend;
            lcnt += __popcll(mk);
            full_l = false;
            while (lcnt >= 64) drain(64);
          }
        };
        auto collect_many = [&](auto nv_tag, const float* xv, const bool* have) {
          constexpr int NV = decltype(nv_tag)::value;
          unsigned bits[NV], idx[NV], w[NV];
#pragma unroll
          for (int e = 0; e < NV; ++e) { bits[e] = __float_as_uint(xv[e] + 0.0f); idx[e] = (bits[e] * 2654435761u) >> SH; }
#if (NMOD_SKIP & 16384)
#pragma unroll
          for (int e = 0; e < NV; ++e) w[e] = idx[e];                // (timing experiment: no bitmap reads)
#else
#pragma unroll
          for (int e = 0; e < NV; ++e) w[e] = B2[idx[e] >> 5];
#endif
#pragma unroll
          for (int e = 0; e < NV; ++e) {
            bool sus = have[e] && ((w[e] >> (idx[e] & 31u)) & 1u) != 0u;
#if (NMOD_SKIP & 8192)
            ppq += sus ? 1u : 0u; sus = false;                        // (timing experiment: no list, no walks)
#endif
            if (__ballot(sus && full_l) != 0ull) flush();
            pend = sus ? bits[e] : pend;
            full_l = full_l || sus;
          }
        };
#pragma unroll 1
        for (int c = 0; c < full_w; c += PF) {
          Q4Raw nxt[PF];
#pragma unroll
          for (int i = 0; i < PF; ++i) nxt[i] = load_q4(sig_q, off_q, (c + PF + i) * (4 * LG) + 4 * gl, c + PF + i < full);
#pragma unroll
          for (int i = 0; i < PF; ++i) {
            if (c + i < full_w) {
              const float xa[4] = {buf[i].x, buf[i].y, buf[i].z, buf[i].w};
              const bool hv[4] = {true, true, true, true};
              collect_many(std::integral_constant<int, 4>{}, xa, hv);
            }
          }
#pragma unroll
          for (int i = 0; i < PF; ++i) buf[i] = nxt[i];
        }
#pragma unroll 1
        for (int c = 0; c < tail_w; ++c) {
          const bool have = full * (4 * LG) + c * LG + gl < q;
          const float xv[1] = {have ? (float)rt : big};
          const int idx = full * (4 * LG) + (c + 1) * LG + gl;
          rt = load_q1(sig_q, off_q, idx, idx < q);
          const bool hv[1] = {have};
          collect_many(std::integral_constant<int, 1>{}, xv, hv);
        }
        flush();
        if (lcnt > 0) drain(lcnt);
      };
      if (!s_grid) {
        bitmap_passes(true);
      } else {
        // ---- S is on the milli-unit grid: rank and sum the moments as above, count Q's ties by VALUE with the direct-address
        // counters of the int16 form (window of 4 x wslots = 8 192 values centred on the median of S) — as long as every
        // sample of Q is on the grid too
        const int wb = (int)__builtin_rintf(__fmul_rn(keys[Lay::word(m >> 1)], 1000.0f)) - 2 * wcount;
        bool offl = false;                            // this lane saw a sample of Q off the grid
        clear_table();
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll 1
        for (int c = 0; c < full_w; ++c) {
          const Q4Raw rb = load_q4(sig_q, off_q, (c + 1) * (4 * LG) + 4 * gl, c + 1 < full);
          const float xa[4] = {ra.x, ra.y, ra.z, ra.w};
          unsigned ad[4];
#if !(NMOD_SKIP & 512)
          rank_many(std::integral_constant<int, 4>{}, xa, true, ad);
#endif
          int iv[4]; bool hv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const double d = (double)xa[e] - KQ; s1w += d; s2w = __fma_rn(d, d, s2w);
            hv[e] = grid_key<false>(xa[e], iv[e]);
            offl = offl || !hv[e];
          }
          count_many(std::integral_constant<int, 8>{}, std::integral_constant<int, 4>{}, iv, hv, wb);
          ra = rb;
        }
#pragma unroll 1
        for (int c = 0; c < tail_w; ++c) {
          const int idx_now = full * (4 * LG) + c * LG + gl;
          const bool have = idx_now < q;
          const float xq1[1] = {have ? (float)rt : big};
          const int idx = full * (4 * LG) + (c + 1) * LG + gl;
          rt = load_q1(sig_q, off_q, idx, idx < q);
          unsigned a1[1];
#if !(NMOD_SKIP & 512)
          rank_many(std::integral_constant<int, 1>{}, xq1, have, a1);
#endif
          const double d = (double)(have ? xq1[0] : kqf) - KQ;
          s1w += d; s2w = __fma_rn(d, d, s2w);
          int iv1[1];
          const bool ok = grid_key<false>(xq1[0], iv1[0]);
          offl = offl || (have && !ok);
          const bool hv[1] = {have && ok};
          count_many(std::integral_constant<int, 8>{}, std::integral_constant<int, 1>{}, iv1, hv, wb);
        }
        if (__ballot(offl) != 0ull) {                 // Q has samples off the grid: its ties through the bitmaps after all
          ppq = 0u;
          bitmap_passes(false);
        } else if (redo != 0ull || wtails > kWideTail) {
          if (!recount16([&](int i, bool have) { int k; grid_key<false>((float)load_q1(sig_q, off_q, i, have), k); return k; })) {
            ppq = 0u;                                 // keys over more than 65 535 milli-units: the bitmaps
            bitmap_passes(false);
          }
        } else if (wtails > 1) tail_ties();
      }
    } else {
    __builtin_amdgcn_s_waitcnt(0x0F70);            // everything requested before the sort has arrived
#if (NMOD_SKIP & 32)
    full_w = 0; tail_w = 0;
#endif
#pragma unroll
    for (int c = 0; c < R / 4; ++c) {
      if (c < full_w) {
        const Q4Raw rb = load_q4(sig_q, off_q, (c + 1) * (4 * LG) + 4 * gl, c + 1 < full);
        const bool have = c < full;
        float xa[4];
        if constexpr (DTYPE == 0) { xa[0] = ra.x; xa[1] = ra.y; xa[2] = ra.z; xa[3] = ra.w; }
        else { xa[0] = have ? (float)ra.x : big; xa[1] = have ? (float)ra.y : big; xa[2] = have ? (float)ra.z : big; xa[3] = have ? (float)ra.w : big; }
        unsigned ad[4];
        rank_many(std::integral_constant<int, 4>{}, xa, have, ad);
#pragma unroll
        for (int e = 0; e < 4; ++e) xs[4 * c + e] = xa[e];
        la[2 * c] = ad[0] | (ad[1] << 16);
        la[2 * c + 1] = ad[2] | (ad[3] << 16);
        ra = rb;
      }
    }
    // tail rounds: one sample per lane, kept in slot 4 * full + c of the lane (never beyond R - 1: q <= C).  Per
    // lane: a position with fewer full rounds than the wave's longest reuses the slots of its idle vector rounds.
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < tail_w) {
        const int idx_now = full * (4 * LG) + c * LG + gl;
        const bool have = idx_now < q;
        const float xq = have ? (float)rt : big;
        const int idx = full * (4 * LG) + (c + 1) * LG + gl;
        rt = load_q1(sig_q, off_q, idx, idx < q);
        unsigned ad;
        { const float x1[1] = {xq}; unsigned a1[1]; rank_many(std::integral_constant<int, 1>{}, x1, have, a1); ad = a1[0]; }
#pragma unroll
        for (int f = 0; f < R / 4; ++f)
          if (4 * f + c < R) {
            const bool me = f == full;
            xs[4 * f + c] = me ? xq : xs[4 * f + c];
            unsigned& w = la[(4 * f + c) / 2];
            w = me ? ((c & 1) ? ((w & 0xffffu) | (ad << 16)) : ((w & 0xffff0000u) | ad)) : w;
          }
      }
    }
    }   // !WIDE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // (a fresh opaque copy of the lane number for the second half of the item: otherwise the index constants of the
    // sweep over S above stay in registers through the ranking rounds for the sweeps below)
    asm volatile("" : "+v"(lane));
    const int gl2 = lane & (LG - 1);
    const int e02 = gl2 * R;

    // ---- one pass over the lane's bins: histograms -> prefix table in their place (bin k gets
    // cumL(k-1) << 16 | (a run of S ends at k) << 15 | cumU(k): the scatter's bases and the float-form pass read it),
    // and on the way the KS numerator, the Mann-Whitney sum and the ties between S and Q
    unsigned cum;
    int maxc;
    unsigned best;
    unsigned acc_l = 0, ab1 = 0, ab3 = 0;
    unsigned long long ab3w = 0ull;                // WIDE: a b (a + b) can pass 2^32 (256 keys tied with 4 096 samples)
    {
      // first the totals of the lane's bins e02 + 1 .. e02 + R (bin e02 + R: row 0 of the next column).  Nothing of this
      // pass is kept: the sweep below reads bins and keys again, four at a time — both arrays in registers next to
      // the samples and bin offsets kept for the scatter did not fit 128 VGPRs
      const unsigned h0 = hist[0];
      unsigned tot = 0, hmax = h0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const unsigned v = (r < R - 1) ? hist[(r + 1) * ROW + gl2] : hist[gl2 + 1];
        tot += v; hmax = max(hmax, v);
      }
      cum = seg_exscan_add_u32<LG>(tot, gl2) + h0;                           // cumL(e02) << 16 | (ties counted up to bin e02)
      maxc = (int)(wave_max_u32(hmax) >> 16);                               // fullest L-bin of the wave's positions
      // the run of S that is open when the lane's first bin begins: its start and the samples of Q equal to it (only
      // needed where Q ties with S: a low half of the wave's bins is not 0)
      int start = 0, brun = (int)(h0 & 0xffffu);
      const bool wave_ties = __ballot(((tot | h0) & 0xffffu) != 0u) != 0ull;
      if (wave_ties) {
        int ls = 0;
        float kp = keys[gl2];
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const float kn = (r < R - 1) ? keys[(r + 1) * ROW + gl2] : keys[gl2 + 1];
          ls = (kp != kn) ? (e02 + r + 1) : ls;                              // a run ends at key e02 + r: the next starts at e02 + r + 1
          kp = kn;
        }
        const int bias = (LG == 8 && (lane & 8)) ? C + 1 : 0;
        int sc = lane_prev_i(seg_scan_max_i32<LG>(ls + bias), 0) - bias;
        start = (gl2 == 0 || sc < 0) ? 0 : sc;
        brun = (int)(hist[Lay::word(start)] & 0xffffu);                      // (read before the table overwrites the bins)
      }
      __builtin_amdgcn_sched_barrier(0);
      int cl = (int)(cum >> 16);                                             // cumL(k-1) entering bin k = e02 + 1
      // k = 0: (cumU(0), 0), cumU(0) = the samples below key 0 = cntL[0] - eq[0]
      int hi = (gl2 == 0) ? __mul24((int)(h0 >> 16) - (int)(h0 & 0xffffu), m) : 0, lo = 0;
      int nkq = -__mul24(e02, q);
      float kprev = keys[gl2];                                               // key e02
      constexpr int CH = (R >= 4) ? 4 : R;                                    // bins per chunk
#pragma unroll
      for (int r0 = 0; r0 < R; r0 += CH) {
        // the chunk's bins and keys, requested together and pinned here: one LDS round trip per chunk
        unsigned hc[CH]; float kc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) {
          const int r = r0 + j;
          hc[j] = (r < R - 1) ? hist[(r + 1) * ROW + gl2] : hist[gl2 + 1];
          kc[j] = (r < R - 1) ? keys[(r + 1) * ROW + gl2] : keys[gl2 + 1]; // key k (or the +inf sentinel)
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) asm volatile("" : "+v"(hc[j]), "+v"(kc[j]));
#pragma unroll
        for (int j = 0; j < CH; ++j) {                                       // bin k = e02 + r + 1, the boundary between keys k - 1 and k
          const int r = r0 + j;
          const unsigned hr = hc[j];
          const float knext = kc[j];
          const bool run_end = kprev != knext;
          kprev = knext;
          const int k = e02 + r + 1;
          nkq -= q;
          acc_l += (unsigned)cl;
          const int cand_b = __mul24(cl, m) + nkq;                           // (cumL(k-1), k)
          const unsigned wl = (unsigned)cl << 16;
          const int eqk = (int)(hr & 0xffffu);                               // samples of Q equal to key k (a run start, or 0)
          cl += (int)(hr >> 16);                                             // cumL(k)
          const int cu = cl - eqk;                                           // cumU(k)
          const unsigned w = wl | (run_end ? 0x8000u : 0u) | (unsigned)cu;
          if (r < R - 1) hist[(r + 1) * ROW + gl2] = w; else hist[gl2 + 1] = w;
          const int cand_a = __mul24(cu, m) + nkq;                           // (cumU(k), k)
          const int ca = run_end ? cand_a : 0, cb = run_end ? cand_b : 0;
          hi = max(hi, max(ca, cb));
          lo = min(lo, min(ca, cb));
          // the ties of the run of S that ends at k with the `brun` samples of Q equal to it: a wave-uniform branch,
          // taken for the few boundaries where some lane closes a tied run
          if (wave_ties && __ballot(run_end && brun != 0) != 0ull) {
            const int b = run_end ? brun : 0;
            const int a = k - start;
            const unsigned t = (unsigned)__mul24(a, b);
            ab1 += t;
            if constexpr (WIDE) ab3w += (unsigned long long)t * (unsigned long long)(unsigned)(a + b);
            else ab3 += (unsigned)__mul24((int)t, a + b);
          }
          start = run_end ? k : start;
          brun = run_end ? eqk : brun;
        }
        __builtin_amdgcn_sched_barrier(0);                                  // (bounds what is alive at once: four bins and keys)
      }
      if (gl2 == 0) hist[0] = 0u;                                            // bin 0: nothing before it
      best = (unsigned)max(hi, -lo);
      // (finished HERE: left to the scheduler, the tie and rank sums sink to the end of the item and keep the
      // per-bin counts they are built from in registers across the scatter, the float-form pass and the clean-up)
      asm volatile("" : "+v"(ab3), "+v"(acc_l), "+v"(ab1), "+v"(best));
      if constexpr (WIDE) asm volatile("" : "+v"(ab3w));
      __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned lbest = best;
    best = seg_allmax_u32<LG>(best);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

#if !(NMOD_SKIP & 64)
    if constexpr (WIDE) {
      // Q's moments from the sums of the ranking rounds (shifted by its first sample)
      const float kqf = (q > 0) ? (float)rk : 0.0f;
      const double KQ = (double)kqf;
      const double s1 = seg_allsum_f64<LG>(s1w), s2 = seg_allsum_f64<LG>(s2w);
      const double rn = rq_w;
      double mu = KQ + s1 * rn;
      double qq = s2 - s1 * s1 * rn;
      if constexpr (DTYPE != 0) { mu = mu * 1e-3; qq = qq * 1e-6; }
      if (valid && gl2 == 0) {
        double* mo = args.moments + pos * 4 + (swap ? 0 : 2);
        mo[0] = mu; mo[1] = qq;
      }
    } else {
    // ---- scatter Q by bin into the key words of S (dead from here on: the table carries its run ends).  The high
    // half of a bin's table word is its bump allocator: afterwards it holds cumL(k).  Q's moments on the way, shifted
    // by its first sample (a slot without a sample adds 0).
    {
      const float kqf = (q > 0) ? (float)rk : 0.0f;
      const double KQ = (double)kqf;
      double s1 = 0.0, s2 = 0.0;
      // (all the allocations first, then all the stores: an LDS store cannot be moved above an atomic it might alias,
      // so interleaving them would expose one atomic round trip per sample)
      unsigned at[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool have = xs[r] != big;
        const unsigned ad = (r & 1) ? (la[r / 2] >> 16) : (la[r / 2] & 0xffffu);
        at[r] = 0u;
        if (have) at[r] = atomicAdd(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(keys) + ad), 0x10000u);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const bool have = xs[r] != big;
        const double d = (double)(have ? xs[r] : kqf) - KQ;
        s1 += d;
        s2 = __fma_rn(d, d, s2);
        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = (int)(at[r] >> 16);
        if (xs[r] != big) keys[__mul24(i & (R - 1), ROW) + (i >> LOG_R)] = xs[r];
      }
      s1 = seg_allsum_f64<LG>(s1);
      s2 = seg_allsum_f64<LG>(s2);
      const double dn = (double)q;
      const double rn = uniform ? recip[1] : 1.0 / dn;             // one division for mean and M2, none for a fixed-stride batch
      double mu = KQ + s1 * rn;
      double qq = s2 - s1 * s1 * rn;
      if constexpr (DTYPE != 0) { mu = mu * 1e-3; qq = qq * 1e-6; }
      if (valid && gl2 == 0) {
        double* mo = args.moments + pos * 4 + (swap ? 0 : 2);
        mo[0] = mu; mo[1] = qq;
      }
    }
    }   // !WIDE
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#endif
    // ---- the float form of D, only for the candidates that reach the integer maximum: the lanes of a position
    // take the bins of one such lane at a time from the table
    double dmax = 0.0;
    {
      int mo = m, qo = q;
      asm volatile("" : "+v"(mo), "+v"(qo));    // (not the (double)m of the moments, kept alive since then)
      const double dm = (double)mo, dq = (double)qo;
      double rm, rq;
      if constexpr (WIDE) { rm = rm_w; rq = rq_w; }
      else if (uniform) { rm = recip[0]; rq = recip[1]; }
      else { rm = 1.0 / dm; rq = 1.0 / dq; }
      // hit lanes of this lane's position as a bit mask (bit j: lane j of the group reached the maximum)
      const unsigned long long hits = __ballot(lbest == best && best != 0u);
      using Mine = typename std::conditional<LG == 64, unsigned long long, unsigned>::type;
      Mine mine = (Mine)(hits >> seg_base);
      if constexpr (LG < 32) mine &= (Mine)((1u << LG) - 1u);
#if (NMOD_SKIP & 2)
      mine = 0;
#endif
      // the candidate (cumU(0), 0) belongs to lane 0 of the position: bin 0's table word was cleared, its counts are
      // what `cum` of lane 0 started from
      if (gl2 == 0 && (mine & (Mine)1)) {
        const int cu0 = (int)(cum >> 16) - (int)(cum & 0xffffu);
        if ((unsigned)__mul24(cu0, mo) == best) dmax = hist_exact_quot(cu0, dq, rq);
      }
      // word offsets of the table entries of bins k - 1 and k (k = hl * R + rr + 1) for hl = 0
      int wofs[(R + LG - 1) / LG][2];
#pragma unroll
      for (int j = 0; j < (R + LG - 1) / LG; ++j) {
        const int rr = gl2 + j * LG;
        wofs[j][0] = Lay::word(rr); wofs[j][1] = Lay::word(rr + 1);
      }
#pragma unroll 1
      while (__ballot(mine != (Mine)0) != 0ull) {
        const bool act = mine != (Mine)0;
        const int hl = act ? (__ffsll((long long)mine) - 1) : 0;              // the lane of this position whose bins are examined
        mine &= mine - (Mine)1;
#pragma unroll
        for (int j = 0; j < (R + LG - 1) / LG; ++j) {
          const int rr = gl2 + j * LG;
          const int k = hl * R + rr + 1;
          const bool in = act && rr < R;
          // (lane hl's bins are one column to the right per unit of hl: + hl words)
          const unsigned wp = hist[wofs[j][0] + hl];
          const unsigned w = hist[wofs[j][1] + hl];
          // (after the scatter the high half of word k - 1 is cumL(k - 1); WIDE has no scatter: word k still holds it)
          const int cl = (int)((WIDE ? w : wp) >> 16), cu = (int)(w & 0x7fffu);
          const bool run_end = (w & 0x8000u) != 0u;
          const int nkq = -__mul24(k, qo);
          const int cand_b = __mul24(cl, mo) + nkq, cand_a = __mul24(cu, mo) + nkq;
          const bool hb = in && run_end && (unsigned)abs(cand_b) == best;
          const bool ha = in && run_end && (unsigned)abs(cand_a) == best;
          const double fk = hist_exact_quot(k, dm, rm);
          const double db = fabs(fk - hist_exact_quot(cl, dq, rq));
          const double da = fabs(fk - hist_exact_quot(cu, dq, rq));
          dmax = hb ? fmax(dmax, db) : dmax;
          dmax = ha ? fmax(dmax, da) : dmax;
        }
      }
      dmax = seg_allmax_f64<LG>(dmax);
    }

    // the next item's S rows: requested as late as the rest of the iteration can still cover the round trip (they
    // occupy 16 registers from here on)
    if constexpr (AFTER) it_next = next_in_windows(); else it_next = it + wave_stride;
    const Item nxt = describe(it_next);
    KsRows<(WIDE ? 4 : R), LG, DTYPE> rows_next;                             // (WIDE: unused; its rows are R < 4 plain loads)
    float xn[WIDE ? R : 1];
    if constexpr (WIDE) load_group<R, DTYPE>(xn, nxt.swap ? args.sig1 : args.sig0, nxt.off_s, nxt.m, lane);
    else rows_next.request(nxt.swap ? args.sig1 : args.sig0, nxt.off_s, nxt.m, gl2);

    if constexpr (WIDE) {
      pp += ppq;
    } else {
    // ---- Q in sorted order: read the scattered keys back, finish inside the bins, count its ties
    __builtin_amdgcn_wave_barrier();
    float y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float v = keys[r * ROW + gl2];
      y[r] = (e02 + r < q) ? v : inf;                                          // words past q still hold keys of S
    }
    // (a bin is unordered inside and ordered against its neighbours: maxc phases sort every bin.  Groups far apart
    // put many samples into one end bin — up to q phases of 2 instructions per key, against the register footprint
    // that a second copy of the full network would add to every item)
#if !(NMOD_SKIP & 1)
    if (maxc > 1) seg_oddeven_phases<R, LG>(y, gl2, maxc);
    pp += seg_tie_pp<R, LG>(y, gl2, lane);
#endif
    }

    // ---- totals of the position
    // (WIDE: the totals can pass 2^32 — per lane pp < 2^31 and ab3w < 2^35 — so they are summed in two parts)
    unsigned long long PP, AB;
    if constexpr (WIDE) {
      PP = ((unsigned long long)pos_allsum_u32<LG>(pp >> 16) << 16) + (unsigned long long)pos_allsum_u32<LG>(pp & 0xffffu);
      AB = ((unsigned long long)pos_allsum_u32<LG>((unsigned)(ab3w >> 20)) << 20) + (unsigned long long)pos_allsum_u32<LG>((unsigned)ab3w & 0xfffffu);
    } else {
      PP = pos_allsum_u32<LG>(pp);
      AB = pos_allsum_u32<LG>(ab3);
    }
    const unsigned AL = pos_allsum_u32<LG>(acc_l);
    const unsigned A1 = pos_allsum_u32<LG>(ab1);
    if (valid && gl2 == 0) {
      // sum_{x in Q} (L + U) = 2 sum_{j<C} (q - cumL(j)) + sum_runs a b
      const unsigned long long slu = 2ull * ((unsigned long long)C * (unsigned long long)q - (unsigned long long)AL) + (unsigned long long)A1;
      // mwu_s = sum_{a in group 1} (#{b < a} + #{b <= a}): Q is group 1 when swapped, else count from group 2's side
      args.mwu_s[pos] = swap ? slu : 2ull * (unsigned long long)m * (unsigned long long)q - slu;
      args.tie[pos] = 3ull * PP + 3ull * AB;
      args.ks_d_ref[pos] = (m > 0 && q > 0) ? dmax : 0.0;
      args.ks_num[pos] = (m > 0 && q > 0) ? best : 0u;
      if (args.tied) args.tied[pos] = (PP != 0ull || AB != 0ull || redo_flag) ? 1 : 0;   // any two samples of the position compare equal (a position on the redo list: not known here)
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if constexpr (WIDE) {
#pragma unroll
      for (int r = 0; r < R; ++r) x[r] = xn[r];
    } else {
      if constexpr (PACKED) rows_next.finish_packed(pk, nxt.m, gl);
      else rows_next.finish(x, nxt.m, gl);
    }
    cur = nxt;
  }
}

}  // namespace nmod
