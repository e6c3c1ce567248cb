// K1, all-tests form for positions whose two groups fall in the same size class
// (tests mask = MWU | Welch | KS: what getKStest computes for every position, myDetect.py:327-343).
//
// Both groups are sorted at once, each in R registers x LG lanes (the two halves of a position's
// 2*LG lanes run the same bitonic network), the moments are taken on the way, and the run extents of
// equal keys are written next to the keys.  Call group 1 "A" (m samples) and group 2 "B" (q samples).
// Every lane of the position then takes B samples x (sorted index j, run [j_s, j_e)) and finds
// L = #{a < x} by a branchless binary search in A; a tie with A gives U = #{a <= x} from A's run table.
// At the end of every B run that is everything the three tests need:
//   * Mann-Whitney:  sum_{a in A} (2 #{b < a} + #{b == a}) = sum_{b in B} (2m - U(b) - L(b))
//   * tie correction: sum over pooled tie groups of t^3 - t
//                     = 3 sum_{elements of A and B} p (p - 1)      (p = position inside its own run)
//                     + 3 sum_{B runs tied with an A run} a b (a + b)
//   * KS:  max over the pooled points of |F_A - F_B| is reached either at a B value
//          (counts (U, j_e)) or at the pooled point just below one (counts (L, j_s)): between two
//          B values F_B is constant and F_A monotone.  ks_2samp forms D as fl(c0/n0) - fl(c1/n1) in
//          fp64; a larger integer numerator |c0 n1 - c1 n0| always gives a larger float D
//          (they differ by >= 1/(n0 n1)), so the maximum of the float form over all candidates is
//          the reference's D bit for bit.  c/n is formed as q0 = c r, q = fma(fma(-q0, n, c), r, q0)
//          with r = fl(1/n): correctly rounded for every c <= n <= 4096 (checked exhaustively).
// This replaces the merge-path walk of the pooled sample (one sequential step per pooled element,
// ~80 VALU instructions each) by one search per B sample spread over all lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ks_rank.hpp"

namespace nmod {

__host__ __device__ constexpr int rank_all_pos_words(int R, int LG) {
  int w = 4 * R * (LG + 1);                               // A keys, A runs, B keys, B runs: four KsLayout regions
  if (LG == 8) while ((w & 31) != 16) ++w;                // two positions share a 32-lane half: 16 banks apart
  return w;
}

// Run extents of equal keys inside each sorted group, (start | end << 16) per element at dst_lane[r * STRIDE],
// and pp = sum over the lane's elements of p (p - 1), p = 1-based position of the element in its run.
template <int R, int LG, int STRIDE>
__device__ __forceinline__ void seg_runs_and_ties(int* dst_lane, float (&y)[R], int gl, bool is_b, unsigned& pp) {
  constexpr int N = R * LG;
  // LG == 8: both groups of a position share one DPP row: the second group's scan values are biased by N,
  // so whatever leaks in from the first group (< N) can never win a max
  const int bias = (LG == 8 && is_b) ? N : 0;
  const float nanv = __builtin_nanf("");
  float prev_last = lane_prev(y[R - 1], nanv);
  float next_first = lane_next(y[0], nanv);
  prev_last = (gl == 0) ? nanv : prev_last;
  next_first = (gl == LG - 1) ? nanv : next_first;
  // each sweep works on an opaque in-place copy of the keys: otherwise the compiler keeps the 2*R comparison
  // masks of all sweeps alive in SGPRs and spills them
  auto launder = [&]() {
#pragma unroll
    for (int r = 0; r < R; ++r) asm volatile("" : "+v"(y[r]));
  };
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
  unsigned short* dst16 = reinterpret_cast<unsigned short*>(dst_lane);
  launder();
  run = carry;
  unsigned acc_pp = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float p = (r == 0) ? prev_last : y[r - 1];
    const int e = gl * R + r + bias;
    run = (y[r] != p) ? e : run;
    const unsigned pm1 = (unsigned)(e - run);
    acc_pp += __umul24(pm1, pm1) + pm1;
    dst16[2 * r * STRIDE] = (unsigned short)(run - bias);
    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
  pp = acc_pp;
  launder();
  acc = carry_r;
#pragma unroll
  for (int r = R - 1; r >= 0; --r) {
    float q = (r == R - 1) ? next_first : y[r + 1];
    acc = (y[r] != q) ? max(acc, N - (gl * R + r + 1) + bias) : acc;
    dst16[2 * r * STRIDE + 1] = (unsigned short)(N - (acc - bias));
    if ((r & 3) == 0) __builtin_amdgcn_sched_barrier(0);
  }
}

// sum_{p = 1..P} p (p - 1) = (P - 1) P (P + 1) / 3 for the run of P <= 2048 pads, in 32-bit arithmetic:
// one of the three factors is divisible by 3 and the quotient is < 2^32
__device__ __forceinline__ unsigned pad_run_pp(int P) {
  const unsigned p = (unsigned)P;
  const unsigned t = (p * 43691u) >> 17;                 // p / 3 for p < 2^16
  const unsigned r = p - 3u * t;
  const unsigned a = (r == 1u) ? (p - 1u) / 3u : p - 1u;  // (the compiler turns / 3 into the same multiply)
  const unsigned b = (r == 0u) ? t : p;
  const unsigned c = (r == 2u) ? (p + 1u) / 3u : p + 1u;
  return a * b * c;
}

// sum of a per-lane fp64 value over the 2*LG lanes of a position
template <int LG>
__device__ __forceinline__ double pos_allsum_f64(double v) {
  if constexpr (LG == 8) return seg_allsum_f64<16>(v);
  else if constexpr (LG == 16) return seg_allsum_f64<32>(v);
  else return wave_sum_f64(v);
}

// fl(c / n) for an integer 0 <= c <= n <= 4096, r = fl(1 / n)
__device__ __forceinline__ double exact_quot(int c, double n, double r) {
  const double dc = (double)c;
  const double q0 = __dmul_rn(dc, r);
  const double rem = __fma_rn(-q0, n, dc);
  return __fma_rn(rem, r, q0);
}

template <int R, int LG, int DTYPE>
__global__ __launch_bounds__(64 * kWavesPerBlock, (R <= 16 ? 4 : 2))
void rank_all_kernel(RankStatsArgs args) {
  static_assert(LG == 8 || LG == 16 || LG == 32, "lanes per group");
  static_assert(R >= 8 && R <= 32, "registers per lane");
  constexpr int C = R * LG;                       // capacity per group
  constexpr int LP = 2 * LG;                      // lanes per position
  constexpr int PW = 64 / LP;                     // positions per wave
  using Lay = KsLayout<R, LG>;
  constexpr int ROW = Lay::ROW;
  constexpr int REGION = Lay::REGION;             // keys or runs of one group (KsLayout: ks_rank.hpp)
  constexpr int POS_WORDS = rank_all_pos_words(R, LG);
  constexpr int LOG_R = (R == 8) ? 3 : (R == 16) ? 4 : 5;
  static_assert((1 << LOG_R) == R, "registers per lane: 8, 16 or 32");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gl = lane & (LG - 1);                 // lane inside its group
  const int pl = lane & (LP - 1);                 // lane inside its position
  const int slot = lane / LP;                     // which of the wave's positions
  const bool is_b = (lane & LG) != 0;             // second group of the position

  float* keysA = lds_all + (wave * PW + slot) * POS_WORDS;      // run words at + REGION
  float* keysB = keysA + 2 * REGION;                              // run words at + REGION
  float* my_keys = (is_b ? keysB : keysA) + gl;                   // this lane's column: register r at + r * ROW
  int* my_runs = reinterpret_cast<int*>(my_keys + REGION);

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  for (int r = pl; r < R; r += LP) {                                              // the spare column of A
    keysA[r * ROW + Lay::END] = inf;                                                // rank C: every key is below x
    reinterpret_cast<int*>(keysA + REGION)[r * ROW + Lay::END] = C | (C << 16);
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

  struct Item { bool valid; int n0, n1; int64_t pos, my_off; };
  auto describe = [&](int64_t it) {
    Item d;
    const int64_t li = it * PW + slot;
    d.valid = it < items && li < count;
    d.pos = d.valid ? (list ? (int64_t)list[li] : li) : 0;
    int64_t o0 = 0, o1 = 0;
    d.n0 = 0; d.n1 = 0;
    if (d.valid) {
      if (args.stride0 > 0) { o0 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride0); d.n0 = (int)args.stride0; }
      else { o0 = args.off0[d.pos]; d.n0 = (int)(args.off0[d.pos + 1] - o0); }
      if (args.stride1 > 0) { o1 = (int64_t)((uint64_t)(uint32_t)d.pos * (uint64_t)(uint32_t)args.stride1); d.n1 = (int)args.stride1; }
      else { o1 = args.off1[d.pos]; d.n1 = (int)(args.off1[d.pos + 1] - o1); }
    }
    d.my_off = is_b ? o1 : o0;
    return d;
  };
  const void* my_sig = is_b ? args.sig1 : args.sig0;

  // software pipeline (see ks_rank_kernel): the rows of the next item are requested at the top of the loop and
  // turned into keys at the bottom, a whole item later
  Item cur = describe(wave_global);
  float x[R];
  {
    KsRows<R, LG, DTYPE> first;
    first.request(my_sig, cur.my_off, is_b ? cur.n1 : cur.n0, gl);
    first.finish(x, is_b ? cur.n1 : cur.n0, gl);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)

  for (int64_t it = wave_global; it < items; it += wave_stride) {
    const bool valid = cur.valid;
    const int64_t pos = cur.pos;
    const int n0 = cur.n0, n1 = cur.n1;
    const Item nxt = describe(it + wave_stride);
    KsRows<R, LG, DTYPE> rows_next;
    rows_next.request(my_sig, nxt.my_off, is_b ? nxt.n1 : nxt.n0, gl);

    {
      double mean, m2;
      seg_moments<R, LG, DTYPE>(x, is_b ? n1 : n0, mean, m2);
      if (valid && gl == 0) {
        double* mo = args.moments + pos * 4 + (is_b ? 2 : 0);
        mo[0] = mean; mo[1] = m2;
      }
    }
    seg_sort<R, LG>(x, sel, lane);
#pragma unroll
    for (int r = 0; r < R; ++r) my_keys[r * ROW] = x[r];
    unsigned pp;
    seg_runs_and_ties<R, LG, ROW>(my_runs, x, gl, is_b, pp);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    // ---- every lane of the position ranks B samples pl, pl + LP, ... into A
    const int m = n0, q = n1;
    const int per = (q + LP - 1) / LP;
    int steps_w;
    if constexpr (PW == 4) {
      steps_w = max(max(__builtin_amdgcn_readlane(per, 0), __builtin_amdgcn_readlane(per, 16)),
                    max(__builtin_amdgcn_readlane(per, 32), __builtin_amdgcn_readlane(per, 48)));
    } else if constexpr (PW == 2) {
      steps_w = max(__builtin_amdgcn_readlane(per, 0), __builtin_amdgcn_readlane(per, 32));
    } else {
      steps_w = __builtin_amdgcn_readfirstlane(per);
    }
    const double dm = (double)m, dq = (double)q;
    const double rm = 1.0 / dm, rq = 1.0 / dq;
    const int two_m = 2 * m;
    const bool a_full = __ballot(m == C) != 0ull;       // no +inf pad in some group 1 of the wave: rank C can occur
    unsigned s_lane = 0;
    unsigned long long tie3 = 0;
    double dmax = 0.0;
#pragma unroll 1
    for (int s = 0; s < steps_w; ++s) {
      const int jq = s * LP + pl;
      const int wq = __mul24(jq & (R - 1), ROW) + (jq >> LOG_R);
      const float xq = keysB[wq];
      const int rb = reinterpret_cast<const int*>(keysB + REGION)[wq];
      const int rb_s = rb & 0xffff, rb_e = (int)((unsigned)rb >> 16);
      const bool cand = (jq < q) && (rb_e == jq + 1);               // the end of a run of B
      const float* p = a_full ? ks_search<R, LG, false, true>(keysA, xq) : ks_search<R, LG, false, false>(keysA, xq);
      const int ra = *reinterpret_cast<const int*>(p + REGION);      // p is the first key of its run: start == L
      const bool tie = (*p == xq);
      const int L = ra & 0xffff;
      const int U = tie ? (int)((unsigned)ra >> 16) : L;
      const int a = U - L, b = rb_e - rb_s;
      s_lane += cand ? (unsigned)__mul24(b, two_m - U - L) : 0u;
      const unsigned ab = cand ? (unsigned)__mul24(a, b) : 0u;
      if (__ballot(ab != 0u) != 0ull) tie3 += (unsigned long long)ab * (unsigned long long)(unsigned)(a + b);   // (a 64-bit multiply-add)
      const double d_at = exact_quot(U, dm, rm) - exact_quot(rb_e, dq, rq);
      const double d_before = exact_quot(L, dm, rm) - exact_quot(rb_s, dq, rq);
      const double dd = fmax(fabs(d_at), fabs(d_before));
      dmax = cand ? fmax(dmax, dd) : dmax;
    }
    dmax = pos_max_f64<LG>(dmax, lane);
    const unsigned long long S = pos_sum_u32<LG>(s_lane, lane);
    const unsigned long long PP = pos_sum_u32<LG>(pp, lane);
    const double t3 = pos_allsum_f64<LG>((double)tie3);              // < 2^53: exact
    if (valid && pl == 0) {
      // the +inf pads of each group form one run of P = C - n keys: take its sum_{p<=P} p (p - 1) = (P^3 - P) / 3 out
      const unsigned long long pads = (unsigned long long)pad_run_pp(C - m) + (unsigned long long)pad_run_pp(C - q);
      args.mwu_s[pos] = S;
      args.tie[pos] = 3ull * (PP - pads) + 3ull * (unsigned long long)t3;
      args.ks_d_ref[pos] = (m > 0 && q > 0) ? dmax : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0x0F70);    // vmcnt(0): requested a whole item ago
    rows_next.finish(x, is_b ? nxt.n1 : nxt.n0, gl);
    cur = nxt;
  }
}

// ---- any two capacity classes: one position per wave, 64 lanes per group ---------------------------------
// (e.g. 1000 v 50 reads.)  Same algorithm as rank_all_kernel; the group with FEWER samples is ranked into the
// one with more, whichever of the two it is:
//   E = group 2, T = group 1:  rank sum += b (2m - U - L);  KS candidates (c0, c1) = (U, j_e), (L, j_s)
//   E = group 1, T = group 2:  rank sum += a (L + U)       (2 #{b < x} + #{b == x} per sample of the run);
//                              KS candidates (c0, c1) = (j_e, U), (j_s, L)
// with (j_s, j_e) the run of E and L / U its lower / upper rank in T.
template <int RT, int RE, bool T_IS_A>
__device__ __forceinline__ void rank_pair_phase(const float* keysT, const float* keysE, int nT, int nE, int lane,
                                                unsigned& s_lane, unsigned long long& tie3, double& dmax) {
  using LT = KsLayout<RT, 64>;
  using LE = KsLayout<RE, 64>;
  constexpr int LOG_RE = (RE == 1) ? 0 : (RE == 2) ? 1 : (RE == 4) ? 2 : (RE == 8) ? 3 : (RE == 16) ? 4 : 5;
  const int steps = __builtin_amdgcn_readfirstlane((nE + 63) >> 6);
  const double dT = (double)nT, dE = (double)nE;
  const double rT = 1.0 / dT, rE = 1.0 / dE;
  const bool t_full = nT == 64 * RT;                   // (one position per wave: uniform)
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    const int jq = s * 64 + lane;
    const int wq = __mul24(jq & (RE - 1), LE::ROW) + (jq >> LOG_RE);
    const float xq = keysE[wq];
    const int re = reinterpret_cast<const int*>(keysE + LE::REGION)[wq];
    const int js = re & 0xffff, je = (int)((unsigned)re >> 16);
    const bool cand = (jq < nE) && (je == jq + 1);                 // the end of a run of E
    const float* p = t_full ? ks_search<RT, 64, false, true>(keysT, xq) : ks_search<RT, 64, false, false>(keysT, xq);
    const int rt = *reinterpret_cast<const int*>(p + LT::REGION);   // p is the first key of its run: start == L
    const bool tie = (*p == xq);
    const int L = rt & 0xffff;
    const int U = tie ? (int)((unsigned)rt >> 16) : L;
    const int t = U - L, e = je - js;
    const int w = T_IS_A ? 2 * nT - U - L : L + U;
    s_lane += cand ? (unsigned)__mul24(e, w) : 0u;
    const unsigned te = cand ? (unsigned)__mul24(t, e) : 0u;
    if (__ballot(te != 0u) != 0ull) tie3 += (unsigned long long)te * (unsigned long long)(unsigned)(t + e);
    const double d_at = exact_quot(U, dT, rT) - exact_quot(je, dE, rE);      // |F_T - F_E|: the sign does not matter
    const double d_before = exact_quot(L, dT, rT) - exact_quot(js, dE, rE);
    const double dd = fmax(fabs(d_at), fabs(d_before));
    dmax = cand ? fmax(dmax, dd) : dmax;
  }
}

template <int R0, int R1, int DTYPE>
__global__ __launch_bounds__(64 * kWavesPerBlock)
void rank_pair_kernel(RankStatsArgs args) {
  using LA = KsLayout<R0, 64>;
  using LB = KsLayout<R1, 64>;
  constexpr int C0 = 64 * R0, C1 = 64 * R1;
  constexpr int WAVE_WORDS = 2 * LA::REGION + 2 * LB::REGION;     // keys A, runs A, keys B, runs B
  extern __shared__ __attribute__((aligned(16))) float lds_all[];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* keysA = lds_all + wave * WAVE_WORDS;
  float* keysB = keysA + 2 * LA::REGION;

  const float inf = __builtin_inff();
  LaneSel sel;
#pragma unroll
  for (int b = 0; b < 6; ++b) sel.s[b] = ((lane >> b) & 1) ? inf : -inf;
  for (int r = lane; r < R0; r += 64) {                            // the spare columns: key / rank C
    keysA[r * LA::ROW + LA::END] = inf;
    reinterpret_cast<int*>(keysA + LA::REGION)[r * LA::ROW + LA::END] = C0 | (C0 << 16);
  }
  for (int r = lane; r < R1; r += 64) {
    keysB[r * LB::ROW + LB::END] = inf;
    reinterpret_cast<int*>(keysB + LB::REGION)[r * LB::ROW + LB::END] = C1 | (C1 << 16);
  }

  int64_t count = args.npos;
  const int32_t* list = nullptr;
  if (args.pos_list) {
    count = args.class_meta[args.class_id];
    list = args.pos_list + args.class_meta[kClassStride + args.class_id];
  }
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t wave_stride = (int64_t)gridDim.x * kWavesPerBlock;

  for (int64_t it = wave_global; it < count; it += wave_stride) {
    const int64_t pos = list ? (int64_t)list[it] : it;
    int64_t o0, o1; int n0, n1;
    if (args.stride0 > 0) { o0 = pos * args.stride0; n0 = (int)args.stride0; }
    else { o0 = args.off0[pos]; n0 = (int)(args.off0[pos + 1] - o0); }
    if (args.stride1 > 0) { o1 = pos * args.stride1; n1 = (int)args.stride1; }
    else { o1 = args.off1[pos]; n1 = (int)(args.off1[pos + 1] - o1); }

    unsigned pp0, pp1;
    {
      float xa[R0];
      load_group<R0, DTYPE>(xa, args.sig0, o0, n0, lane);
      double mean, m2;
      group_moments<R0, DTYPE>(xa, n0, lane, mean, m2);
      if (lane == 0) { double* mo = args.moments + pos * 4; mo[0] = mean; mo[1] = m2; }
      wave_sort<R0>(xa, sel, lane);
#pragma unroll
      for (int r = 0; r < R0; ++r) keysA[r * LA::ROW + lane] = xa[r];
      seg_runs_and_ties<R0, 64, LA::ROW>(reinterpret_cast<int*>(keysA + LA::REGION) + lane, xa, lane, false, pp0);
    }
    {
      float xb[R1];
      load_group<R1, DTYPE>(xb, args.sig1, o1, n1, lane);
      double mean, m2;
      group_moments<R1, DTYPE>(xb, n1, lane, mean, m2);
      if (lane == 0) { double* mo = args.moments + pos * 4 + 2; mo[0] = mean; mo[1] = m2; }
      wave_sort<R1>(xb, sel, lane);
#pragma unroll
      for (int r = 0; r < R1; ++r) keysB[r * LB::ROW + lane] = xb[r];
      seg_runs_and_ties<R1, 64, LB::ROW>(reinterpret_cast<int*>(keysB + LB::REGION) + lane, xb, lane, false, pp1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    unsigned s_lane = 0;
    unsigned long long tie3 = 0;
    double dmax = 0.0;
    if (n1 <= n0) rank_pair_phase<R0, R1, true>(keysA, keysB, n0, n1, lane, s_lane, tie3, dmax);
    else rank_pair_phase<R1, R0, false>(keysB, keysA, n1, n0, lane, s_lane, tie3, dmax);

    dmax = wave_max_f64(dmax);
    const unsigned long long S = wave_sum_u64((unsigned long long)s_lane);
    const unsigned long long PP = wave_sum_u64((unsigned long long)pp0 + (unsigned long long)pp1);
    const unsigned long long T3 = wave_sum_u64(tie3);
    if (lane == 0) {
      const unsigned long long pads = (unsigned long long)pad_run_pp(C0 - n0) + (unsigned long long)pad_run_pp(C1 - n1);
      args.mwu_s[pos] = S;
      args.tie[pos] = 3ull * (PP - pads) + 3ull * T3;
      args.ks_d_ref[pos] = (n0 > 0 && n1 > 0) ? dmax : 0.0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nmod
