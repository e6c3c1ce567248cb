// K1, all-tests form for positions whose two groups fall in DIFFERENT capacity classes (e.g. 1000 v 50 reads):
// rank_pair_kernel, one position per wave, both groups sorted by the 64-lane network (tests mask = MWU | Welch | KS:
// what getKStest computes for every position, myDetect.py:327-343).  Same-class positions take rank_hist.hpp.
//
// Call group 1 "A" (m samples) and group 2 "B" (q samples).  Both groups are sorted, the moments are taken on the way,
// and the run extents of equal keys are written next to the keys.  Every lane then takes samples x of the group with
// FEWER samples (sorted index j, run [j_s, j_e)) and finds L = #{t < x} by a branchless binary search in the other
// group; a tie gives U = #{t <= x} from that group's run table.  At the end of every run that is everything the
// three tests need:
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
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ks_rank.hpp"

namespace nmod {


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

// fl(c / n) for an integer 0 <= c <= n <= 4096, r = fl(1 / n)
__device__ __forceinline__ double exact_quot(int c, double n, double r) {
  const double dc = (double)c;
  const double q0 = __dmul_rn(dc, r);
  const double rem = __fma_rn(-q0, n, dc);
  return __fma_rn(rem, r, q0);
}

// ---- any two capacity classes: one position per wave, 64 lanes per group ---------------------------------
// (e.g. 1000 v 50 reads.)  The group with FEWER samples is ranked into the
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
  if (args.alt_gates != nullptr && args.alt_gates[args.class_id] != 0) {     // what rank_count_wide_kernel left of the class
    count = args.alt_meta[args.class_id];
    list = args.alt_list + args.alt_meta[kClassStride + args.class_id];
  } else if (args.pos_list) {
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
      if (args.tied) args.tied[pos] = (PP != pads || T3 != 0ull) ? 1 : 0;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace nmod
