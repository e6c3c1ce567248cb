// K1, counting form in the VALUE domain (round 6) — all tests or KS only, event-like positions whose groups BOTH hold more than 1 024 samples
// (launch class (5, 5): 1 025 ... 2 048 v 1 025 ... 2 048).  rank_count_wide.hpp keeps the smaller group in registers (<= 16 per lane:
// 1 024 samples) and looks every one of its samples up in the scanned table; here neither group fits, and neither has to: with a word
// per VALUE of the window, a[v] | b[v] << 16 (the copies of value v in group 1 / group 2), every statistic is a sum or a maximum over
// the VALUES in order —
//   A, B      the running sums of a and b: the groups' samples at or below v
//   KS        max over v of |A n2 - B n1| (scipy's ks_2samp evaluates both CDFs at every pooled point: myDetect.py:339); the float form
//             |fl(A / n1) - fl(B / n2)| at the values that reach the integer maximum, in a second walk over the table
//   MWU       mwu_s = sum_v a (B - b + B) = sum over group 1 of (#{b < a} + #{b <= a})          (myDetect.py:331: mannwhitneyu)
//   ties      sum_v (a + b)^3 - n
//   Welch     exact integer moment sums about the window's centre while the samples stream (float32 rows: fp64 sums of x - first sample)
// — so both groups STREAM once (16- / 8-byte loads, one ds_add per sample, nothing comes back) and the work after that is 2 048 table
// entries per position, 32 per lane, whatever the coverage.  Before round 6 these positions ran on round 2's rank_pair_kernel<32,32>
// (both groups sorted by the 64-lane network: 230 registers, two waves per SIMD): 2.5e7 positions/s at 1 025 v 1 025, a twelfth of the
// 1 024 v 1 024 rate (tools/coverage_sweep.sh).
// Window and outliers as in rank_count_wide.hpp: 2 048 values around a robust centre (here: of the first 64 samples of either group);
// a sample outside it goes to the tail list (<= kCvTail = 128: two per lane; ballot + mbcnt); the samples below the window enter the scan as its
// carry-in, and the listed samples are finished by an all-pairs pass in the same value-domain terms: a distinct tail value's copies
// in either group, the samples below it.  A position with a float32 sample off the grid or more than 128 samples outside the window is
// handed on to the class's sorting form (the work list rank_count_wide_kernel appends to).
#pragma once
#include "rank_count_wide.hpp"

namespace nmod {

// the tail list of this form: two listed samples per lane.  A position of this class holds 2 050 ... 4 096 samples, so 10 per mille
// outliers are ~33 tail samples and 64 entries would hand every other position back
constexpr int kCvTail = 128;
constexpr int kCvWaveWords = kCwTableWords + kCvTail;          // 9 760 B per wave: four blocks of four waves per CU
__host__ __device__ constexpr size_t rank_count_value_lds_bytes() { return (size_t)kWavesPerBlock * kCvWaveWords * 4; }

template <int DTYPE, bool KS>
__device__ __forceinline__ void cv_segment(const RankStatsArgs& args, int32_t* work_list, int32_t* work_cnt, unsigned* tbl, int64_t count, int64_t loff,
                                           const int32_t* list, int64_t start, int64_t wave_stride, int lane) {
  constexpr int RDT = (DTYPE == 1) ? 1 : 0;
  using Q4Raw = typename std::conditional<RDT == 0, KsF4, KsS4>::type;
  using Q1Raw = typename std::conditional<RDT == 0, float, int16_t>::type;
  constexpr int W = kCwWindow;
  unsigned* const tail = tbl + kCwTableWords;
  // lane L owns values 32 L .. 32 L + 31 of the window: its block of 36 words (4 pad words first: the lanes' 16-byte reads fall on different banks)
  uint4* const blk = reinterpret_cast<uint4*>(__builtin_assume_aligned(tbl + 36 * lane + 4, 16));
  auto key_of = [&](Q1Raw x, int& k) -> bool {
    if constexpr (DTYPE == 2) return cnt_int_key((float)x, k);
    else if constexpr (DTYPE == 0) return grid_key<true>((float)x, k);
    else { k = (int)x; return true; }
  };
  // the table starts clear (and is left clear by the second walk of every position)
#pragma unroll
  for (int i = 0; i < 8; ++i) blk[i] = make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();

#pragma unroll 1
  for (int64_t it = start; it < count; it += wave_stride) {
    const int64_t pos = list ? (int64_t)list[it] : it;
    int64_t o[2]; int n[2];
    if (args.stride0 > 0) { o[0] = pos * args.stride0; n[0] = (int)args.stride0; } else { o[0] = args.off0[pos]; n[0] = (int)(args.off0[pos + 1] - o[0]); }
    if (args.stride1 > 0) { o[1] = pos * args.stride1; n[1] = (int)args.stride1; } else { o[1] = args.off1[pos]; n[1] = (int)(args.off1[pos + 1] - o[1]); }
    n[0] = __builtin_amdgcn_readfirstlane(n[0]); n[1] = __builtin_amdgcn_readfirstlane(n[1]);
    if (n[0] < 4 || n[0] > kCwMaxQ || n[1] < 4 || n[1] > kCwMaxQ) {          // (not this form's sizes: the sorting form)
      if (lane == 0) work_list[loff + atomicAdd(work_cnt, 1)] = (int)pos;
      continue;
    }
    const Q1Raw* row[2] = {reinterpret_cast<const Q1Raw*>(args.sig0) + o[0], reinterpret_cast<const Q1Raw*>(args.sig1) + o[1]};
    bool bad = false;

    // ---- the window: a robust centre of the first 64 samples of either group — their mean, then the mean of those within 1 024 of it
    int base, c;
    float xf[2] = {0.0f, 0.0f};                           // float32 rows: the groups' first samples (the shift of the moment sums)
    {
      int kk[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const Q1Raw x = ks_global_load<Q1Raw>(row[g] + min(lane, n[g] - 1));
        bad = bad || !key_of(x, kk[g]);
        if constexpr (RDT == 0) xf[g] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)x)));
      }
      const int c0 = (int)__builtin_rintf((float)(int)cw_wave_sum_u32((unsigned)(kk[0] + kk[1])) * (1.0f / 128.0f));
      unsigned pk = 0u;                                    // count << 21 | sum of (k - c0 + 1 024) over the samples within 1 024 of c0
#pragma unroll
      for (int g = 0; g < 2; ++g) { const unsigned dd = (unsigned)(kk[g] - c0 + 1024); pk += dd < 2048u ? dd + (1u << 21) : 0u; }
      const unsigned P2 = cw_wave_sum_u32(pk);
      const int cnt2 = (int)(P2 >> 21), sd2 = (int)(P2 & 0x1fffffu) - 1024 * cnt2;
      const int centre = cnt2 > 0 ? c0 + (int)__builtin_rintf((float)sd2 * __builtin_amdgcn_rcpf((float)cnt2)) : c0;
      base = max(-32768, min(centre - (W >> 1), 32768 - W));
      c = base + (W >> 1);
    }

    // ---- both groups stream once: counted by value, summed for the moments; what lies outside the window is listed
    int listed = 0;
    int is1[2] = {0, 0}; unsigned long long is2[2] = {0ull, 0ull};           // int16 rows: exact sums of k - c and its square
    double fs1[2] = {0.0, 0.0}, fs2[2] = {0.0, 0.0};                         // float32 rows
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const unsigned inc = g ? 0x10000u : 1u;
      const double K = (double)xf[g];
      auto sample = [&](Q1Raw x, bool have) {
        int k;
        const bool ok = key_of(x, k);
        bad = bad || (have && !ok);
        if constexpr (KS) {
          // (KS only: no moments)
        } else if constexpr (DTYPE == 1) {
          const int d = have ? k - c : 0;
          const unsigned ud = (unsigned)d;
          is1[g] += d;
          is2[g] += (unsigned long long)(ud * ud);                            // (|d| < 2^16: d^2 is the unsigned 32-bit product)
        } else if constexpr (DTYPE == 0) {
          const double d = have ? (double)(float)x - K : 0.0;
          fs1[g] += d; fs2[g] = __fma_rn(d, d, fs2[g]);
        }
        const unsigned u = (unsigned)(k - base);
        const bool in = u < (unsigned)W;
        if (have && ok && in) __hip_atomic_fetch_add(tbl + (u + ((u >> 5) << 2) + 4u), inc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        const unsigned long long mk = __ballot(have && ok && !in);
        if (mk != 0ull) {                                  // (wave-uniform)
          const unsigned idx = (unsigned)listed + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
          if (have && ok && !in && idx < (unsigned)kCvTail) tail[idx] = ((unsigned)k & 0xffffu) | ((unsigned)g << 16);
          listed += (int)__popcll(mk);
        }
      };
      const int full = n[g] / 256;
      Q4Raw cur = ks_global_load<Q4Raw>(row[g] + min(4 * lane, n[g] - 4));
#pragma unroll 1
      for (int j = 0; j < full; ++j) {
        const Q4Raw nxt = ks_global_load<Q4Raw>(row[g] + min((j + 1) * 256 + 4 * lane, n[g] - 4));
        sample((Q1Raw)cur.x, true); sample((Q1Raw)cur.y, true); sample((Q1Raw)cur.z, true); sample((Q1Raw)cur.w, true);
        cur = nxt;
      }
      Q1Raw rt[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) rt[j] = ks_global_load<Q1Raw>(row[g] + min(full * 256 + j * 64 + lane, n[g] - 1));
#pragma unroll
      for (int j = 0; j < 4; ++j) sample(rt[j], full * 256 + j * 64 + lane < n[g]);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const bool fit = __ballot(bad) == 0ull && listed <= kCvTail;

    if (!fit) {                                            // (wave-uniform) left to the sorting form; the table is cleared for the next position
#pragma unroll
      for (int i = 0; i < 8; ++i) blk[i] = make_uint4(0u, 0u, 0u, 0u);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) work_list[loff + atomicAdd(work_cnt, 1)] = (int)pos;
      continue;
    }

    // ---- the listed samples: lane i holds samples i and i + 64
    const int nt = listed;
    bool tv[2], tlow[2]; unsigned te[2]; int tk[2];
    unsigned low = 0u;                                     // the listed samples below the window, A | B << 16: the scan's carry-in
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      tv[h] = lane + 64 * h < nt;
      te[h] = tv[h] ? tail[lane + 64 * h] : 0u;
      tk[h] = (int)(short)(te[h] & 0xffffu);
      tlow[h] = tk[h] < base;
      low += (unsigned)__popcll(__ballot(tv[h] && tlow[h] && (te[h] >> 16) == 0u)) | ((unsigned)__popcll(__ballot(tv[h] && tlow[h] && (te[h] >> 16) != 0u)) << 16);
    }

    // ---- first walk over the lane's 32 values: A | B << 16 running, the integer KS maximum, the MWU sum, the cubes
    unsigned tot = 0u;
#pragma unroll
    for (int i = 0; i < 8; ++i) { const uint4 q = blk[i]; tot += q.x + q.y + q.z + q.w; }      // (both halves <= 4 095: no carry between them)
    const unsigned run0 = seg_exscan_add_u32<64>(tot, lane) + low;
    const unsigned inwin = (unsigned)__builtin_amdgcn_readlane((int)(run0 + tot), 63) - low;     // the samples inside the window, A | B << 16
    const int n0 = n[0], n1 = n[1];
    unsigned best = 0u, mws = 0u;
    unsigned long long cubes = 0ull;
    [[maybe_unused]] bool any_tie = false;
    auto value = [&](unsigned w, unsigned& run) -> unsigned {                                     // the value's candidate |A n2 - B n1|; run: A | B << 16 at the value
      run += w;
      const int A = (int)(run & 0xffffu), B = (int)(run >> 16);
      const unsigned a = w & 0xffffu, b = w >> 16, t = a + b;
      if constexpr (KS) {
        any_tie = any_tie || t > 1u;                       // (float64 front end: do two float32 images tie anywhere?)
      } else {
        mws += a * (unsigned)(2 * B - (int)b);
        cubes += (unsigned long long)(t * t) * (unsigned long long)t;
      }
      const int x = A * n1 - B * n0;
      return (unsigned)(x < 0 ? -x : x);
    };
    {
      unsigned run = run0;
#pragma unroll 1
      for (int i = 0; i < 8; ++i) {
        const uint4 q = blk[i];
        best = max(best, value(q.x, run)); best = max(best, value(q.y, run)); best = max(best, value(q.z, run)); best = max(best, value(q.w, run));
      }
    }
    // the listed samples, every one against every other: the tail samples below its value and the copies of its value, by group
    unsigned t_run[2] = {0u, 0u}, t_cand[2] = {0u, 0u};     // of a distinct tail value (its first copy's slot): A | B << 16 at the value, its candidate
    bool t_first[2] = {false, false};
    if (nt > 0) {                                          // (wave-uniform)
      unsigned lt[2] = {0u, 0u}, eq[2] = {0u, 0u}; int earlier[2] = {0, 0};
#pragma unroll 1
      for (int j = 0; j < nt; ++j) {
        const unsigned wj = (unsigned)(j < 64 ? __builtin_amdgcn_readlane((int)te[0], j) : __builtin_amdgcn_readlane((int)te[1], j - 64));
        const int kj = (int)(short)(wj & 0xffffu);
        const unsigned one = (wj >> 16) ? 0x10000u : 1u;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          lt[h] += kj < tk[h] ? one : 0u;
          eq[h] += kj == tk[h] ? one : 0u;
          earlier[h] += (kj == tk[h] && j < lane + 64 * h) ? 1 : 0;
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        t_first[h] = tv[h] && earlier[h] == 0;
        unsigned run = lt[h] + (tlow[h] ? 0u : inwin);     // (a value above the window: every sample inside it lies below)
        const unsigned cand = value(t_first[h] ? eq[h] : 0u, run);
        t_run[h] = run; t_cand[h] = t_first[h] ? cand : 0u;
        best = max(best, t_cand[h]);
      }
    }
    best = wave_max_u32(best);

    // ---- second walk: the float form of D where the integer maximum is reached; the table is cleared on the way
    const double dn0 = (double)n0, dn1 = (double)n1;
    const double r0 = kCwRcp.v[n0], r1 = kCwRcp.v[n1];
    double dmax = 0.0;
    {
      unsigned run = run0;
#pragma unroll 1
      for (int i = 0; i < 8; ++i) {
        const uint4 q = blk[i];
        blk[i] = make_uint4(0u, 0u, 0u, 0u);
        const unsigned ww[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          run += ww[e];
          const int A = (int)(run & 0xffffu), B = (int)(run >> 16);
          const int x = A * n1 - B * n0;
          const bool hit = best != 0u && (unsigned)(x < 0 ? -x : x) == best;
          if (__ballot(hit) != 0ull) {
            const double d = fabs(hist_exact_quot(A, dn0, r0) - hist_exact_quot(B, dn1, r1));
            dmax = hit ? fmax(dmax, d) : dmax;
          }
        }
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool hit = t_first[h] && best != 0u && t_cand[h] == best;
        if (__ballot(hit) != 0ull) {
          const double d = fabs(hist_exact_quot((int)(t_run[h] & 0xffffu), dn0, r0) - hist_exact_quot((int)(t_run[h] >> 16), dn1, r1));
          dmax = hit ? fmax(dmax, d) : dmax;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    dmax = wave_max_f64(dmax);
    if constexpr (KS) {
      const bool tie = __ballot(any_tie) != 0ull;
      if (lane == 0) {
        args.ks_num[pos] = best;
        args.ks_d_ref[pos] = dmax;
        if (args.tied) args.tied[pos] = tie ? 1 : 0;
      }
    } else {
      const unsigned MWS = cw_wave_sum_u32(mws);           // <= 2 n1 n2 < 2^26
      const unsigned long long CUBES = wave_sum_u64(cubes);
      const unsigned long long TIE = CUBES - (unsigned long long)(n0 + n1);
      double mean[2] = {0.0, 0.0}, m2[2] = {0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const double dn = g ? dn1 : dn0, rn = g ? r1 : r0;
        if constexpr (DTYPE == 0) {
          const double s1 = wave_sum_f64(fs1[g]), s2 = wave_sum_f64(fs2[g]);
          mean[g] = (double)xf[g] + s1 * rn; m2[g] = s2 - s1 * s1 * rn;
        } else if constexpr (DTYPE == 1) {
          // exact integers: |S1| <= 4 095 * 1 024 + 128 * 2^16, S2 <= 4 095 * 2^20 + 128 * 2^32; n S2 and S1^2 below 2^53
          const double S1 = (double)(int)cw_wave_sum_u32((unsigned)is1[g]), S2 = (double)wave_sum_u64(is2[g]);
          mean[g] = ((double)c + S1 * rn) * 1e-3; m2[g] = __fma_rn(dn, S2, -S1 * S1) * rn * 1e-6;
        }
      }
      if (lane == 0) {
        args.ks_num[pos] = best;
        args.ks_d_ref[pos] = dmax;
        args.mwu_s[pos] = (unsigned long long)MWS;
        args.tie[pos] = TIE;
        if constexpr (DTYPE != 2) {
          double* mo = args.moments + pos * 4;
          mo[0] = mean[0]; mo[1] = m2[0]; mo[2] = mean[1]; mo[3] = m2[1];
        }
        if (args.tied) args.tied[pos] = TIE != 0ull ? 1 : 0;
      }
    }
  }
}

// the classes of index 5 the probe accepted (today: one), laid end to end like rank_count_wide_kernel's
template <int DTYPE, bool KS>
__global__ __launch_bounds__(64 * kWavesPerBlock, 4)
void rank_count_value_kernel(CntWideArgs cw) {
  extern __shared__ __attribute__((aligned(16))) unsigned lds_cv[];
  const RankStatsArgs& args = cw.rs;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* tbl = lds_cv + wave * kCvWaveWords;
  const int64_t wave_global = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  const int64_t nw = (int64_t)gridDim.x * kWavesPerBlock;
  const int nseg = cw.segs[0];
  int64_t rot = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    const int cls = cw.segs[1 + sg];
    if (count_wide_rs_index(cls) != 5 || cw.gates[cls] == 0) continue;
    int64_t count = args.npos, loff = 0;
    const int32_t* list = nullptr;
    if (args.pos_list) { count = args.class_meta[cls]; loff = args.class_meta[kClassStride + cls]; list = args.pos_list + loff; }
    int64_t start = wave_global - rot;
    if (start < 0) start += nw;
    rot = (rot + count) % nw;
    cv_segment<DTYPE, KS>(args, cw.work_list, cw.work_meta + cls, tbl, count, loff, list, start, nw, lane);
  }
}

}  // namespace nmod
