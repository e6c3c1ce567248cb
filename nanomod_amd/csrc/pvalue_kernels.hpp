// K2 (finalize) and K3 (window combine): one thread per genomic position, fp64.
//
// K2 turns K1's exact integers / moments into the (stat, p) pairs getKStest
// returns (myDetect.py:327-343,363) following the scipy 1.2.1 formulas
// (SURVEY.md §8a rows A1-A3).  K3 restates get_combin_pvalue
// (myDetect.py:379-414) over the whole KS track.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "special_math.hpp"
#include "../../include/nanomod_hip.h"

namespace nmod {

struct FinalizeArgs {
  int64_t npos;
  const int64_t* off0; const int64_t* off1;
  int64_t stride0, stride1;
  const uint32_t* ks_num; const uint64_t* mwu_s; const uint64_t* tie; const double* moments;
  const double* ks_d_ref;               // non-null in all-tests mode: the reference's float form of D
  int32_t tests; int32_t want_mstd;
  int64_t max_n0, max_n1;               // per-group limits of this launch
  int64_t min_cap;                      // KS-only: capacity limit of the smaller (sorted) group, else 0
  const uint8_t* nonfinite;             // [npos] or null: 1 where nonfinite_scan_kernel found a NaN / infinite sample (NMOD_FLAG_CHECK_FINITE)
  nmod_out out;
};

__global__ __launch_bounds__(256) void finalize_kernel(FinalizeArgs a) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= a.npos) return;
  const int64_t n0 = a.stride0 > 0 ? a.stride0 : a.off0[p + 1] - a.off0[p];
  const int64_t n1 = a.stride1 > 0 ? a.stride1 : a.off1[p + 1] - a.off1[p];
  const double nan = __builtin_nan("");
  unsigned status = 0;
  const bool too_large = (n0 > a.max_n0 || n1 > a.max_n1) || (a.min_cap > 0 && (n0 < n1 ? n0 : n1) > a.min_cap);
  const bool empty = (n0 <= 0 || n1 <= 0) || too_large;    // "empty": nothing was computed by K1
  if (n0 <= 0 || n1 <= 0) status |= NMOD_STATUS_EMPTY;
  if (too_large) status |= NMOD_STATUS_TOO_LARGE;
  const double dn0 = (double)n0, dn1 = (double)n1;
  const double prod = dn0 * dn1;

  if (a.tests & NMOD_TEST_KS) {
    double d = nan, pv = nan;
    if (!empty) {
      // ks_2samp (scipy 1.2.1): d = max|cdf1 - cdf2|; en = sqrt(n1*n2/float(n1+n2));
      // prob = kstwobign.sf((en + 0.12 + 0.11/en) * d)
      // KS-only mode: the exact rational, correctly rounded (<= 1 ulp from the float-CDF form)
      d = a.ks_d_ref ? a.ks_d_ref[p] : (double)a.ks_num[p] / prod;
      double en = sqrt(prod / (double)(n0 + n1));
      pv = kolmogorov_sf((en + 0.12 + 0.11 / en) * d);
    }
    if (a.out.ks_d) a.out.ks_d[p] = clamp_stat(d);
    if (a.out.ks_p) a.out.ks_p[p] = clamp_p(pv);
  }

  if (a.tests & NMOD_TEST_MWU) {
    double u = nan, pv = nan;
    if (!empty) {
      // mannwhitneyu(x, y, use_continuity=True, alternative=None) of scipy 1.2.1
      double u1 = prod - 0.5 * (double)a.mwu_s[p];      // n1*n2 + n1(n1+1)/2 - sum(rank x)
      double u2 = prod - u1;
      double size = (double)(n0 + n1);
      double T = (size < 2.0) ? 1.0 : 1.0 - (double)a.tie[p] / (size * size * size - size);
      if (T == 0.0) {
        status |= NMOD_STATUS_MWU_ALL_IDENTICAL;          // the reference raises here
      } else {
        double sd = sqrt(T * dn0 * dn1 * (double)(n0 + n1 + 1) / 12.0);
        double meanrank = prod / 2.0 + 0.5;
        double bigu = fmax(u1, u2);
        double z = (bigu - meanrank) / sd;
        pv = norm_sf(fabs(z));
        u = fmin(u1, u2);
      }
    }
    if (a.out.mwu_u) a.out.mwu_u[p] = clamp_stat(u);
    if (a.out.mwu_p) a.out.mwu_p[p] = clamp_p(pv);
  }

  if ((a.tests & NMOD_TEST_WELCH) || a.want_mstd) {
    const double* mo = a.moments + p * 4;
    double mean0 = nan, m20 = nan, mean1 = nan, m21 = nan;
    if (!empty) { mean0 = mo[0]; m20 = mo[1]; mean1 = mo[2]; m21 = mo[3]; }
    // finite samples have finite moments: a NaN or an infinity here is one in the position's samples (NMOD_STATUS_NONFINITE)
    if (!empty && !(fabs(mean0) <= 1.7976931348623157e308 && fabs(mean1) <= 1.7976931348623157e308 &&
                    fabs(m20) <= 1.7976931348623157e308 && fabs(m21) <= 1.7976931348623157e308)) status |= NMOD_STATUS_NONFINITE;
    if (a.tests & NMOD_TEST_WELCH) {
      double t = nan, pv = nan;
      if (!empty) {
        // ttest_ind(equal_var=False): _unequal_var_ttest_denom + _ttest_finish
        // (vn = var / n with one division each, df with one: the products of the small integers are exact)
        const double e0 = dn0 - 1.0, e1 = dn1 - 1.0;
        double vn1 = m20 / (e0 * dn0), vn2 = m21 / (e1 * dn1);
        double df = (vn1 + vn2) * (vn1 + vn2) * (e0 * e1) / (vn1 * vn1 * e1 + vn2 * vn2 * e0);
        if (df != df) df = 1.0;
        double denom = sqrt(vn1 + vn2);
        t = (mean0 - mean1) / denom;
        pv = student_t_two_sided(t, df);
      }
      if (pv != pv) status |= NMOD_STATUS_T_NAN;
      if (a.out.t_t) a.out.t_t[p] = clamp_stat(t);
      if (a.out.t_p) a.out.t_p[p] = clamp_p(pv);
    }
    if (a.want_mstd) {
      if (a.out.mean0) a.out.mean0[p] = mean0;
      if (a.out.std0) a.out.std0[p] = sqrt(m20 / dn0);      // np.std, ddof = 0
      if (a.out.mean1) a.out.mean1[p] = mean1;
      if (a.out.std1) a.out.std1[p] = sqrt(m21 / dn1);
    }
  }
  if (a.nonfinite && !empty && a.nonfinite[p]) status |= NMOD_STATUS_NONFINITE;
  if (a.out.status) a.out.status[p] = (uint8_t)status;
}

// NMOD_FLAG_CHECK_FINITE: one wave per position reads both rows and flags a NaN / infinite sample (float32 or float64 rows)
struct NonfiniteArgs {
  const void* sig0; const void* sig1; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1; int64_t npos; int32_t f64;
  int64_t lim0, lim1; uint8_t* flag;
};
__global__ __launch_bounds__(256) void nonfinite_scan_kernel(NonfiniteArgs a) {
  const int lane = threadIdx.x & 63;
  for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < a.npos; p += (int64_t)gridDim.x * 4) {
    bool bad = false;
    for (int g = 0; g < 2; ++g) {
      const int64_t st = g ? a.stride1 : a.stride0;
      const int64_t* off = g ? a.off1 : a.off0;
      const int64_t o = st > 0 ? p * st : off[p];
      int64_t n = st > 0 ? st : off[p + 1] - o;
      if (n > (g ? a.lim1 : a.lim0)) n = 0;                       // (a position beyond the limits is skipped everywhere: NMOD_STATUS_TOO_LARGE)
      const void* sig = g ? a.sig1 : a.sig0;
      for (int64_t i = lane; i < n; i += 64) {
        if (a.f64) bad = bad || !(fabs(reinterpret_cast<const double*>(sig)[o + i]) <= 1.7976931348623157e308);
        else bad = bad || !(fabsf(reinterpret_cast<const float*>(sig)[o + i]) <= 3.4028234663852886e38f);
      }
    }
    const bool any = __ballot(bad) != 0ull;
    if (lane == 0) a.flag[p] = any ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
constexpr int kCombineTile = 256;

struct CombineArgs {
  int64_t npos;
  const double* ks_d; const double* ks_p; const int32_t* run_id;
  double* comb_st; double* comb_p;
  int32_t nb; int32_t method;
  double wnorm;                         // ||w||_2
  double w[2 * NMOD_MAX_NB + 1];        // Stouffer weights, w[nb] = 100 (myDetect.py:396-400)
};

__global__ __launch_bounds__(kCombineTile) void combine_kernel(CombineArgs a) {
  // per-position transform of the KS p-value (z = isf(p) or ln p), tile + halo
  __shared__ double tr[kCombineTile + 2 * NMOD_MAX_NB];
  const int nb = a.nb;
  const int64_t tile0 = (int64_t)blockIdx.x * kCombineTile;
  const int t = threadIdx.x;
  const bool stouffer = a.method == NMOD_METHOD_STOUFFER;
  auto transform = [&](int64_t j) -> double {
    double pj = a.ks_p[j];
    return stouffer ? norm_isf(pj) : log(pj);
  };
  const int64_t p = tile0 + t;
  if (nb == 0) {                                  // myDetect.py:413: the KS tuple itself
    if (p < a.npos) { a.comb_st[p] = a.ks_d[p]; a.comb_p[p] = a.ks_p[p]; }
    return;
  }
  if (p < a.npos) tr[nb + t] = transform(p);
  if (t < 2 * nb) {
    int64_t j = (t < nb) ? tile0 - nb + t : tile0 + kCombineTile + (t - nb);
    int slot = (t < nb) ? t : kCombineTile + t;
    if (j >= 0 && j < a.npos) tr[slot] = transform(j);
  }
  __syncthreads();
  if (p >= a.npos) return;
  const int32_t rid = a.run_id[p];
  // window = [p_KS[j] if usable else 1.0]  (myDetect.py:383-389); isf(1) = -inf, ln(1) = 0
  const double pad = stouffer ? -__builtin_inf() : 0.0;
  double acc = 0.0;
  for (int k = -nb; k <= nb; ++k) {
    int64_t j = p + k;
    bool ok = (j >= 0) && (j < a.npos);
    if (ok) ok = (a.run_id[j] == rid);
    double v = ok ? tr[nb + t + k] : pad;
    acc += stouffer ? a.w[nb + k] * v : v;
  }
  double st, pv;
  if (stouffer) {
    st = acc / a.wnorm;                 // Z = dot(w, Zi) / ||w||
    pv = norm_sf(st);
  } else {
    st = -2.0 * acc;                    // Xsq = -2 sum(log p)
    pv = chi2_sf_even(st, 2 * nb + 1);
  }
  a.comb_p[p] = clamp_p(pv);
  a.comb_st[p] = clamp_stat(st);
}

// ---------------------------------------------------------------------------
// K5 synthetic two-group signal generator (include/nanomod_hip.h: nmod_synth_fill)
__device__ __forceinline__ uint64_t synth_mix(uint64_t seed, int64_t pos, int32_t group, uint32_t read) {
  uint64_t x = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(pos * 2 + group);
  x ^= (uint64_t)read * 0xD1B54A32D192ED03ull;
  x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27; x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}

struct SynthArgs {
  uint64_t seed; int64_t pos_begin; int64_t npos; int32_t group; int32_t n_per_pos;
  int64_t plant_period; float plant_shift; int32_t dtype; void* out;
};

__global__ __launch_bounds__(256) void synth_kernel(SynthArgs a) {
  const int64_t total = a.npos * (int64_t)a.n_per_pos;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    int64_t rel = idx / a.n_per_pos;
    uint32_t read = (uint32_t)(idx - rel * a.n_per_pos);
    int64_t pos = a.pos_begin + rel;
    uint64_t h = synth_mix(a.seed, pos, a.group, read);
    int s = (int)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48));
    float x = __fmul_rn((float)(s - 131070), 2.6428997e-05f);
    asm volatile("" : "+v"(x));                         // (rounded here: never contracted with the planted shift)
    if (a.group == 1 && a.plant_period > 0) {
      int64_t m = pos % a.plant_period;
      if (m == 0 || m == 1 || m == a.plant_period - 1) x = __fadd_rn(x, a.plant_shift);
    }
    if (a.dtype == NMOD_DTYPE_F32) reinterpret_cast<float*>(a.out)[idx] = x;
    else reinterpret_cast<int16_t*>(a.out)[idx] = (int16_t)rintf(__fmul_rn(x, 1000.0f));
  }
}

// the same values for ragged rows: sample `read` of position pos_begin + i goes to sig[off[i] + read]; one wave per row
struct SynthCsrArgs {
  uint64_t seed; int64_t pos_begin; int64_t npos; int32_t group; int32_t dtype;
  int64_t plant_period; float plant_shift; const int64_t* off; void* out;
};

__global__ __launch_bounds__(256) void synth_csr_kernel(SynthCsrArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t rel = wave; rel < a.npos; rel += (int64_t)gridDim.x * 4) {
    const int64_t o = a.off[rel];
    const int n = (int)(a.off[rel + 1] - o);
    const int64_t pos = a.pos_begin + rel;
    bool planted = false;
    if (a.group == 1 && a.plant_period > 0) {
      const int64_t m = pos % a.plant_period;
      planted = (m == 0 || m == 1 || m == a.plant_period - 1);
    }
    for (int read = lane; read < n; read += 64) {
      const uint64_t h = synth_mix(a.seed, pos, a.group, (uint32_t)read);
      const int s = (int)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48));
      float x = __fmul_rn((float)(s - 131070), 2.6428997e-05f);
      asm volatile("" : "+v"(x));                       // (the product is rounded before the shift is added: no fused multiply-add)
      if (planted) x = __fadd_rn(x, a.plant_shift);
      if (a.dtype == NMOD_DTYPE_F32) reinterpret_cast<float*>(a.out)[o + read] = x;
      else reinterpret_cast<int16_t*>(a.out)[o + read] = (int16_t)rintf(__fmul_rn(x, 1000.0f));
    }
  }
}

// ---------------------------------------------------------------------------
// K5b synthetic EVENT rows (include/nanomod_hip.h: nmod_synth_fill_events): a level per position (both groups share it)
// plus a per-read spread, on the milli-unit grid of stored NanoMod events.  Integer-only up to the final quotient:
//   level(pos) = (mix(seed ^ kLevelSalt, pos, 0, 0) >> 40) % 6001 - 3000                       milli-units, +-3 units
//   k = level + floor((2 z spread_milli + 37837) / 75674)  [+ shift_milli: planted positions of group 1],  z = s - 131070
// (z has standard deviation 37837.2: k - level is z spread / sd rounded to the nearest milli-unit), clamped to int16;
// float32 output is the float64 quotient k / 1000.0 rounded to float32 — what a stored 3-decimal event value is.
constexpr uint64_t kLevelSalt = 0xA5A5A5A5DEADBEEFull;

__device__ __forceinline__ int synth_event_level(uint64_t seed, int64_t pos) {
  return (int)((synth_mix(seed ^ kLevelSalt, pos, 0, 0u) >> 40) % 6001ull) - 3000;
}
constexpr uint64_t kOutlierSalt = 0x0DDBA11C0FFEE123ull;
// (the quotient's floor in 64 bits: 2 z spread reaches 2.1e9 at spread 8 000)
__device__ __forceinline__ int synth_event_key(uint64_t seed, int64_t pos, int32_t group, uint32_t read, int level, int spread_milli, int shift,
                                               int outlier_permille) {
  const uint64_t h = synth_mix(seed, pos, group, read);
  const int z = (int)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + (h >> 48)) - 131070;
  constexpr long long kBias = 32768;                             // (floor division through a non-negative numerator)
  const int q = (int)((2ll * z * spread_milli + 37837 + kBias * 75674) / 75674 - kBias);
  int k = level + q + shift;
  if (outlier_permille > 0) {
    // a read in `outlier_permille` of 1 000 is a mis-segmented event: a uniform draw over the +-5 unit clip range of the raw
    // normalisation (myRefBaseSignalAnnotation.py:251-259), whatever the position's level
    const uint64_t o = synth_mix(seed ^ kOutlierSalt, pos, group, read);
    if ((int)((o >> 20) % 1000ull) < outlier_permille) k = (int)((o >> 32) % 10001ull) - 5000;
  }
  return min(max(k, -32767), 32767);
}

struct SynthEventArgs {
  uint64_t seed; int64_t pos_begin; int64_t npos; int32_t group; int32_t n_per_pos; const int64_t* off;
  int64_t plant_period; int32_t plant_shift_milli; int32_t spread_milli; int32_t dtype; int32_t outlier_permille; void* out;
};

// one wave per row (fixed stride or CSR)
__global__ __launch_bounds__(256) void synth_event_kernel(SynthEventArgs a) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t rel = wave; rel < a.npos; rel += (int64_t)gridDim.x * 4) {
    const int64_t o = a.n_per_pos > 0 ? rel * (int64_t)a.n_per_pos : a.off[rel];
    const int n = a.n_per_pos > 0 ? a.n_per_pos : (int)(a.off[rel + 1] - o);
    const int64_t pos = a.pos_begin + rel;
    int shift = 0;
    if (a.group == 1 && a.plant_period > 0) {
      const int64_t m = pos % a.plant_period;
      if (m == 0 || m == 1 || m == a.plant_period - 1) shift = a.plant_shift_milli;
    }
    const int level = synth_event_level(a.seed, pos);
    for (int read = lane; read < n; read += 64) {
      const int k = synth_event_key(a.seed, pos, a.group, (uint32_t)read, level, a.spread_milli, shift, a.outlier_permille);
      if (a.dtype == NMOD_DTYPE_F32) reinterpret_cast<float*>(a.out)[o + read] = (float)((double)k / 1000.0);
      else reinterpret_cast<int16_t*>(a.out)[o + read] = (int16_t)k;
    }
  }
}

}  // namespace nmod
