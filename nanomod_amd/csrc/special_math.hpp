// fp64 tail functions for the p-values of the per-base tests (device side).
//
// Each function states the scipy 1.2.1 primitive it stands for (the reference
// reaches them through scipy.stats at myDetect.py:331,335,341,393,401).  They
// are written for gfx950 only (ocml exp/log/erfc/erfcx/lgamma/log1p in fp64);
// relative accuracy target 1e-12 down to p = DBL_MIN.
#pragma once
#include <hip/hip_runtime.h>

namespace nmod {

constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kDblMax = 1.7976931348623157e+308;
constexpr double kSqrt2Pi = 2.5066282746310002;
constexpr double kInvSqrt2 = 0.70710678118654752;

// m_min_float / m_max_float (myDetect.py:317-325); NaN passes through both.
__device__ __forceinline__ double clamp_p(double p) { return (p < kDblMin) ? kDblMin : p; }
__device__ __forceinline__ double clamp_stat(double s) { return (s > kDblMax) ? kDblMax : s; }

// kstwobign.sf == scipy.special.kolmogorov: Q(x) = 2 sum_{k>=1} (-1)^(k-1) exp(-2 k^2 x^2).
// Small x uses the theta-function dual 1 - sqrt(2 pi)/x sum exp(-(2k-1)^2 pi^2/(8 x^2)),
// which is what keeps the value accurate where the alternating series cancels.
__device__ inline double kolmogorov_sf(double x) {
  if (!(x > 0.0)) return (x != x) ? x : 1.0;
  if (x < 0.82) {
    const double pi2_8 = 1.2337005501361697;  // pi^2/8
    double w = pi2_8 / (x * x);
    double s = exp(-w) + exp(-9.0 * w) + exp(-25.0 * w) + exp(-49.0 * w);
    return 1.0 - kSqrt2Pi / x * s;
  }
  double q = -2.0 * x * x;
  double s = 0.0, sign = 1.0;
#pragma unroll 1
  for (int k = 1; k <= 8; ++k) {
    double t = exp(q * (double)(k * k));
    s += sign * t;
    sign = -sign;
    if (t < 1e-18 * s) break;
  }
  return 2.0 * s;
}

// norm.sf(z) = ndtr(-z)
__device__ __forceinline__ double norm_sf(double z) { return 0.5 * erfc(z * kInvSqrt2); }

// norm.isf(p) = -ndtri(p).  Acklam's rational approximation (relative error 1.2e-9 over the whole double
// range) followed by ONE Newton step on log Q(z) - log p; Q is evaluated through erfcx so the step is well
// conditioned out to p = DBL_MIN (z = 37.52).  Checked against ndtri on 3e5 points from 1e-307 to 0.5: 7e-13.
__device__ inline double norm_isf(double p) {
  if (p != p) return p;
  if (p <= 0.0) return __builtin_inf();
  if (p >= 1.0) return -__builtin_inf();
  const bool flip = p > 0.5;
  const double q = flip ? 1.0 - p : p;          // exact for p in (0.5, 1)
  const double lq = log(q);
  double z;
  if (q < 0.02425) {
    const double t = sqrt(-2.0 * lq);
    z = -(((((-7.784894002430293e-03 * t - 3.223964580411365e-01) * t - 2.400758277161838e+00) * t - 2.549732539343734e+00) * t
            + 4.374664141464968e+00) * t + 2.938163982698783e+00) /
         ((((7.784695709041462e-03 * t + 3.224671290700398e-01) * t + 2.445134137142996e+00) * t + 3.754408661907416e+00) * t + 1.0);
  } else {
    const double c = q - 0.5, r = c * c;
    z = -(((((-3.969683028665376e+01 * r + 2.209460984245205e+02) * r - 2.759285104469687e+02) * r + 1.383577518672690e+02) * r
            - 3.066479806614716e+01) * r + 2.506628277459239e+00) * c /
         (((((-5.447609879822406e+01 * r + 1.615858368580409e+02) * r - 1.556989798598866e+02) * r + 6.680131188771972e+01) * r
            - 1.328068155288572e+01) * r + 1.0);
  }
  if (z < 0.0) z = 0.0;
  {
    const double u = z * kInvSqrt2;
    const double ex = 0.5 * erfcx(u);           // Q(z) = ex * exp(-u^2)
    const double lQ = log(ex) - u * u;
    z += (lQ - lq) * ex * kSqrt2Pi;             // Q / phi = ex * sqrt(2 pi)
    if (z < 0.0) z = 0.0;
  }
  return flip ? -z : z;
}

// log B(a, 1/2) without the lgamma cancellation: shift a up to z >= 16 with the recurrence, then the asymptotic series
//   lgamma(z + 1/2) - lgamma(z) = 1/2 ln z + sum_{n even} (2^(1-n) - 2) B_n / (n (n - 1) z^(n-1))
//                               = 1/2 ln z - 1/(8 z) + 1/(192 z^3) - 1/(640 z^5) + 17/(14336 z^7) - 31/(18432 z^9) + 691/(180224 z^11)
// (next term 3e-18 at z = 16; against mpmath's log(beta(a, 1/2)) on 0.5 <= a <= 1e7: 7.3e-16 absolute).  One division
// and one logarithm; the Stirling-difference form before it took four divisions, log1p and two logarithms.
__device__ inline double lbeta_half(double a) {
  double ratio = 1.0;                     // prod (a+i)/(a+i+1/2)
  double z = a;
  bool shifted = false;
#pragma unroll 1
  while (z < 16.0) { ratio *= z / (z + 0.5); z += 1.0; shifted = true; }
  const double r = 1.0 / z, r2 = r * r;
  const double d = 0.5 * log(z) + r * (-1.0 / 8.0 + r2 * (1.0 / 192.0 + r2 * (-1.0 / 640.0 + r2 * (17.0 / 14336.0
                   + r2 * (-31.0 / 18432.0 + r2 * (691.0 / 180224.0))))));
  // lgamma(a+1/2) - lgamma(a) = d + log(ratio)
  double lr = 0.0;
  if (__ballot(shifted) != 0ull) lr = log(ratio);
  return 0.57236494292470009 /* log(pi)/2 */ - d - lr;
}

// Continued fraction of the regularised incomplete beta, 1/(1 + d_1/(1 + d_2/(1 + ...))) with
//   d_{2m+1} = -(a+m)(a+b+m) x / ((a+2m)(a+2m+1)),   d_{2m} = m (b-m) x / ((a+2m-1)(a+2m)).
// Evaluated by the forward (Wallis) recurrence on an equivalent fraction without divisions: with d_k = n_k/e_k
// and c_k = e_k rho_k, rho_k any approximation of 1/e_k (v_rcp_f64), the fraction with partial numerators
// rho_k c_{k-1} n_k and partial denominators c_k has the same convergents and stays O(1) in magnitude.
// (The modified-Lentz form costs six fp64 divisions per m; this one none: finalize_kernel 1.0 -> 0.4 ms.)
__device__ inline double betacf(double a, double b, double x) {
  const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double A0 = 0.0, B0 = 1.0, A1 = 1.0, B1 = 1.0, cprev = 1.0;
  auto step = [&](double n, double e) {
    const double rho = __builtin_amdgcn_rcp(e);
    const double c = e * rho;
    const double t = rho * cprev * n;
    const double A2 = c * A1 + t * A0, B2 = c * B1 + t * B0;
    A0 = A1; B0 = B1; A1 = A2; B1 = B2; cprev = c;
  };
  step(-qab * x, qap);                                            // d_1
#pragma unroll 1
  for (int m = 1; m <= 2000; ++m) {
    const double dm = (double)m, m2 = 2.0 * dm;
    step(dm * (b - dm) * x, (qam + m2) * (a + m2));               // d_{2m}
    step(-(a + dm) * (qab + dm) * x, (a + m2) * (qap + m2));      // d_{2m+1}
    const double u = A1 * B0, v = A0 * B1;                         // successive convergents agree to 2e-16 (relative)
    if (fabs(u - v) < 2e-16 * fabs(u)) break;
    // the convergents are ratios: rescale all four terms when they drift (x near 1 with large a shrinks them by
    // ~(1 - x) per m and the fraction needs hundreds of terms)
    const double mag = fabs(B1);
    if (mag < 1e-60 || mag > 1e60) {
      const double sc = __builtin_amdgcn_rcp(fmax(mag, 1e-300));
      A0 *= sc; B0 *= sc; A1 *= sc; B1 *= sc;
    }
#if defined(NMOD_EXP) && (NMOD_EXP & 8)
    break;
#endif
  }
  const double tiny = 1e-300;
  if (fabs(B1) < tiny) B1 = tiny;
  return A1 / B1;
}

// 1 / (k + 3/2), k = 0 .. 127 (correctly rounded): the coefficient ratios of the series below
__device__ __constant__ double kInvKPlus15[128] = {
  0.66666666666666663, 0.40000000000000002, 0.2857142857142857, 0.22222222222222221,
  0.18181818181818182, 0.15384615384615385, 0.13333333333333333, 0.11764705882352941,
  0.10526315789473684, 0.095238095238095233, 0.086956521739130432, 0.080000000000000002,
  0.07407407407407407, 0.068965517241379309, 0.064516129032258063, 0.060606060606060608,
  0.057142857142857141, 0.054054054054054057, 0.05128205128205128, 0.04878048780487805,
  0.046511627906976744, 0.044444444444444446, 0.042553191489361701, 0.040816326530612242,
  0.039215686274509803, 0.037735849056603772, 0.036363636363636362, 0.035087719298245612,
  0.033898305084745763, 0.032786885245901641, 0.031746031746031744, 0.030769230769230771,
  0.029850746268656716, 0.028985507246376812, 0.028169014084507043, 0.027397260273972601,
  0.026666666666666668, 0.025974025974025976, 0.025316455696202531, 0.024691358024691357,
  0.024096385542168676, 0.023529411764705882, 0.022988505747126436, 0.02247191011235955,
  0.02197802197802198, 0.021505376344086023, 0.021052631578947368, 0.020618556701030927,
  0.020202020202020204, 0.019801980198019802, 0.019417475728155338, 0.019047619047619049,
  0.018691588785046728, 0.01834862385321101, 0.018018018018018018, 0.017699115044247787,
  0.017391304347826087, 0.017094017094017096, 0.01680672268907563, 0.016528925619834711,
  0.016260162601626018, 0.016, 0.015748031496062992, 0.015503875968992248,
  0.015267175572519083, 0.015037593984962405, 0.014814814814814815, 0.014598540145985401,
  0.014388489208633094, 0.014184397163120567, 0.013986013986013986, 0.013793103448275862,
  0.013605442176870748, 0.013422818791946308, 0.013245033112582781, 0.013071895424836602,
  0.012903225806451613, 0.012738853503184714, 0.012578616352201259, 0.012422360248447204,
  0.012269938650306749, 0.012121212121212121, 0.011976047904191617, 0.011834319526627219,
  0.011695906432748537, 0.011560693641618497, 0.011428571428571429, 0.011299435028248588,
  0.0111731843575419, 0.011049723756906077, 0.01092896174863388, 0.010810810810810811,
  0.0106951871657754, 0.010582010582010581, 0.010471204188481676, 0.010362694300518135,
  0.010256410256410256, 0.01015228426395939, 0.010050251256281407, 0.0099502487562189053,
  0.009852216748768473, 0.0097560975609756097, 0.0096618357487922701, 0.0095693779904306216,
  0.0094786729857819912, 0.0093896713615023476, 0.0093023255813953487, 0.0092165898617511521,
  0.0091324200913242004, 0.0090497737556561094, 0.0089686098654708519, 0.0088888888888888889,
  0.0088105726872246704, 0.0087336244541484712, 0.008658008658008658, 0.0085836909871244635,
  0.0085106382978723406, 0.0084388185654008432, 0.008368200836820083, 0.0082987551867219917,
  0.00823045267489712, 0.0081632653061224497, 0.0080971659919028341, 0.0080321285140562242,
  0.0079681274900398405, 0.0079051383399209481, 0.0078431372549019607, 0.0077821011673151752,
};

// 2 * t.sf(|t|, df) = I_{df/(df+t^2)}(df/2, 1/2)   (ttest_ind's _ttest_finish)
__device__ inline double student_t_two_sided(double t, double df) {
  if (t != t || df != df) return __builtin_nan("");
  double t2 = t * t;
  if (t2 == 0.0) return 1.0;
  if (isinf(t2)) return 0.0;
  double a = 0.5 * df, b = 0.5;
  const double inv = 1.0 / (df + t2);     // (one division for both)
  double x = df * inv;                    // df/(df+t^2)
  double y = t2 * inv;                    // 1 - x, without cancellation
  double lnx = log1p(-y);
  double front = sqrt(y) * exp(a * lnx - lbeta_half(a));       // x^a y^(1/2) / B(a, 1/2)
  // Moderate |t| (t^2 < 9: p > 2.7e-3, so 1 - v keeps 12 digits) and y < 0.3: the hypergeometric series
  //   I_y(1/2, a) = 2 front * sum_k [(a + 1/2)_k / (3/2)_k] y^k      (DLMF 8.17.8; all terms positive)
  // whose term ratio y (a + 1/2 + k) / (3/2 + k) falls below 1/2 within a few terms: 6 fp64 operations per term against
  // ~35 per step of the continued fraction, and nearly every position of a null-like batch is in this range.
  const bool fast = t2 < 9.0 && y < 0.3;
  double p_fast = 0.0;
  bool done = false;
  if (__ballot(fast) != 0ull) {
    double term = 1.0, sum = 1.0;
    const double ah = a + 0.5;
    bool conv = false;
#pragma unroll 1
    for (int k = 0; k < 128; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        term *= y * (ah + (double)(k + j)) * kInvKPlus15[k + j];
        sum += term;
      }
      conv = !(term > 1e-17 * sum);                      // (the ratio is decreasing in k: once below this, the tail is too)
      if (__ballot(fast && !conv) == 0ull) break;
    }
    p_fast = 1.0 - 2.0 * front * sum;
    done = fast && conv;
  }
  if (__ballot(!done) == 0ull) return p_fast;
  // one call for both regimes: a wave whose lanes fall on both sides would otherwise run the fraction twice
  const bool direct = x < (a + 1.0) / (a + b + 2.0);
  const double ca = direct ? a : b, cb = direct ? b : a, cx = direct ? x : y;
  const double v = front * betacf(ca, cb, cx) / ca;
  const double p_cf = direct ? v : 1.0 - v;
  return done ? p_fast : p_cf;
}

// chi2.sf(X, 2W) = exp(-x) sum_{m<W} x^m/m!,  x = X/2   (combine_pvalues, fisher)
__device__ inline double chi2_sf_even(double X, int W) {
  double x = 0.5 * X;
  if (x != x) return x;
  if (!(x > 0.0)) return 1.0;
  if (isinf(x)) return 0.0;
  int ms = W - 1;
  if ((double)ms > x) ms = (int)x;
  double lt = -x + (double)ms * log(x) - lgamma((double)ms + 1.0);
  double t = exp(lt);
  double sum = t, tm = t;
#pragma unroll 1
  for (int m = ms; m >= 1; --m) { tm *= (double)m / x; sum += tm; if (tm < 1e-18 * sum) break; }
  tm = t;
#pragma unroll 1
  for (int m = ms + 1; m <= W - 1; ++m) { tm *= x / (double)m; sum += tm; }
  return sum > 1.0 ? 1.0 : sum;
}

}  // namespace nmod
