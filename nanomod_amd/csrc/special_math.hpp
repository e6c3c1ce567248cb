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

// norm.isf(p) = -ndtri(p).  Newton on log Q(z) - log p, started from
// Abramowitz-Stegun 26.2.23; Q is evaluated through erfcx so the iteration is
// well conditioned out to p = DBL_MIN (z = 37.52).
__device__ inline double norm_isf(double p) {
  if (p != p) return p;
  if (p <= 0.0) return __builtin_inf();
  if (p >= 1.0) return -__builtin_inf();
  bool flip = p > 0.5;
  double q = flip ? 1.0 - p : p;          // exact for p in (0.5, 1)
  double lq = log(q);
  double t = sqrt(-2.0 * lq);
  double z = t - (2.515517 + t * (0.802853 + t * 0.010328)) /
                     (1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
  if (z < 0.0) z = 0.0;
#pragma unroll 1
  for (int it = 0; it < 3; ++it) {
    double u = z * kInvSqrt2;
    double ex = 0.5 * erfcx(u);           // Q(z) = ex * exp(-u^2)
    double lQ = log(ex) - u * u;
    z += (lQ - lq) * ex * kSqrt2Pi;       // Q / phi = ex * sqrt(2 pi)
    if (z < 0.0) z = 0.0;
  }
  return flip ? -z : z;
}

// log B(a, 1/2) without the lgamma cancellation: shift a up to >= 16 with the
// recurrence, then Stirling differences.
__device__ inline double lbeta_half(double a) {
  double ratio = 1.0;                     // prod (a+i)/(a+i+1/2)
  double z = a;
#pragma unroll 1
  while (z < 16.0) { ratio *= z / (z + 0.5); z += 1.0; }
  auto S = [](double w) {
    double w2 = 1.0 / (w * w);
    return (1.0 / w) * (1.0 / 12.0 + w2 * (-1.0 / 360.0 + w2 * (1.0 / 1260.0 + w2 * (-1.0 / 1680.0 + w2 * (1.0 / 1188.0)))));
  };
  // lgamma(z+1/2) - lgamma(z)
  double d = z * log1p(0.5 / z) + 0.5 * log(z) - 0.5 + S(z + 0.5) - S(z);
  // lgamma(a+1/2) - lgamma(a) = d + log(ratio)
  return 0.57236494292470009 /* log(pi)/2 */ - d - log(ratio);
}

// Continued fraction of the regularised incomplete beta (modified Lentz).
__device__ inline double betacf(double a, double b, double x) {
  const double tiny = 1e-300;
  double qab = a + b, qap = a + 1.0, qam = a - 1.0;
  double c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
#pragma unroll 1
  for (int m = 1; m <= 2000; ++m) {
    double m2 = 2.0 * m;
    double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c;
    h *= del;
    if (fabs(del - 1.0) < 2e-16) break;
  }
  return h;
}

// 2 * t.sf(|t|, df) = I_{df/(df+t^2)}(df/2, 1/2)   (ttest_ind's _ttest_finish)
__device__ inline double student_t_two_sided(double t, double df) {
  if (t != t || df != df) return __builtin_nan("");
  double t2 = t * t;
  if (t2 == 0.0) return 1.0;
  if (isinf(t2)) return 0.0;
  double a = 0.5 * df, b = 0.5;
  double r = t2 / df;
  double x = 1.0 / (1.0 + r);             // df/(df+t^2)
  double y = r / (1.0 + r);               // 1 - x, without cancellation
  double lnx = -log1p(r);
  double lny = log(y);
  double front = exp(a * lnx + b * lny - lbeta_half(a));
  if (x < (a + 1.0) / (a + b + 2.0)) return front * betacf(a, b, x) / a;
  return 1.0 - front * betacf(b, a, y) / b;
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
