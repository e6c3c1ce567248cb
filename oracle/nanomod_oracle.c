/* nanomod_oracle.c — plain-C CPU restatement of NanoMod's per-base two-sample
 * testing hot path.  TEST INFRASTRUCTURE ONLY: used by tests/ (as the checker
 * at sizes the Python oracle is too slow for) and by bench.py's cpu_baseline
 * leg (kind "port").  The product (nanomod_amd/) never links or loads it.
 *
 * It follows the same reference lines as oracle/nanomod_oracle.py:
 *   getKStest            myDetect.py:327-343,363  (default branch)
 *   m_min/m_max_float    myDetect.py:317-325
 *   get_combin_pvalue    myDetect.py:379-414, pos_check :366-371
 * and the scipy 1.2.1 algorithms those lines call (scipy/stats/stats.py at tag
 * v1.2.1: ks_2samp, mannwhitneyu(alternative=None), ttest_ind(equal_var=False),
 * combine_pvalues; rankdata/tiecorrect).  scipy itself is absent from
 * /root/reference; the special functions are written here on libm
 * (erfc, exp, log, lgamma) and are pinned by tests/test_oracle_c.py against
 * oracle/nanomod_oracle.py (scipy.special primitives) and the golden fixtures.
 * Pinning status: see the header of oracle/nanomod_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_METHOD_KS 0
#define ORC_METHOD_STOUFFER 1
#define ORC_METHOD_FISHER 2
#define ORC_STATUS_MWU_ALL_IDENTICAL 1
#define ORC_STATUS_T_NAN 2
#define ORC_STATUS_EMPTY 4

/* myDetect.py:317-325 (NaN passes through both) */
static double m_min_float(double v) { return (v < DBL_MIN) ? DBL_MIN : v; }
static double m_max_float(double v) { return (v > DBL_MAX) ? DBL_MAX : v; }

/* ---- special functions ------------------------------------------------ */
/* scipy.special.kolmogorov */
static double kolmogorov_sf(double x) {
  if (isnan(x)) return x;
  if (x <= 0.0) return 1.0;
  if (x < 0.82) {
    double w = M_PI * M_PI / (8.0 * x * x), s = 0.0;
    for (int k = 1; k <= 7; k += 2) s += exp(-(double)(k * k) * w);
    return 1.0 - sqrt(2.0 * M_PI) / x * s;
  }
  double s = 0.0, sign = 1.0;
  for (int k = 1; k <= 10; ++k) {
    double t = exp(-2.0 * (double)(k * k) * x * x);
    s += sign * t; sign = -sign;
    if (t < 1e-18 * s) break;
  }
  return 2.0 * s;
}

static double norm_sf(double z) { return 0.5 * erfc(z / M_SQRT2); }

/* norm.isf(p): bisection-free Newton on Q(z) - p in log space, libm only */
static double norm_isf(double p) {
  if (isnan(p)) return p;
  if (p <= 0.0) return INFINITY;
  if (p >= 1.0) return -INFINITY;
  int flip = p > 0.5;
  double q = flip ? 1.0 - p : p;
  double lq = log(q);
  double t = sqrt(-2.0 * lq);
  double z = t - (2.515517 + t * (0.802853 + t * 0.010328)) / (1.0 + t * (1.432788 + t * (0.189269 + t * 0.001308)));
  if (z < 0) z = 0;
  for (int it = 0; it < 4; ++it) {
    double Q = norm_sf(z);
    double phi = exp(-0.5 * z * z) / sqrt(2.0 * M_PI);
    z += (log(Q) - lq) * Q / phi;
    if (z < 0) z = 0;
  }
  return flip ? -z : z;
}

/* regularised incomplete beta I_x(a, b), continued fraction (Lentz) */
static double betacf(double a, double b, double x) {
  const double tiny = 1e-300;
  double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
  if (fabs(d) < tiny) d = tiny;
  d = 1.0 / d;
  double h = d;
  for (int m = 1; m <= 5000; ++m) {
    double m2 = 2.0 * m, aa = m * (b - m) * x / ((qam + m2) * (a + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d; h *= d * c;
    aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
    d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
    c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
    d = 1.0 / d;
    double del = d * c; h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return h;
}

/* log B(a, 1/2) = lgamma(a) + lgamma(1/2) - lgamma(a + 1/2), via log1p to avoid the cancellation */
static double lbeta_half(double a) {
  double ratio = 0.0, z = a;
  while (z < 20.0) { ratio += log1p(-0.5 / (z + 0.5)); z += 1.0; }   /* log prod z/(z+1/2) */
  double s = 0.0;
  const double c[6] = {1.0 / 12, -1.0 / 360, 1.0 / 1260, -1.0 / 1680, 1.0 / 1188, -691.0 / 360360};
  double w1 = z + 0.5, w0 = z, p1 = 1.0 / w1, p0 = 1.0 / w0;
  for (int k = 0; k < 6; ++k) { s += c[k] * (p1 - p0); p1 /= w1 * w1; p0 /= w0 * w0; }
  double d = z * log1p(0.5 / z) + 0.5 * log(z) - 0.5 + s;   /* lgamma(z+1/2) - lgamma(z) */
  return 0.5 * log(M_PI) - d - ratio;
}

/* 2 * t.sf(|t|, df)  ==  stdtr(df, -|t|) * 2 */
static double student_t_two_sided(double t, double df) {
  if (isnan(t) || isnan(df)) return NAN;
  double t2 = t * t;
  if (t2 == 0.0) return 1.0;
  if (isinf(t2)) return 0.0;
  double a = 0.5 * df, b = 0.5, r = t2 / df;
  double x = 1.0 / (1.0 + r), y = r / (1.0 + r);
  double front = exp(a * (-log1p(r)) + b * log(y) - lbeta_half(a));
  if (x < (a + 1.0) / (a + b + 2.0)) return front * betacf(a, b, x) / a;
  return 1.0 - front * betacf(b, a, y) / b;
}

/* chi2.sf(X, 2W) */
static double chi2_sf_even(double X, int W) {
  double x = 0.5 * X;
  if (isnan(x)) return x;
  if (x <= 0.0) return 1.0;
  if (isinf(x)) return 0.0;
  /* sum_{m<W} exp(-x + m log x - lgamma(m+1)), largest term first for range safety */
  double sum = 0.0;
  for (int m = W - 1; m >= 0; --m) sum += exp(-x + m * log(x) - lgamma(m + 1.0));
  return sum > 1.0 ? 1.0 : sum;
}

/* ---- per-position tests ------------------------------------------------ */
/* numpy's add.reduce over a contiguous double array (numpy/core/src/umath/loops_utils.h, DOUBLE_pairwise_sum; the same
 * since numpy 1.9, so also the reference's 1.16.6): fewer than 8 elements in order, up to 128 with eight strided
 * accumulators, longer arrays split in halves (the first a multiple of 8).  np.mean and np.var — and through them
 * scipy's ttest_ind, myDetect.py:335 — sum this way, so the Welch statistic is restated with the same rounding order:
 * on rows with a large common level a sequential sum differs from it by ~1e-13 in t. */
static double np_pairwise_sum(const double* a, int64_t n) {
  if (n < 8) {
    double res = 0.0;
    for (int64_t i = 0; i < n; ++i) res += a[i];
    return res;
  }
  if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int64_t i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  }
  int64_t n2 = n / 2;
  n2 -= n2 % 8;
  return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
}

static int cmp_double(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b;
  return (x > y) - (x < y);
}

typedef struct { double mwu_u, mwu_p, t_t, t_p, ks_d, ks_p; unsigned status; } pos_result;

static void test_position(const double* a, int n0, const double* b, int n1, int tests, double* work, pos_result* r) {
  r->status = 0;
  r->mwu_u = r->mwu_p = r->t_t = r->t_p = r->ks_d = r->ks_p = NAN;
  if (n0 <= 0 || n1 <= 0) { r->status |= ORC_STATUS_EMPTY; return; }
  double* sa = work; double* sb = work + n0;
  memcpy(sa, a, sizeof(double) * n0); memcpy(sb, b, sizeof(double) * n1);
  qsort(sa, n0, sizeof(double), cmp_double);
  qsort(sb, n1, sizeof(double), cmp_double);

  if (tests & 1) {
    /* ks_2samp: cdf = searchsorted(side='right')/n at every pooled value; d = max |cdf1 - cdf2| */
    double d = 0.0;
    int i = 0, j = 0;
    while (i < n0 || j < n1) {
      double v = (j >= n1 || (i < n0 && sa[i] <= sb[j])) ? sa[i] : sb[j];
      while (i < n0 && sa[i] <= v) ++i;
      while (j < n1 && sb[j] <= v) ++j;
      double diff = fabs((double)i / (1.0 * n0) - (double)j / (1.0 * n1));
      if (diff > d) d = diff;
    }
    double en = sqrt((double)((int64_t)n0 * n1) / (double)(n0 + n1));
    double p = kolmogorov_sf((en + 0.12 + 0.11 / en) * d);
    r->ks_d = m_max_float(d); r->ks_p = m_min_float(p);
  }
  if (tests & 2) {
    /* mannwhitneyu(alternative=None): average ranks of x in the pooled sample, tie correction */
    double ranksum = 0.0, tiesum = 0.0;
    int i = 0, j = 0, rank0 = 0;       /* rank0 = number of pooled elements strictly below the current tie group */
    while (i < n0 || j < n1) {
      double v = (j >= n1 || (i < n0 && sa[i] <= sb[j])) ? sa[i] : sb[j];
      int ca = 0, cb = 0;
      while (i < n0 && sa[i] == v) { ++i; ++ca; }
      while (j < n1 && sb[j] == v) { ++j; ++cb; }
      int t = ca + cb;
      ranksum += ca * (rank0 + 0.5 * (t + 1));
      tiesum += (double)t * t * t - t;
      rank0 += t;
    }
    double prod = (double)n0 * n1;
    double u1 = prod + (n0 * (n0 + 1.0)) / 2.0 - ranksum, u2 = prod - u1;
    double size = n0 + n1;
    double T = size < 2 ? 1.0 : 1.0 - tiesum / (size * size * size - size);
    if (T == 0) { r->status |= ORC_STATUS_MWU_ALL_IDENTICAL; }
    else {
      double sd = sqrt(T * n0 * n1 * (n0 + n1 + 1) / 12.0);
      double meanrank = prod / 2.0 + 0.5;
      double bigu = u1 > u2 ? u1 : u2;
      double z = (bigu - meanrank) / sd;
      r->mwu_p = m_min_float(norm_sf(fabs(z)));
      r->mwu_u = m_max_float(u1 < u2 ? u1 : u2);
    }
  }
  if (tests & 4) {
    /* ttest_ind(equal_var=False) */
    /* np.mean / np.var(ddof=1) as scipy's ttest_ind calls them: numpy's pairwise summation (below), the squared
       deviations formed first (work) and summed the same way */
    double m0 = np_pairwise_sum(a, n0) / n0, m1 = np_pairwise_sum(b, n1) / n1;
    for (int i = 0; i < n0; ++i) work[i] = (a[i] - m0) * (a[i] - m0);
    double q0 = np_pairwise_sum(work, n0);
    for (int j = 0; j < n1; ++j) work[j] = (b[j] - m1) * (b[j] - m1);
    double q1 = np_pairwise_sum(work, n1);
    double v1 = q0 / (n0 - 1.0), v2 = q1 / (n1 - 1.0), vn1 = v1 / n0, vn2 = v2 / n1;
    double df = (vn1 + vn2) * (vn1 + vn2) / (vn1 * vn1 / (n0 - 1.0) + vn2 * vn2 / (n1 - 1.0));
    if (isnan(df)) df = 1.0;
    double t = (m0 - m1) / sqrt(vn1 + vn2);
    double p = student_t_two_sided(t, df);
    if (isnan(p)) r->status |= ORC_STATUS_T_NAN;
    r->t_t = m_max_float(t); r->t_p = m_min_float(p);
  }
}

/* get_combin_pvalue over a whole track (myDetect.py:379-414) */
static void combine_track(int64_t npos, const double* ks_d, const double* ks_p, const int32_t* run_id, int nb,
                          double wdif, int method, double* comb_st, double* comb_p) {
  if (nb == 0) { for (int64_t i = 0; i < npos; ++i) { comb_st[i] = ks_d[i]; comb_p[i] = ks_p[i]; } return; }
  int W = 2 * nb + 1;
  double* w = (double*)malloc(sizeof(double) * W);
  w[nb] = 100.0;
  for (int k = 1; k <= nb; ++k) { w[nb - k] = w[nb - k + 1] / wdif; w[nb + k] = w[nb + k - 1] / wdif; }
  double norm = 0; for (int k = 0; k < W; ++k) norm += w[k] * w[k];
  norm = sqrt(norm);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < npos; ++i) {
    double acc = 0.0;
    for (int k = -nb; k <= nb; ++k) {
      int64_t j = i + k;
      double p = (j < 0 || j > npos - 1 || run_id[j] != run_id[i]) ? 1.0 : ks_p[j];
      acc += (method == ORC_METHOD_STOUFFER) ? w[nb + k] * norm_isf(p) : log(p);
    }
    double st, pv;
    if (method == ORC_METHOD_STOUFFER) { st = acc / norm; pv = norm_sf(st); }
    else { st = -2.0 * acc; pv = chi2_sf_even(st, W); }
    comb_p[i] = m_min_float(pv); comb_st[i] = m_max_float(st);
  }
  free(w);
}

/* Batch entry: CSR float32 signals (up-cast exactly), all outputs fp64[npos], status u8[npos].
 * dtype: 0 = float32, 1 = int16 milli-units (value = k/1000.0). */
int nmod_oracle_detect(int64_t npos, int dtype, const void* sig0, const int64_t* off0, const void* sig1,
                       const int64_t* off1, const int32_t* run_id, int nb, double wdif, int method, int tests,
                       double* mwu_u, double* mwu_p, double* t_t, double* t_p, double* ks_d, double* ks_p,
                       double* comb_st, double* comb_p, uint8_t* status, int threads) {
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#endif
  int64_t maxn = 0;
  for (int64_t i = 0; i < npos; ++i) {
    int64_t n = (off0[i + 1] - off0[i]) + (off1[i + 1] - off1[i]);
    if (n > maxn) maxn = n;
  }
  int tests_eff = tests | (method != ORC_METHOD_KS ? 1 : 0);
#pragma omp parallel
  {
    double* buf = (double*)malloc(sizeof(double) * (size_t)(4 * maxn + 8));
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < npos; ++i) {
      int n0 = (int)(off0[i + 1] - off0[i]), n1 = (int)(off1[i + 1] - off1[i]);
      double* a = buf; double* b = buf + n0; double* work = buf + n0 + n1;
      if (dtype == 0) {
        const float* s0 = (const float*)sig0 + off0[i]; const float* s1 = (const float*)sig1 + off1[i];
        for (int k = 0; k < n0; ++k) a[k] = (double)s0[k];
        for (int k = 0; k < n1; ++k) b[k] = (double)s1[k];
      } else {
        const int16_t* s0 = (const int16_t*)sig0 + off0[i]; const int16_t* s1 = (const int16_t*)sig1 + off1[i];
        for (int k = 0; k < n0; ++k) a[k] = (double)s0[k] / 1000.0;
        for (int k = 0; k < n1; ++k) b[k] = (double)s1[k] / 1000.0;
      }
      pos_result r;
      test_position(a, n0, b, n1, tests_eff, work, &r);
      if (mwu_u) mwu_u[i] = r.mwu_u; if (mwu_p) mwu_p[i] = r.mwu_p;
      if (t_t) t_t[i] = r.t_t; if (t_p) t_p[i] = r.t_p;
      if (ks_d) ks_d[i] = r.ks_d; if (ks_p) ks_p[i] = r.ks_p;
      if (status) status[i] = (uint8_t)r.status;
    }
    free(buf);
  }
  if (method != ORC_METHOD_KS && comb_st && comb_p && ks_d && ks_p)
    combine_track(npos, ks_d, ks_p, run_id, nb, wdif, method, comb_st, comb_p);
  return 0;
}

int nmod_oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
