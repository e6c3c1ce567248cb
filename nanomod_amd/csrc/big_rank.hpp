// K1, large-position form: positions whose sorted group does not fit the wave-resident kernels
// (more than 2 048 samples in a group in all-tests mode, or in the SMALLER group in KS-only mode).
// The reference has no size limit (scipy sorts whatever getKStest is given, myDetect.py:327-343); deep
// coverage is rare, so this path is built for correctness and generality, not for the roofline:
// one 256-thread workgroup per position,
//   1. copies both groups into a power-of-two scratch slab in HBM (+inf pads) and takes the fp64 moments,
//   2. sorts each slab with a block-wide bitonic network — in LDS when the group fits 8 192 keys, in the
//      (L2-resident) slab otherwise,
//   3. ranks group 2 into group 1 with plain binary searches and applies the same per-run formulas as
//      rank_all.hpp: rank sum, tie term, the KS candidates (U, j_e) / (L, j_s) in the float form of
//      ks_2samp (all-tests) or as the exact integer numerator (KS-only).
// Scratch comes from a bump allocator over a slab that the host sizes from the classifier's totals.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rank_stats.hpp"

namespace nmod {

constexpr int kBigThreads = 256;          // (1024 threads per block: 1.5x slower, the barriers dominate)
constexpr int kBigLdsBytes = 32768;         // groups whose padded keys fit are sorted in LDS (8 192 fp32 / 4 096 fp64 keys)

struct BigArgs {
  const void* sig0; const void* sig1;
  const int64_t* off0; const int64_t* off1;
  int64_t stride0, stride1;
  const int32_t* pos_list;                  // the big positions (class kBigClass of the binning)
  const int32_t* class_meta;                // [kBigClass] count, [kClassStride + kBigClass] offset into pos_list
  int32_t big_class;
  int32_t all;                              // 1: MWU / Welch / float-form D; 0: KS numerator only
  void* scratch;                            // slab of sum (pow2(n0) + pow2(n1)) keys (fp32, or fp64 for DTYPE 2)
  unsigned long long* cursor;               // bump allocator over the slab
  uint32_t* ks_num; uint64_t* mwu_s; uint64_t* tie; double* moments; double* ks_d_ref;
};

__host__ __device__ inline int64_t big_pow2_ceil(int64_t n) {
  int64_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

// DTYPE 0: fp32 samples, 1: int16 milli-units (fp32 keys), 2: fp64 samples with fp64 keys — the path for float64
// input that is neither float32-exact nor on the 0.001 grid (NMOD_DTYPE_F64): every position then comes here
template <int DTYPE> struct BigKey { typedef float type; };
template <> struct BigKey<2> { typedef double type; };

template <int DTYPE>
__device__ __forceinline__ typename BigKey<DTYPE>::type big_load(const void* sig, int64_t i) {
  if constexpr (DTYPE == 0) return reinterpret_cast<const float*>(sig)[i];
  else if constexpr (DTYPE == 1) return (float)reinterpret_cast<const int16_t*>(sig)[i];
  else return reinterpret_cast<const double*>(sig)[i];
}

__device__ __forceinline__ double big_block_sum(double v, double* red) {
  const int tid = threadIdx.x;
  v = wave_sum_f64(v);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kBigThreads / 64; ++w) t += red[w];
  return t;
}

// bitonic network over P keys (P a power of two) by the whole block; `keys` is LDS or global memory
template <typename K>
__device__ __forceinline__ void big_bitonic(K* keys, int P) {
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < (P >> 1); t += kBigThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // the element of the pair with bit j clear
        const int p = i | j;
        const bool up = (i & k) == 0;
        const K a = keys[i], b = keys[p];
        if ((a > b) == up) { keys[i] = b; keys[p] = a; }
      }
      __syncthreads();
    }
  }
}

template <typename K>
__device__ __forceinline__ int big_lower_bound(const K* s, int n, K x) {   // #{s < x}
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (s[mid] < x) lo = mid + 1; else hi = mid; }
  return lo;
}
template <typename K>
__device__ __forceinline__ int big_upper_bound(const K* s, int n, K x) {   // #{s <= x}
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (s[mid] <= x) lo = mid + 1; else hi = mid; }
  return lo;
}

template <int DTYPE>
__global__ __launch_bounds__(kBigThreads)
void big_rank_kernel(BigArgs a) {
  typedef typename BigKey<DTYPE>::type K;
  constexpr int kBigLdsKeys = kBigLdsBytes / (int)sizeof(K);
  __shared__ K lds_keys[kBigLdsKeys];
  __shared__ double red[kBigThreads / 64];
  __shared__ unsigned long long sh_base, sh_s, sh_t, sh_best, sh_dbest;
  const int tid = threadIdx.x;
  const int64_t count = a.class_meta[a.big_class];
  const int32_t* list = a.pos_list + a.class_meta[kClassStride + a.big_class];
  const K inf = (K)__builtin_inff();

  for (int64_t bi = blockIdx.x; bi < count; bi += gridDim.x) {
    const int64_t pos = list[bi];
    int64_t o0, o1; int n0, n1;
    if (a.stride0 > 0) { o0 = pos * a.stride0; n0 = (int)a.stride0; } else { o0 = a.off0[pos]; n0 = (int)(a.off0[pos + 1] - o0); }
    if (a.stride1 > 0) { o1 = pos * a.stride1; n1 = (int)a.stride1; } else { o1 = a.off1[pos]; n1 = (int)(a.off1[pos + 1] - o1); }
    const int P0 = (int)big_pow2_ceil(n0), P1 = (int)big_pow2_ceil(n1);
    if (tid == 0) {
      sh_base = atomicAdd(a.cursor, (unsigned long long)(P0 + P1));
      sh_s = 0ull; sh_t = 0ull; sh_best = 0ull; sh_dbest = 0ull;
    }
    __syncthreads();
    K* A = reinterpret_cast<K*>(a.scratch) + sh_base;
    K* B = A + P0;

    // ---- moments (two-pass, fp64) and the sorted copies
    for (int g = 0; g < 2; ++g) {
      const void* sig = g ? a.sig1 : a.sig0;
      const int64_t off = g ? o1 : o0;
      const int n = g ? n1 : n0, P = g ? P1 : P0;
      K* dst = g ? B : A;
      if (a.all) {
        double s = 0.0;
        for (int i = tid; i < n; i += kBigThreads) s += (double)big_load<DTYPE>(sig, off + i);
        s = big_block_sum(s, red);
        const double mu = s / (double)n;
        double q = 0.0;
        for (int i = tid; i < n; i += kBigThreads) { const double d = (double)big_load<DTYPE>(sig, off + i) - mu; q += d * d; }
        q = big_block_sum(q, red);
        if (tid == 0) {
          double* mo = a.moments + pos * 4 + 2 * g;
          if constexpr (DTYPE == 1) { mo[0] = s / 1000.0 / (double)n; mo[1] = q * 1e-6; }
          else { mo[0] = mu; mo[1] = q; }
        }
      }
      if (P <= kBigLdsKeys) {
        for (int i = tid; i < P; i += kBigThreads) lds_keys[i] = (i < n) ? big_load<DTYPE>(sig, off + i) : inf;
        __syncthreads();
        big_bitonic(lds_keys, P);
        for (int i = tid; i < P; i += kBigThreads) dst[i] = lds_keys[i];
      } else {
        for (int i = tid; i < P; i += kBigThreads) dst[i] = (i < n) ? big_load<DTYPE>(sig, off + i) : inf;
        __syncthreads();
        big_bitonic(dst, P);
      }
      __syncthreads();
    }

    // ---- rank group 2 into group 1, one run end of group 2 at a time (formulas: rank_all.hpp)
    const int m = n0, q = n1;
    const double dm = (double)m, dq = (double)q;
    unsigned long long s_acc = 0ull, t_acc = 0ull, best = 0ull;
    double dmax = 0.0;
    for (int j = tid; j < q; j += kBigThreads) {
      const K x = B[j];
      if (j + 1 < q && B[j + 1] == x) continue;                     // not the end of its run
      const int js = (j > 0 && B[j - 1] == x) ? big_lower_bound(B, q, x) : j, je = j + 1;   // a run of one: no search
      const int L = big_lower_bound(A, m, x);
      const int U = (L < m && A[L] == x) ? big_upper_bound(A, m, x) : L;
      const unsigned long long ta = (unsigned long long)(U - L), tb = (unsigned long long)(je - js);
      s_acc += tb * (unsigned long long)(2 * m - U - L);
      t_acc += tb * tb * tb - tb + 3ull * ta * tb * (ta + tb);
      {
        const double d_at = (double)U / dm - (double)je / dq;            // ks_2samp's float form: fl(c0/n0) - fl(c1/n1)
        const double d_before = (double)L / dm - (double)js / dq;
        dmax = fmax(dmax, fmax(fabs(d_at), fabs(d_before)));
      }
      if (!a.all) {
        const long long n_at = (long long)U * q - (long long)je * m, n_before = (long long)L * q - (long long)js * m;
        const unsigned long long m_at = (unsigned long long)(n_at < 0 ? -n_at : n_at);
        const unsigned long long m_before = (unsigned long long)(n_before < 0 ? -n_before : n_before);
        best = max(best, max(m_at, m_before));
      }
    }
    if (a.all) {
      for (int i = tid; i < m; i += kBigThreads) {                   // runs of group 1: a^3 - a each
        const K x = A[i];
        if (i + 1 < m && A[i + 1] == x) continue;
        if (i == 0 || A[i - 1] != x) continue;                       // a run of one adds 1^3 - 1 = 0
        const unsigned long long ta = (unsigned long long)(i + 1 - big_lower_bound(A, m, x));
        t_acc += ta * ta * ta - ta;
      }
      atomicAdd(&sh_s, s_acc);
      atomicAdd(&sh_t, t_acc);
    }
    atomicMax(&sh_dbest, (unsigned long long)__double_as_longlong(dmax));   // non-negative doubles order like their bits
    if (!a.all) atomicMax(&sh_best, best);
    __syncthreads();
    if (tid == 0) {
      a.ks_d_ref[pos] = __longlong_as_double((long long)sh_dbest);  // every mode: the float form K2 reports as D
      if (a.all) {
        a.mwu_s[pos] = sh_s; a.tie[pos] = sh_t;
      } else {
        a.ks_num[pos] = (uint32_t)sh_best;                           // <= 65 535^2 < 2^32
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// All-tests mode, positions whose LARGER group is beyond the wave-resident kernels (up to 4 096 samples) and whose
// SMALLER group has at most 1 024 (config 5: ~2 500 v ~60 reads): the block sorts only the smaller group S in LDS, every sample of the larger group Q
// finds its ranks L, U in S by binary search and is counted in LDS histograms — the formulas of rank_hist.hpp — and
// the ties INSIDE Q, the one thing that would need Q sorted, are counted by an open-addressing hash table in LDS:
// an arrival that finds its key present gets the number of earlier arrivals from the slot's counter and adds
// p (p - 1).  One pass over Q, no scratch slab, no copy of Q anywhere.
constexpr int kBigHistMaxS = 1024;              // keys of S in LDS, a power of two (the sort pads to one); two blocks per CU
constexpr int kBigHistMaxQ = 4096;              // samples of Q: half the slots of the hash table
constexpr int kBigHistSlots = 8192;
constexpr unsigned kBigHistEmpty = 0xffffffffu;  // (a NaN pattern no arithmetic produces)
constexpr int kBigHistAux = 24;                  // words of block-wide accumulators behind the tables
constexpr size_t kBigHistLds = ((size_t)kBigHistMaxS + (size_t)(kBigHistMaxS + 1) + (size_t)kBigHistSlots * 2 + kBigHistAux + 3) / 4 * 16;
static_assert(2 * kBigHistLds <= 160 * 1024, "two blocks per CU");

template <int DTYPE>
__global__ __launch_bounds__(kBigThreads)
void big_hist_kernel(BigArgs a) {
  static_assert(DTYPE == 0 || DTYPE == 1, "float32 or int16 samples");
  extern __shared__ __attribute__((aligned(16))) unsigned big_lds[];
  float* keys = reinterpret_cast<float*>(big_lds);                   // sorted S, padded with +inf to a power of two
  unsigned* hist = big_lds + kBigHistMaxS;                             // bin k: (#L == k) << 16 | (#U == k), k = 0 .. m
  unsigned* hkey = hist + (kBigHistMaxS + 1);                          // hash table: key bits / occurrences - 1
  unsigned* hcnt = hkey + kBigHistSlots;
  // block-wide accumulators (in the dynamic allocation: a static __shared__ array would push the block over half a CU)
  unsigned* aux = hcnt + kBigHistSlots + ((kBigHistMaxS + (kBigHistMaxS + 1)) & 1);     // 8-byte aligned
  double* red = reinterpret_cast<double*>(aux);                                          // [4]
  unsigned long long& sh_pp = *reinterpret_cast<unsigned long long*>(aux + 8);
  unsigned long long& sh_ab = *reinterpret_cast<unsigned long long*>(aux + 10);
  unsigned long long& sh_slu = *reinterpret_cast<unsigned long long*>(aux + 12);
  unsigned long long& sh_dmax = *reinterpret_cast<unsigned long long*>(aux + 14);
  unsigned& sh_best = aux[16];
  unsigned* sh_scan = aux + 17;                                                          // [4]
  const int tid = threadIdx.x;
  const int64_t count = a.class_meta[a.big_class];
  const int32_t* list = a.pos_list + a.class_meta[kClassStride + a.big_class];
  const float inf = __builtin_inff();

  for (int64_t bi = blockIdx.x; bi < count; bi += gridDim.x) {
    const int64_t pos = list[bi];
    int64_t o0, o1; int n0, n1;
    if (a.stride0 > 0) { o0 = pos * a.stride0; n0 = (int)a.stride0; } else { o0 = a.off0[pos]; n0 = (int)(a.off0[pos + 1] - o0); }
    if (a.stride1 > 0) { o1 = pos * a.stride1; n1 = (int)a.stride1; } else { o1 = a.off1[pos]; n1 = (int)(a.off1[pos + 1] - o1); }
    const bool swap = n1 < n0;                       // S = the smaller group (ties: group 1)
    const void* sig_s = swap ? a.sig1 : a.sig0; const void* sig_q = swap ? a.sig0 : a.sig1;
    const int64_t off_s = swap ? o1 : o0, off_q = swap ? o0 : o1;
    const int m = swap ? n1 : n0, q = swap ? n0 : n1;
    const int P = (int)big_pow2_ceil(m);
    if (tid == 0) { sh_pp = 0ull; sh_ab = 0ull; sh_slu = 0ull; sh_best = 0u; sh_dmax = 0ull; }

    // ---- both groups into registers with every load in flight at once (a block works on one position at a time: a
    // dependent load per loop trip would leave it waiting on HBM latency), then moments (two-pass, fp64) from there
    constexpr int NS = kBigHistMaxS / kBigThreads, NQ = kBigHistMaxQ / kBigThreads;
    static_assert(NS * kBigThreads == kBigHistMaxS && NQ * kBigThreads == kBigHistMaxQ && (kBigHistMaxS & (kBigHistMaxS - 1)) == 0, "whole registers per thread");
    float xs[NS], xq[NQ];
#pragma unroll
    for (int t = 0; t < NS; ++t) { const int i = tid + t * kBigThreads; xs[t] = (i < m) ? big_load<DTYPE>(sig_s, off_s + i) : 0.0f; }
#pragma unroll
    for (int t = 0; t < NQ; ++t) { const int j = tid + t * kBigThreads; xq[t] = (j < q) ? big_load<DTYPE>(sig_q, off_q + j) : 0.0f; }
    for (int g = 0; g < 2; ++g) {
      const int n = g ? q : m;
      double s = 0.0;
      if (g == 0) {
#pragma unroll
        for (int t = 0; t < NS; ++t) if (tid + t * kBigThreads < m) s += (double)xs[t];
      } else {
#pragma unroll
        for (int t = 0; t < NQ; ++t) if (tid + t * kBigThreads < q) s += (double)xq[t];
      }
      s = big_block_sum(s, red);
      const double mu = s / (double)n;
      double qq = 0.0;
      if (g == 0) {
#pragma unroll
        for (int t = 0; t < NS; ++t) if (tid + t * kBigThreads < m) { const double d = (double)xs[t] - mu; qq += d * d; }
      } else {
#pragma unroll
        for (int t = 0; t < NQ; ++t) if (tid + t * kBigThreads < q) { const double d = (double)xq[t] - mu; qq += d * d; }
      }
      qq = big_block_sum(qq, red);
      if (tid == 0) {
        double* mo = a.moments + pos * 4 + 2 * ((g == 1) != swap ? 1 : 0);     // S is group 1 unless swapped
        if constexpr (DTYPE == 1) { mo[0] = s / 1000.0 / (double)n; mo[1] = qq * 1e-6; }
        else { mo[0] = mu; mo[1] = qq; }
      }
    }

    // ---- S sorted in LDS; histograms and hash table cleared
#pragma unroll
    for (int t = 0; t < NS; ++t) { const int i = tid + t * kBigThreads; if (i < P) keys[i] = (i < m) ? xs[t] + 0.0f : inf; }
    for (int i = tid; i <= m; i += kBigThreads) hist[i] = 0u;
    for (int i = tid; i < kBigHistSlots; i += kBigThreads) { hkey[i] = kBigHistEmpty; hcnt[i] = 0u; }
    __syncthreads();
    big_bitonic(keys, P);

    // ---- every sample of Q: ranks in S -> histograms; its own ties -> hash table.  Four samples per trip with
    // branchless fixed-depth searches over the padded keys (P = 2^k, pads +inf): four independent chains of LDS reads
    // in flight instead of one (two waves per SIMD hide little of a dependent chain)
    unsigned long long pp = 0ull;
    const int levels = 31 - __builtin_clz((unsigned)P);                     // log2(P), uniform over the block
#pragma unroll 1
    for (int t0 = 0; t0 < NQ; t0 += 4) {
      if (t0 * kBigThreads >= q) break;                                     // (uniform: no thread has a sample left)
      float x[4]; bool have[4]; int L[4], U[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        have[e] = tid + (t0 + e) * kBigThreads < q;
        x[e] = have[e] ? xq[t0 + e] + 0.0f : inf;                          // (-0.0 -> +0.0: one key per value)
        L[e] = 0; U[e] = 0;
      }
      for (int lv = 0, h = P >> 1; lv < levels; ++lv, h >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float kl = keys[L[e] + h - 1], ku = keys[U[e] + h - 1];
          L[e] += (kl < x[e]) ? h : 0;                                      // #{s < x}
          U[e] += (ku <= x[e]) ? h : 0;                                     // #{s <= x}
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (!have[e]) continue;
        // (one more comparison: the loop ranks among P - 1 keys)
        const int Lf = L[e] + ((keys[L[e]] < x[e] && L[e] == P - 1) ? 1 : 0);
        const int Uf = U[e] + ((keys[U[e]] <= x[e] && U[e] == P - 1) ? 1 : 0);
        atomicAdd(&hist[min(Lf, m)], 0x10000u);
        atomicAdd(&hist[min(Uf, m)], 1u);
        const unsigned bits = __float_as_uint(x[e]);
        unsigned h = (bits * 2654435761u) >> 19;                            // 13 bits: kBigHistSlots = 8192
        for (;;) {
          const unsigned old = atomicCAS(&hkey[h], kBigHistEmpty, bits);
          if (old == kBigHistEmpty) break;                                  // first of its value
          if (old == bits) {                                                // the p-th, p = c + 2: adds p (p - 1)
            const unsigned long long c = atomicAdd(&hcnt[h], 1u);
            pp += (c + 2ull) * (c + 1ull);
            break;
          }
          h = (h + 1u) & (kBigHistSlots - 1u);
        }
      }
    }
    // ties inside S: every run of a equal keys adds (a^3 - a) / 3 = sum p (p - 1)
    for (int i = tid; i < m; i += kBigThreads) {
      const float x = keys[i];
      if (i + 1 < m && keys[i + 1] == x) continue;
      if (i == 0 || keys[i - 1] != x) continue;
      const unsigned long long ta = (unsigned long long)(i + 1 - big_lower_bound(keys, m, x));
      pp += (ta * ta * ta - ta) / 3ull;
    }
    __syncthreads();

    // ---- prefix sums over the bins 0 .. m (each thread a contiguous chunk), candidates at the run ends of S
    const int per = (m + 1 + kBigThreads - 1) / kBigThreads;
    const int k0 = min(tid * per, m + 1), k1 = min(k0 + per, m + 1);
    unsigned loc = 0;
    for (int k = k0; k < k1; ++k) loc += hist[k];
    // block exclusive scan of `loc` (packed 16 | 16: both halves stay <= q <= 4096)
    unsigned inc = loc;
    for (int d = 1; d < 64; d <<= 1) { const unsigned t = __shfl_up(inc, d); if ((tid & 63) >= d) inc += t; }
    if ((tid & 63) == 63) sh_scan[tid >> 6] = inc;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += sh_scan[w];
    unsigned c2 = base + inc - loc;                                         // cumL(k0 - 1) << 16 | cumU(k0 - 1)
    unsigned best = 0; unsigned long long slu = 0ull, ab = 0ull;
    for (int pass = 0; pass < 2; ++pass) {
      // pass 0: the integer maximum, the Mann-Whitney sum and the ties with S; pass 1: the float form at the maximum
      unsigned c = c2;
      double dmax = 0.0;
      const unsigned target = sh_best;
      for (int k = k0; k < k1; ++k) {
        const int cl_prev = (int)(c >> 16), cu_prev = (int)(c & 0xffffu);  // cumL(k-1), cumU(k-1)
        c += hist[k];
        const int cu = (int)(c & 0xffffu);                                  // cumU(k)
        const bool run_end = (k == 0) || (k == m) || (keys[k - 1] != keys[k]);   // k = 0: the candidate (cumU(0), 0)
        const long long cand_a = (long long)cu * m - (long long)k * q;
        const long long cand_b = (k > 0) ? (long long)cl_prev * m - (long long)k * q : 0ll;
        const unsigned ma = run_end ? (unsigned)(cand_a < 0 ? -cand_a : cand_a) : 0u;
        const unsigned mb = (run_end && k > 0) ? (unsigned)(cand_b < 0 ? -cand_b : cand_b) : 0u;
        if (pass == 0) {
          best = max(best, max(ma, mb));
          if (k > 0) {
            slu += (unsigned long long)(2 * q - cl_prev - cu_prev);
            if (run_end && cl_prev != cu_prev) {                            // Q ties with the run of S that ends at k
              const unsigned long long b = (unsigned long long)(cl_prev - cu_prev);
              const unsigned long long ta = (unsigned long long)(k - big_lower_bound(keys, m, keys[k - 1]));
              ab += ta * b * (ta + b);
            }
          }
        } else {
          const double dm = (double)m, dq = (double)q;                      // ks_2samp's float form: |fl(c0/n0) - fl(c1/n1)|
          if (ma == target && run_end) dmax = fmax(dmax, fabs((double)k / dm - (double)cu / dq));
          if (mb == target && run_end && k > 0) dmax = fmax(dmax, fabs((double)k / dm - (double)cl_prev / dq));
        }
      }
      if (pass == 0) {                                                      // one atomic per wave, not per thread
        const unsigned wb = wave_max_u32(best);
        const unsigned long long ws = wave_sum_u64(slu), wa = wave_sum_u64(ab), wp = wave_sum_u64(pp);
        if ((tid & 63) == 0) { atomicMax(&sh_best, wb); atomicAdd(&sh_slu, ws); atomicAdd(&sh_ab, wa); atomicAdd(&sh_pp, wp); }
      } else {
        const double wd = wave_max_f64(dmax);
        if ((tid & 63) == 0) atomicMax(&sh_dmax, (unsigned long long)__double_as_longlong(wd));   // non-negative doubles order like their bits
      }
      __syncthreads();
    }
    if (tid == 0) {
      const unsigned long long s_lu = sh_slu;                                // sum over Q of (L + U)
      a.mwu_s[pos] = swap ? s_lu : 2ull * (unsigned long long)m * (unsigned long long)q - s_lu;
      a.tie[pos] = 3ull * sh_pp + 3ull * sh_ab;
      a.ks_d_ref[pos] = (sh_best != 0u) ? __longlong_as_double((long long)sh_dmax) : 0.0;
      a.ks_num[pos] = sh_best;
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wide_redo_kernel — the ties INSIDE the streamed group of the positions that rank_hist_kernel's WIDE float32 form put on
// its redo list (more samples on shared bitmap bits than its exact table takes: hundreds to thousands of equal samples off
// the milli-unit grid).  One workgroup per listed position: the larger group (<= 4 096 samples) is sorted in LDS and
// 3 x sum over its elements of p (p - 1), p = place of the element in its run of equal keys, is ADDED to tie[pos] — the
// WIDE kernel has stored everything else of the position (the ties of the sorted group and between the groups included).
// The list length is read on the device: the launch needs no host round trip and returns at once when the list is empty.
struct WideRedoArgs {
  const void* sig0; const void* sig1; const int64_t* off0; const int64_t* off1; int64_t stride0, stride1;
  const int32_t* list; const int32_t* count; uint64_t* tie;
};
constexpr int kWideRedoMaxQ = 4096;

__global__ __launch_bounds__(kBigThreads)
void wide_redo_kernel(WideRedoArgs a) {
  __shared__ float keys[kWideRedoMaxQ];
  __shared__ int carry[kBigThreads];
  __shared__ double red[kBigThreads / 64];
  const int n_items = *a.count;
  const int tid = threadIdx.x;
  for (int it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int64_t pos = a.list[it];
    int64_t o0, o1; int n0, n1;
    if (a.stride0 > 0) { o0 = pos * a.stride0; n0 = (int)a.stride0; } else { o0 = a.off0[pos]; n0 = (int)(a.off0[pos + 1] - o0); }
    if (a.stride1 > 0) { o1 = pos * a.stride1; n1 = (int)a.stride1; } else { o1 = a.off1[pos]; n1 = (int)(a.off1[pos + 1] - o1); }
    const bool swap = n1 < n0;                       // rank_hist_kernel: S = the smaller group (ties: group 1), Q = the other
    const float* src = reinterpret_cast<const float*>(swap ? a.sig0 : a.sig1) + (swap ? o0 : o1);
    const int q = swap ? n0 : n1;
    int P = 64;
    while (P < q) P <<= 1;
    if (q > kWideRedoMaxQ) continue;                 // (not reached: the WIDE classes end at 4 096 samples)
    __syncthreads();
    for (int i = tid; i < P; i += kBigThreads) keys[i] = (i < q) ? src[i] + 0.0f : __builtin_inff();   // (-0.0 -> +0.0)
    __syncthreads();
    big_bitonic(keys, P);
    // place of every element in its run: chunks of P / 256 consecutive keys per thread, the start of the run open at a chunk's
    // first key from a max-scan of the chunks' last run starts
    const int per = P / kBigThreads > 0 ? P / kBigThreads : 1;
    const int lo = tid * per, hi = min(lo + per, q);
    int last = -1;                                   // start of the last run that begins inside this chunk
    for (int i = lo; i < hi; ++i) if (i == 0 || keys[i] != keys[i - 1]) last = i;
    carry[tid] = last;
    __syncthreads();
    int start = -1;
    for (int t = tid - 1; t >= 0 && start < 0; --t) start = carry[t];     // (short: most chunks begin a run)
    double sum = 0.0;
    for (int i = lo; i < hi; ++i) {
      if (i == 0 || keys[i] != keys[i - 1]) start = i;
      const double d = (double)(i - start);          // p - 1
      sum += d * (d + 1.0);                          // p (p - 1): exact in fp64 (< 2^53 for 4 096 samples)
    }
    const double tot = big_block_sum(sum, red);
    if (tid == 0) a.tie[pos] += 3ull * (unsigned long long)tot;
  }
}

}  // namespace nmod
