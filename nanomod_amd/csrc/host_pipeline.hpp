// The host-resident entry (NMOD_MEM_HOST) of nmod_detect_batch: what the reference-shaped drop-in takes — everything
// mtest2 touches is host memory (myDetect.py:416-445; the lists are built at :124).
//
// The batch is cut into chunks of positions.  Per chunk: the rows of both groups (+ their offsets, rebased) go through a
// pinned bounce slot (filled by a small team of host threads; skipped when the caller's arrays are already pinned) and
// one hipMemcpyAsync on a copy stream into a device slot; K1 + K2 run on a compute stream; the chunk's result tracks come
// back through a pinned slab on a third stream and are scattered into the caller's arrays.  The three streams overlap:
// H2D of chunk k+1, kernels of chunk k, D2H of chunk k-1.  The KS track stays on the device for the whole batch and ONE
// K3 runs over it at the end, so windows that straddle a chunk cut and run edges are exactly those of the unchunked
// path.  Device footprint: `slots` x (chunk rows + workspace + result slab) + 36 B per position (KS track, combined
// track, run ids) — bounded by the chunk size, not by the batch.  The path is PCIe-bound (1.6 KB per position in,
// 0.1 KB out): its roofline is the pinned hipMemcpy rate, which bench.py measures in the same run (`host_path`).
//
// (included by nanomod_hip.hip inside namespace nmod, after detect_device / detect_f64 / launch_combine)
#pragma once


// ---------------------------------------------------------------- tunables (nmod_host_pipeline_config; 0 = default)
static std::atomic<int64_t> g_hp_chunk_bytes{0};
static std::atomic<int> g_hp_slots{0}, g_hp_threads{0}, g_hp_mode{0};
constexpr int64_t kHpDefaultChunk = 64ll << 20;      // 64 MiB of samples per chunk (measured at 7.4 GB, pageable input: 0.947 of the pinned H2D rate; 32 MiB: 0.933, 16 MiB: 0.898)
constexpr int kHpMaxSlots = 8;

static int64_t env_i64(const char* name, int64_t dflt) {
  const char* s = getenv(name);
  if (!s || !*s) return dflt;
  char* e = nullptr;
  const long long v = strtoll(s, &e, 10);
  return (e && *e == '\0' && v > 0) ? (int64_t)v : dflt;
}

// CPUs this process may use: the affinity mask capped by the cgroup-v2 quota (a GPU box shows 256 CPUs and allows 16)
static int usable_cpus() {
  int n = (int)std::thread::hardware_concurrency();
  if (n <= 0) n = 1;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64]; long long period = 0;
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
      const long long quota = atoll(q);
      if (quota > 0) n = (int)std::max<long long>(1, std::min<long long>(n, quota / period));
    }
    fclose(f);
  }
  return n;
}

// ---------------------------------------------------------------- float64 on the milli-unit grid -> int16, on the way into the bounce slot
// The reference holds its samples as lists of numpy.float64 (myDetect.py:124) and every stored event value is a 3-decimal number
// (myRefBaseSignalAnnotation.py:1108): x = k / 1000.0 with a small integer k.  The threads that fill a bounce slot touch every
// sample anyway; where ALL samples of a chunk are such values with |k| <= 32 767 they write k as int16 instead of copying the
// double — 2 bytes per sample over PCIe instead of 8, and the chunk runs as NMOD_DTYPE_I16_MILLI (the keys the float64 front end
// would have picked on the device: the same rank statistics bit for bit; the Welch moments as exact integer sums).  A chunk with
// any other sample (off the grid, |k| > 32 767, NaN, infinite) is sent as float64 as before.
// x == k / 1000.0 is tested without a division: q = fl(k / 1000) by one Newton step on k * fl(1 / 1000) — correctly rounded for
// every |k| <= 32 767 (checked exhaustively on the host: tests/test_abi_and_host.py through nmod_narrow_probe).
static inline bool narrow_one(double x, int16_t& out) {
  const double k = nearbyint(x * 1000.0);
  const double q0 = k * 0.001;
  const double q = fma(fma(-q0, 1000.0, k), 0.001, q0);
  out = (int16_t)(int)k;
  return q == x && fabs(k) <= 32767.0;                   // (NaN: false)
}
static bool narrow_scalar(const double* src, int16_t* dst, size_t n) {
  bool ok = true;
  for (size_t i = 0; i < n && ok; ++i) {
    int16_t v = 0;
    if (fabs(src[i]) <= 33.0) ok = narrow_one(src[i], v); else ok = false;     // (the cast of an out-of-range k is undefined: refuse first)
    dst[i] = v;
  }
  return ok;
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
__attribute__((target("avx2,fma"))) static bool narrow_avx2(const double* src, int16_t* dst, size_t n) {
  const __m256d k1000 = _mm256_set1_pd(1000.0), r = _mm256_set1_pd(0.001), lim = _mm256_set1_pd(32767.0);
  const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffll));
  size_t i = 0;
  while (i + 8 <= n) {
    const size_t stop = std::min(n, i + 8192) & ~(size_t)7;              // a verdict per 8 192 samples: an off-grid batch gives up early
    __m256d good = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
    for (; i + 8 <= stop; i += 8) {
      __m128i kk[2];
      for (int h = 0; h < 2; ++h) {
        const __m256d x = _mm256_loadu_pd(src + i + 4 * h);
        const __m256d k = _mm256_round_pd(_mm256_mul_pd(x, k1000), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
        const __m256d q0 = _mm256_mul_pd(k, r);
        const __m256d q = _mm256_fmadd_pd(_mm256_fnmadd_pd(q0, k1000, k), r, q0);
        const __m256d ok = _mm256_and_pd(_mm256_cmp_pd(q, x, _CMP_EQ_OQ), _mm256_cmp_pd(_mm256_and_pd(k, absmask), lim, _CMP_LE_OQ));
        good = _mm256_and_pd(good, ok);
        kk[h] = _mm256_cvtpd_epi32(k);                                    // (garbage for a refused sample: the chunk is then not used)
      }
      _mm_storeu_si128((__m128i*)(dst + i), _mm_packs_epi32(kk[0], kk[1]));
    }
    if (_mm256_movemask_pd(good) != 0xf) return false;
  }
  return narrow_scalar(src + i, dst + i, n - i);
}
#endif
static bool narrow_f64_to_i16(const double* src, int16_t* dst, size_t n) {
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
  static const bool have_avx2 = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
  if (have_avx2) return narrow_avx2(src, dst, n);
#endif
  return narrow_scalar(src, dst, n);
}

// ---------------------------------------------------------------- copy team
// T - 1 helper threads + the caller copy one range each; created per call (a call that needs it moves >= tens of MB).
class CopyTeam {
 public:
  explicit CopyTeam(int threads) : n_(std::max(1, threads)) {
    for (int t = 1; t < n_; ++t) th_.emplace_back([this, t] { worker(t); });
  }
  ~CopyTeam() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; ++gen_; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  void copy(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return;
    if (n_ == 1 || bytes < (size_t)(256 << 10)) { memcpy(dst, src, bytes); return; }
    run(dst, src, bytes, false);
  }
  // `n` doubles -> int16 milli-units (narrow_f64_to_i16), a range per thread; false when a sample refuses (dst is then unspecified)
  bool narrow(int16_t* dst, const double* src, size_t n) {
    if (n == 0) return true;
    if (n_ == 1 || n < (size_t)(32 << 10)) return narrow_f64_to_i16(src, dst, n);
    ok_.store(true);
    run(dst, src, n * 8, true);
    return ok_.load();
  }
 private:
  void run(void* dst, const void* src, size_t bytes, bool narrowing) {
    {
      std::lock_guard<std::mutex> l(m_);
      dst_ = (char*)dst; src_ = (const char*)src; bytes_ = bytes; narrow_ = narrowing; pending_ = n_ - 1; ++gen_;
    }
    cv_.notify_all();
    slice(0);
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this] { return pending_ == 0; });
  }
  void slice(int t) {
    const size_t per = ((bytes_ + n_ - 1) / n_ + 4095) & ~(size_t)4095;       // (bytes of the SOURCE: a multiple of 8 for doubles)
    const size_t lo = (size_t)t * per, hi = std::min(bytes_, lo + per);
    if (lo >= hi) return;
    if (!narrow_) memcpy(dst_ + lo, src_ + lo, hi - lo);
    else if (ok_.load(std::memory_order_relaxed) &&
             !narrow_f64_to_i16((const double*)(src_ + lo), (int16_t*)dst_ + lo / 8, (hi - lo) / 8)) ok_.store(false);
  }
  void worker(int t) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      slice(t);
      { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_one(); }
    }
  }
  int n_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  char* dst_ = nullptr; const char* src_ = nullptr; size_t bytes_ = 0; bool narrow_ = false;
  std::atomic<bool> ok_{true};
  int pending_ = 0; uint64_t gen_ = 0; bool stop_ = false;
};

// ---------------------------------------------------------------- cached per-device resources
// The pinned ring and the three streams survive the call (hipHostMalloc costs ~0.2 ms per MiB); one call at a time
// uses the cached set, a concurrent one makes its own and drops it.  nmod_trim_scratch() frees the cached ring.
struct HostPipeRes {
  char* pinned = nullptr; size_t pinned_bytes = 0;
  hipStream_t s_in = nullptr, s_k = nullptr, s_out = nullptr;
  bool cached = false;
};
static std::mutex g_hp_mutex;
static HostPipeRes g_hp_cache[kMaxDevices];      // guarded by g_hp_mutex
static bool g_hp_busy[kMaxDevices] = {false};

static void hp_destroy(HostPipeRes& r) {
  if (r.pinned) hipHostFree(r.pinned);
  if (r.s_in) hipStreamDestroy(r.s_in);
  if (r.s_k) hipStreamDestroy(r.s_k);
  if (r.s_out) hipStreamDestroy(r.s_out);
  r = HostPipeRes();
}

static int hp_acquire(int dev, size_t pinned_bytes, HostPipeRes& r) {
  r = HostPipeRes();
  if (dev >= 0 && dev < kMaxDevices) {
    std::lock_guard<std::mutex> l(g_hp_mutex);
    if (!g_hp_busy[dev]) { g_hp_busy[dev] = true; r = g_hp_cache[dev]; g_hp_cache[dev] = HostPipeRes(); r.cached = true; }
  }
  if (!r.s_in) NMOD_HIP(hipStreamCreateWithFlags(&r.s_in, hipStreamNonBlocking));
  if (!r.s_k) NMOD_HIP(hipStreamCreateWithFlags(&r.s_k, hipStreamNonBlocking));
  if (!r.s_out) NMOD_HIP(hipStreamCreateWithFlags(&r.s_out, hipStreamNonBlocking));
  if (r.pinned_bytes < pinned_bytes) {
    if (r.pinned) { hipHostFree(r.pinned); r.pinned = nullptr; r.pinned_bytes = 0; }
    NMOD_HIP(hipHostMalloc((void**)&r.pinned, pinned_bytes, hipHostMallocDefault));
    r.pinned_bytes = pinned_bytes;
  }
  return NMOD_OK;
}

static void hp_release(int dev, HostPipeRes& r) {
  if (r.cached && dev >= 0 && dev < kMaxDevices) {
    std::lock_guard<std::mutex> l(g_hp_mutex);
    g_hp_cache[dev] = r; g_hp_cache[dev].cached = false; g_hp_busy[dev] = false;
    r = HostPipeRes();
    return;
  }
  hp_destroy(r);
}

static void hp_trim(int dev) {
  std::lock_guard<std::mutex> l(g_hp_mutex);
  if (!g_hp_busy[dev]) hp_destroy(g_hp_cache[dev]);
}

// is [p, p + bytes) page-locked host memory the runtime knows (hipHostMalloc / hipHostRegister / torch pin_memory)?
static bool hp_is_pinned(const void* p, size_t bytes) {
  if (!p || bytes == 0) return false;
  hipPointerAttribute_t at;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  if (at.type != hipMemoryTypeHost) return false;
  memset(&at, 0, sizeof(at));
  if (hipPointerGetAttributes(&at, (const char*)p + bytes - 1) != hipSuccess) { (void)hipGetLastError(); return false; }
  return at.type == hipMemoryTypeHost;
}

thread_local nmod_host_stats g_host_stats;       // of this thread's last NMOD_MEM_HOST call

// everything a call holds, released in the one order that is safe on every path: drain the three streams, give the device
// slab back to the pool (stream-ordered on the compute stream, idle by then), drop the events, return ring + streams
struct HpCall {
  int dev; HostPipeRes res; DevScratch dmem;
  hipEvent_t ev_in[kHpMaxSlots], ev_k[kHpMaxSlots], ev_out[kHpMaxSlots];
  explicit HpCall(int d) : dev(d) { for (int s = 0; s < kHpMaxSlots; ++s) ev_in[s] = ev_k[s] = ev_out[s] = nullptr; }
  ~HpCall() {
    if (res.s_in) hipStreamSynchronize(res.s_in);
    if (res.s_k) hipStreamSynchronize(res.s_k);
    if (res.s_out) hipStreamSynchronize(res.s_out);
    if (dmem.p) { dmem.release(res.s_k); if (res.s_k) hipStreamSynchronize(res.s_k); }
    for (int s = 0; s < kHpMaxSlots; ++s) {
      if (ev_in[s]) hipEventDestroy(ev_in[s]);
      if (ev_k[s]) hipEventDestroy(ev_k[s]);
      if (ev_out[s]) hipEventDestroy(ev_out[s]);
    }
    hp_release(dev, res);
  }
};

// ---------------------------------------------------------------- the pipeline
static int detect_host_pipelined(const nmod_params* prm, int64_t npos, const void* sig0, const int64_t* off0,
                                 const void* sig1, const int64_t* off1, const int32_t* run_id, nmod_out* out) {
  memset(&g_host_stats, 0, sizeof(g_host_stats));
  if (npos == 0) return NMOD_OK;
  if (!sig0 || !sig1 || !out) return NMOD_ERR_INVALID_ARG;
  if ((prm->stride0 <= 0 && !off0) || (prm->stride1 <= 0 && !off1)) return NMOD_ERR_INVALID_ARG;
  if (npos > INT32_MAX) return NMOD_ERR_INVALID_ARG;
  const bool want_comb = prm->method != NMOD_METHOD_KS && (out->comb_st || out->comb_p);
  if (want_comb && (!out->comb_st || !out->comb_p)) return NMOD_ERR_INVALID_ARG;
  if (want_comb && prm->nb > 0 && !run_id) return NMOD_ERR_INVALID_ARG;
  if (want_comb && prm->method == NMOD_METHOD_STOUFFER && !(prm->weights_dif > 0.0)) return NMOD_ERR_INVALID_ARG;
  const int dev = prm->device;
  const size_t esz = prm->dtype == NMOD_DTYPE_F32 ? 4 : (prm->dtype == NMOD_DTYPE_F64 ? 8 : 2);
  const bool csr0 = prm->stride0 <= 0, csr1 = prm->stride1 <= 0;

  // ---- sizes: the offsets are host memory, so the maxima and the chunk plan cost one pass
  int64_t m0 = csr0 ? 0 : prm->stride0, m1 = csr1 ? 0 : prm->stride1;
  if (csr0) for (int64_t i = 0; i < npos; ++i) { const int64_t n = off0[i + 1] - off0[i]; if (n < 0) return NMOD_ERR_INVALID_ARG; m0 = std::max(m0, n); }
  if (csr1) for (int64_t i = 0; i < npos; ++i) { const int64_t n = off1[i + 1] - off1[i]; if (n < 0) return NMOD_ERR_INVALID_ARG; m1 = std::max(m1, n); }
  // (a group beyond NMOD_MAX_RANKED: the device skips that position and flags it NMOD_STATUS_TOO_LARGE; its rows still cross the bus)
  const int64_t lim0 = std::min<int64_t>(m0, NMOD_MAX_RANKED), lim1 = std::min<int64_t>(m1, NMOD_MAX_RANKED);
  auto row0 = [&](int64_t i) { return csr0 ? off0[i] : i * prm->stride0; };     // first sample of position i (element index)
  auto row1 = [&](int64_t i) { return csr1 ? off1[i] : i * prm->stride1; };
  const int64_t total_bytes = (row0(npos) - row0(0) + row1(npos) - row1(0)) * (int64_t)esz;

  int64_t chunk_bytes = g_hp_chunk_bytes.load();
  if (chunk_bytes <= 0) chunk_bytes = env_i64("NMOD_HOST_CHUNK_BYTES", 0);
  if (chunk_bytes <= 0) chunk_bytes = std::min<int64_t>(kHpDefaultChunk, std::max<int64_t>(1 << 20, total_bytes / 32));   // smaller batches: ~32 chunks, so that fill, copy and kernels still overlap
  int slots = g_hp_slots.load();
  if (slots <= 0) slots = (int)env_i64("NMOD_HOST_SLOTS", 3);
  slots = std::max(2, std::min(slots, kHpMaxSlots));
  constexpr int64_t kMaxChunkPos = 1 << 22;

  std::vector<int64_t> cut;                       // chunk c = positions [cut[c], cut[c + 1])
  cut.push_back(0);
  if (!csr0 && !csr1) {
    const int64_t per = std::max<int64_t>(1, std::min<int64_t>(kMaxChunkPos, chunk_bytes / std::max<int64_t>(1, (m0 + m1) * (int64_t)esz)));
    for (int64_t p = per; p < npos; p += per) cut.push_back(p);
  } else {
    int64_t lo = 0;
    while (lo < npos) {
      // the largest hi with bytes(lo, hi) <= chunk_bytes (at least one position): the byte count is monotone in hi
      int64_t a = lo + 1, b = std::min(npos, lo + kMaxChunkPos);
      const int64_t base = (row0(lo) + row1(lo)) * (int64_t)esz;
      while (a < b) {
        const int64_t mid = a + (b - a + 1) / 2;
        if ((row0(mid) + row1(mid)) * (int64_t)esz - base <= chunk_bytes) a = mid; else b = mid - 1;
      }
      lo = a;
      if (lo < npos) cut.push_back(lo);
    }
  }
  cut.push_back(npos);
  const int nchunks = (int)cut.size() - 1;
  slots = std::min(slots, std::max(2, nchunks));
  int64_t cap_pos = 0, cap_b0 = 0, cap_b1 = 0;
  for (int c = 0; c < nchunks; ++c) {
    cap_pos = std::max(cap_pos, cut[c + 1] - cut[c]);
    cap_b0 = std::max(cap_b0, (row0(cut[c + 1]) - row0(cut[c])) * (int64_t)esz);
    cap_b1 = std::max(cap_b1, (row1(cut[c + 1]) - row1(cut[c])) * (int64_t)esz);
  }

  // ---- which tracks come back per chunk (K1 + K2), in nmod_out member order; the combined pair follows K3
  int tests = prm->tests | (want_comb ? NMOD_TEST_KS : 0);
  double** hp = (double**)out;                    // the 12 leading members are double*
  bool per_chunk[12];
  for (int k = 0; k < 12; ++k) per_chunk[k] = false;
  if (prm->tests & NMOD_TEST_MWU) { per_chunk[0] = hp[0] != nullptr; per_chunk[1] = hp[1] != nullptr; }
  if (prm->tests & NMOD_TEST_WELCH) { per_chunk[2] = hp[2] != nullptr; per_chunk[3] = hp[3] != nullptr; }
  if (tests & NMOD_TEST_KS) { per_chunk[4] = hp[4] != nullptr || want_comb; per_chunk[5] = hp[5] != nullptr || want_comb; }
  if (prm->want_mstd) for (int k = 8; k < 12; ++k) per_chunk[k] = hp[k] != nullptr;
  int ntr = 0, slab_idx[12];
  for (int k = 0; k < 12; ++k) slab_idx[k] = per_chunk[k] ? ntr++ : -1;
  const bool want_status = out->status != nullptr;

  // ---- slot layouts.  input slot (device and, for pageable callers, pinned): [off0 | off1 | sig0 | sig1]
  const int64_t o_off0 = 0;
  const int64_t o_off1 = o_off0 + (csr0 ? align256((cap_pos + 1) * 8) : 0);
  const int64_t o_sig0 = o_off1 + (csr1 ? align256((cap_pos + 1) * 8) : 0);
  const int64_t o_sig1 = o_sig0 + align256(cap_b0 + 256);          // (+256: the kernels' 16-byte row loads may run past a row's end)
  const int64_t in_slot = o_sig1 + align256(cap_b1 + 256);
  // float64 input: the bounce fill narrows a chunk to int16 milli-units where every sample allows it (above); such a batch always
  // goes through the bounce slots
  const bool try_narrow = prm->dtype == NMOD_DTYPE_F64 && (prm->flags & NMOD_FLAG_NO_HOST_NARROW) == 0;
  std::vector<char> narrowed(nchunks, 0);
  const bool pinned_in = !try_narrow && g_hp_mode.load() != 2 && hp_is_pinned((const char*)sig0 + row0(0) * esz, (size_t)((row0(npos) - row0(0)) * (int64_t)esz)) &&
                         hp_is_pinned((const char*)sig1 + row1(0) * esz, (size_t)((row1(npos) - row1(0)) * (int64_t)esz));
  const int64_t in_bounce = pinned_in ? o_sig0 : in_slot;          // a pinned caller's rows are copied from where they are
  const int64_t out_slab = align256(cap_pos * (8 * (int64_t)ntr + 1));
  nmod_params dp = *prm;                         // the chunk calls: device memory, K1 + K2 only
  dp.memspace = NMOD_MEM_DEVICE; dp.method = NMOD_METHOD_KS; dp.tests = tests;
  dp.max_n0 = (int32_t)std::max<int64_t>(lim0, 1); dp.max_n1 = (int32_t)std::max<int64_t>(lim1, 1);
  if (csr0) dp.stride0 = 0;
  if (csr1) dp.stride1 = 0;
  const int64_t wsb = align256(nmod_workspace_bytes(&dp, cap_pos));
  const int64_t dev_slot = in_slot + wsb + out_slab;
  const int64_t stats_bytes = align256(kStatsWords * 8);                                     // dispatch_stats_kernel's accumulator of the call
  const int64_t full_tracks = stats_bytes + (want_comb ? align256(npos * 8) * 4 + align256(npos * 4) : 0);   // + ks_d, ks_p, comb_st, comb_p, run ids
  const size_t pinned_need = (size_t)slots * (size_t)(in_bounce + out_slab);

  HpCall call(dev);
  HostPipeRes& res = call.res;
  DevScratch& dmem = call.dmem;                   // one stream-ordered slab from the library's pool; goes back to it at the end
  int rc = hp_acquire(dev, pinned_need, res);
  if (rc != NMOD_OK) return rc;
  dp.stream = res.s_k;
  NMOD_HIP(dmem.alloc((size_t)(slots * dev_slot + full_tracks), res.s_k, dev));
  NMOD_HIP(hipStreamSynchronize(res.s_k));        // (the copy streams use it too: it must exist before they do)
  char* dbase = (char*)dmem.p;
  unsigned long long* d_stats = (unsigned long long*)(dbase + (int64_t)slots * dev_slot);
  NMOD_HIP(hipMemsetAsync(d_stats, 0, kStatsWords * 8, res.s_k));
  g_dispatch.valid = false;
  char* dfull = dbase + (int64_t)slots * dev_slot + stats_bytes;
  double* f_ksd = (double*)dfull; double* f_ksp = (double*)(dfull + align256(npos * 8));
  double* f_cst = (double*)(dfull + 2 * align256(npos * 8)); double* f_cp = (double*)(dfull + 3 * align256(npos * 8));
  int32_t* f_run = (int32_t*)(dfull + 4 * align256(npos * 8));
  if (want_comb && prm->nb > 0) NMOD_HIP(hipMemcpyAsync(f_run, run_id, (size_t)npos * 4, hipMemcpyHostToDevice, res.s_in));

  int threads = g_hp_threads.load();
  if (threads <= 0) threads = (int)env_i64("NMOD_HOST_THREADS", try_narrow ? 8 : 4);     // (narrowing reads 8 bytes per sample it sends: more readers)
  threads = std::max(1, std::min(threads, usable_cpus()));
  if (pinned_in || total_bytes < (8 << 20)) threads = 1;
  CopyTeam team(threads);

  hipEvent_t* ev_in = call.ev_in; hipEvent_t* ev_k = call.ev_k; hipEvent_t* ev_out = call.ev_out;
  for (int s = 0; s < slots; ++s) {
    NMOD_HIP(hipEventCreateWithFlags(&ev_in[s], hipEventDisableTiming));
    NMOD_HIP(hipEventCreateWithFlags(&ev_k[s], hipEventDisableTiming));
    NMOD_HIP(hipEventCreateWithFlags(&ev_out[s], hipEventDisableTiming));
  }
  auto in_pin = [&](int s) { return res.pinned + (int64_t)s * (in_bounce + out_slab); };
  auto out_pin = [&](int s) { return in_pin(s) + in_bounce; };
  auto in_dev = [&](int s) { return dbase + (int64_t)s * dev_slot; };
  auto ws_dev = [&](int s) { return in_dev(s) + in_slot; };
  auto out_dev = [&](int s) { return ws_dev(s) + wsb; };

  // stage A: rows of chunk c -> bounce slot -> device slot (copy stream)
  auto stage_in = [&](int c) -> int {
    const int s = c % slots;
    const int64_t lo = cut[c], hi = cut[c + 1], n = hi - lo;
    if (c >= slots) {
      NMOD_HIP(hipEventSynchronize(ev_in[s]));                    // the bounce slot's previous copy has left it
      NMOD_HIP(hipStreamWaitEvent(res.s_in, ev_k[s], 0));         // the device slot's previous kernels have read it
    }
    char* pb = in_pin(s);
    const int64_t e0 = row0(lo), e1 = row1(lo);
    int64_t b0 = (row0(hi) - e0) * (int64_t)esz, b1 = (row1(hi) - e1) * (int64_t)esz;
    if (csr0) { int64_t* d = (int64_t*)(pb + o_off0); for (int64_t i = 0; i <= n; ++i) d[i] = off0[lo + i] - e0; }
    if (csr1) { int64_t* d = (int64_t*)(pb + o_off1); for (int64_t i = 0; i <= n; ++i) d[i] = off1[lo + i] - e1; }
    if (pinned_in) {
      if (o_sig0 > 0) NMOD_HIP(hipMemcpyAsync(in_dev(s), pb, (size_t)o_sig0, hipMemcpyHostToDevice, res.s_in));
      NMOD_HIP(hipMemcpyAsync(in_dev(s) + o_sig0, (const char*)sig0 + e0 * esz, (size_t)b0, hipMemcpyHostToDevice, res.s_in));
      NMOD_HIP(hipMemcpyAsync(in_dev(s) + o_sig1, (const char*)sig1 + e1 * esz, (size_t)b1, hipMemcpyHostToDevice, res.s_in));
    } else {
      if (try_narrow && team.narrow((int16_t*)(pb + o_sig0), (const double*)sig0 + e0, (size_t)(b0 / 8)) &&
          team.narrow((int16_t*)(pb + o_sig1), (const double*)sig1 + e1, (size_t)(b1 / 8))) {
        narrowed[c] = 1; b0 /= 4; b1 /= 4;
        ++g_host_stats.narrowed_chunks;
      } else {
        team.copy(pb + o_sig0, (const char*)sig0 + e0 * esz, (size_t)b0);
        team.copy(pb + o_sig1, (const char*)sig1 + e1 * esz, (size_t)b1);
      }
      // one copy when the slot is nearly full, else the parts (a short last chunk does not move the whole slot)
      if (o_sig1 - o_sig0 - b0 <= 4096) {
        NMOD_HIP(hipMemcpyAsync(in_dev(s), pb, (size_t)(o_sig1 + b1), hipMemcpyHostToDevice, res.s_in));
      } else {
        NMOD_HIP(hipMemcpyAsync(in_dev(s), pb, (size_t)(o_sig0 + b0), hipMemcpyHostToDevice, res.s_in));
        NMOD_HIP(hipMemcpyAsync(in_dev(s) + o_sig1, pb + o_sig1, (size_t)b1, hipMemcpyHostToDevice, res.s_in));
      }
    }
    NMOD_HIP(hipEventRecord(ev_in[s], res.s_in));
    g_host_stats.h2d_bytes += o_sig0 + b0 + b1;
    return NMOD_OK;
  };
  // results of chunk c: pinned slab -> the caller's arrays
  auto retire = [&](int c) -> int {
    const int s = c % slots;
    const int64_t lo = cut[c], n = cut[c + 1] - lo;
    NMOD_HIP(hipEventSynchronize(ev_out[s]));
    const char* ps = out_pin(s);
    for (int k = 0; k < 12; ++k)
      if (per_chunk[k] && hp[k]) memcpy(hp[k] + lo, ps + (int64_t)slab_idx[k] * n * 8, (size_t)n * 8);
    if (want_status) memcpy(out->status + lo, ps + (int64_t)ntr * n * 8, (size_t)n);
    return NMOD_OK;
  };
  // stage B: K1 + K2 of chunk c (compute stream), its tracks back (third stream)
  auto stage_compute = [&](int c) -> int {
    const int s = c % slots;
    const int64_t lo = cut[c], n = cut[c + 1] - lo;
    if (c >= slots) { const int r = retire(c - slots); if (r != NMOD_OK) return r; }   // frees the slot's device slab and pinned slab
    NMOD_HIP(hipStreamWaitEvent(res.s_k, ev_in[s], 0));
    nmod_out dout;
    memset(&dout, 0, sizeof(dout));
    double** dv = (double**)&dout;
    char* slab = out_dev(s);
    for (int k = 0; k < 12; ++k) if (per_chunk[k]) dv[k] = (double*)(slab + (int64_t)slab_idx[k] * n * 8);
    dout.status = (uint8_t*)(slab + (int64_t)ntr * n * 8);
    const char* di = in_dev(s);
    int r;
    if (narrowed[c]) {                              // the chunk arrived as int16 milli-units
      nmod_params dp16 = dp;
      dp16.dtype = NMOD_DTYPE_I16_MILLI;
      r = detect_device(&dp16, n, di + o_sig0, csr0 ? (const int64_t*)(di + o_off0) : nullptr, di + o_sig1,
                        csr1 ? (const int64_t*)(di + o_off1) : nullptr, nullptr, ws_dev(s), wsb, &dout);
    } else if (prm->dtype == NMOD_DTYPE_F64) {
      const int64_t t0 = row0(cut[c + 1]) - row0(lo), t1 = row1(cut[c + 1]) - row1(lo);
      const int64_t bounds[4] = {0, t0, 0, t1};
      r = detect_f64(&dp, n, di + o_sig0, csr0 ? (const int64_t*)(di + o_off0) : nullptr, di + o_sig1,
                     csr1 ? (const int64_t*)(di + o_off1) : nullptr, nullptr, ws_dev(s), wsb, &dout, bounds);
    } else {
      r = detect_device(&dp, n, di + o_sig0, csr0 ? (const int64_t*)(di + o_off0) : nullptr, di + o_sig1,
                        csr1 ? (const int64_t*)(di + o_off1) : nullptr, nullptr, ws_dev(s), wsb, &dout);
    }
    if (r != NMOD_OK) return r;
    {                                             // which K1 form took the chunk's positions (nmod_last_dispatch_stats)
      StatsArgs sa = g_dispatch.args;
      sa.acc = d_stats;
      NMOD_HIP(enqueue_dispatch_stats(sa, res.s_k));
    }
    if (want_comb) {                              // the KS track of the whole batch stays on the device for K3
      NMOD_HIP(hipMemcpyAsync(f_ksd + lo, dv[4], (size_t)n * 8, hipMemcpyDeviceToDevice, res.s_k));
      NMOD_HIP(hipMemcpyAsync(f_ksp + lo, dv[5], (size_t)n * 8, hipMemcpyDeviceToDevice, res.s_k));
    }
    NMOD_HIP(hipEventRecord(ev_k[s], res.s_k));
    NMOD_HIP(hipStreamWaitEvent(res.s_out, ev_k[s], 0));
    NMOD_HIP(hipMemcpyAsync(out_pin(s), slab, (size_t)(n * (8 * (int64_t)ntr + 1)), hipMemcpyDeviceToHost, res.s_out));
    NMOD_HIP(hipEventRecord(ev_out[s], res.s_out));
    g_host_stats.d2h_bytes += n * (8 * (int64_t)ntr + 1);
    return NMOD_OK;
  };

  // software pipeline: the rows of chunk c + 1 are on their way before chunk c's kernels are enqueued (a chunk call
  // that has to wait for the device — float64 keys, large positions — then waits beside a running copy)
  rc = stage_in(0);
  for (int c = 0; c < nchunks && rc == NMOD_OK; ++c) {
    if (c + 1 < nchunks) rc = stage_in(c + 1);
    if (rc == NMOD_OK) rc = stage_compute(c);
  }
  if (rc != NMOD_OK) return rc;
  // K3 once over the whole track (run edges and windows across chunk cuts as in the unchunked path), beside the last D2H copies
  if (want_comb) {
    // (the run ids went first on the copy stream and the compute stream has waited for every chunk's copy since)
    nmod_params cp = *prm;
    cp.stream = res.s_k;
    rc = launch_combine(&cp, res.s_k, npos, f_ksd, f_ksp, f_run, f_cst, f_cp);
    if (rc != NMOD_OK) return rc;
    NMOD_HIP(hipMemcpyAsync(out->comb_st, f_cst, (size_t)npos * 8, hipMemcpyDeviceToHost, res.s_k));
    NMOD_HIP(hipMemcpyAsync(out->comb_p, f_cp, (size_t)npos * 8, hipMemcpyDeviceToHost, res.s_k));
    g_host_stats.d2h_bytes += npos * 16;
  }
  NMOD_HIP(hipMemcpyAsync(g_dispatch.host_totals, d_stats, kStatsWords * 8, hipMemcpyDeviceToHost, res.s_k));
  for (int c = std::max(0, nchunks - slots); c < nchunks; ++c) { rc = retire(c); if (rc != NMOD_OK) return rc; }
  NMOD_HIP(hipStreamSynchronize(res.s_k));
  g_dispatch.valid = true; g_dispatch.host = true; g_dispatch.npos = npos;
  g_host_stats.chunks = nchunks; g_host_stats.slots = slots; g_host_stats.copy_threads = threads;
  g_host_stats.pinned_input = pinned_in ? 1 : 0;
  g_host_stats.chunk_positions = cap_pos;
  g_host_stats.device_bytes = (int64_t)slots * dev_slot + full_tracks;
  g_host_stats.pinned_bytes = (int64_t)pinned_need;
  return NMOD_OK;
}
