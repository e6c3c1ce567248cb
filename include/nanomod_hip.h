/* nanomod_hip.h — C ABI of the MI355X (gfx950) implementation of NanoMod's
 * per-base two-sample testing hot path.
 *
 * The reference (WGLab/NanoMod) has no FFI / plugin layer: its boundary for
 * this path is the Python function level of bin/scripts/myDetect.py
 * (SURVEY.md §8b).  Each entry point below names the reference lines it
 * replaces.  Plain pointers and sizes only; no exceptions cross the ABI; the
 * library never retains caller buffers.  Process-wide state it does keep, all
 * of it thread-safe and none of it result-affecting: per-device caches of the
 * CU count and of each kernel's occupancy (atomics, idempotent), and one HIP
 * memory pool per device, owned by the library, from which the scratch of the
 * large-position pass is allocated stream-ordered (freed slabs stay cached in
 * that pool until nmod_trim_scratch(); the device's default pool is not touched),
 * and the four tunables of nmod_host_pipeline_config (atomics; they change how a
 * host-resident batch is chunked and copied, never a result).  The library
 * reads no environment variable that changes which kernel runs: the choice
 * between kernel forms that produce the same numbers is per call
 * (NMOD_FLAG_NO_COUNTING / NMOD_FLAG_NO_COUNT_WIDE below).
 *
 * Data layout: the tested positions, in the
 * reference's iteration order (sorted (chrom,strand), then ascending
 * position, myDetect.py:421,427-431), are rows of two CSR arrays
 *     sig0[off0[i] .. off0[i+1])   samples of group 1 (--wrkBase1) at position i
 *     sig1[off1[i] .. off1[i+1])   samples of group 2 (--wrkBase2)
 * plus run_id[i]: equal ids <=> same chrom, same strand and consecutive
 * positions (restates pos_check, myDetect.py:366-371).
 * Samples are expected to be finite: NanoMod's normalised event means always
 * are.  The kernels order keys with bare v_min / v_max and pad with +inf, so a
 * position with a NaN or an infinite sample gets unspecified statistics (never
 * a fault, never another position's) — and NMOD_STATUS_NONFINITE: whenever the
 * Welch moments are computed (tests & NMOD_TEST_WELCH, or want_mstd: every call
 * the reference-shaped entry points make) a non-finite moment sets the bit at
 * no cost; NMOD_FLAG_CHECK_FINITE adds one pass over the samples that sets it
 * exactly in every mode (KS-only included).
 */
#ifndef NANOMOD_HIP_H
#define NANOMOD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NMOD_ABI_VERSION 4

/* sample dtype of sig0 / sig1 */
enum {
  NMOD_DTYPE_F32 = 0,       /* canonical: float32 (values up-cast exactly to the fp64 the reference sees) */
  NMOD_DTYPE_I16_MILLI = 1, /* int16 = round(norm_mean*1000): NanoMod's Events are 3-dp rounded
                               (myRefBaseSignalAnnotation.py:1108); value = k/1000.0 in fp64 */
  NMOD_DTYPE_F64 = 2        /* float64 as the reference holds it (lists of numpy.float64, myDetect.py:124).  The rank
                               statistics depend on the order of a position's samples only, so the library gives every
                               POSITION order- and tie-preserving float32 keys on the device: the samples themselves if all
                               of them are float32-exact, k if all are k/1000.0 with |k| <= 2^24 (NanoMod's 3-dp Events),
                               else their float32 roundings — monotone, so the only possible damage is a false tie between
                               two different doubles; the kernels report the positions whose keys tie and those (rare for
                               real-valued signals) are redone on the float64 samples with 64-bit keys (big_rank.hpp).  The
                               Welch moments always come from the float64 samples.  One extra pass over the samples, one
                               host round trip (the count of positions to redo).  ~3.5e8 positions/s KS-only, 2e8 with all
                               tests at 200 v 200; a batch made of doubles with exact ties that are neither float32-exact
                               nor on the 0.001 grid runs at the 64-bit-key rate, ~4e7 positions/s. */
};

/* where the caller's buffers live */
enum { NMOD_MEM_HOST = 0, NMOD_MEM_DEVICE = 1 };

/* --testMethod (NanoMod.py:359; consumed at myDetect.py:392,395,443) */
enum { NMOD_METHOD_KS = 0, NMOD_METHOD_STOUFFER = 1, NMOD_METHOD_FISHER = 2 };

/* which per-position tests to compute.  The reference always computes all
 * three (myDetect.py:331-343); the mask exists for the KS+combine benchmark
 * configuration of BASELINE.json. */
enum { NMOD_TEST_KS = 1, NMOD_TEST_MWU = 2, NMOD_TEST_WELCH = 4, NMOD_TEST_ALL = 7 };

/* per-position status bits (data-dependent conditions that make the
 * reference raise or return NaN; reported instead of aborting) */
enum {
  NMOD_STATUS_MWU_ALL_IDENTICAL = 1, /* scipy 1.2.1 mannwhitneyu raises ValueError (T == 0), uncaught at myDetect.py:331 */
  NMOD_STATUS_T_NAN = 2,             /* zero variance in both groups: ttest_ind returns (nan, nan), myDetect.py:335 */
  NMOD_STATUS_EMPTY = 4,             /* n0 == 0 or n1 == 0 (cannot occur after mfilter_coverage, myDetect.py:301-314) */
  NMOD_STATUS_TOO_LARGE = 8,         /* more samples in a group than the max_n0 / max_n1 the caller promised, or than NMOD_MAX_RANKED:
                                        the position is skipped, its outputs are NaN, the rest of the batch is computed */
  NMOD_STATUS_NONFINITE = 16         /* a NaN or infinite sample (see the header comment): the position's statistics are unspecified */
};

/* return codes */
enum {
  NMOD_OK = 0,
  NMOD_ERR_INVALID_ARG = -1,
  NMOD_ERR_HIP = -2,          /* a HIP runtime call failed; see nmod_strerror */
  NMOD_ERR_TOO_LARGE = -3,    /* nmod_describe_dispatch / nmod_downsample_ks: a group beyond NMOD_MAX_RANKED (nmod_detect_batch reports
                                 such positions per position: NMOD_STATUS_TOO_LARGE) */
  NMOD_ERR_WORKSPACE = -4,    /* workspace missing or too small (device-memory mode) */
  NMOD_ERR_NO_DEVICE = -5
};

#define NMOD_MAX_GROUP 2048   /* largest group the wave-resident kernels sort (both groups in all-tests mode, the
                                 smaller one in KS-only mode); positions beyond it take the workgroup-per-position
                                 pass (big_rank.hpp), slower but unlimited up to NMOD_MAX_RANKED */
#define NMOD_MAX_RANKED 65535 /* max samples per group per position in any mode (16-bit ranks, 32-bit KS numerator); a position
                                 beyond it gets NMOD_STATUS_TOO_LARGE */
#define NMOD_MAX_NB 64        /* max --neighborPvalues */

typedef struct nmod_params {
  int32_t struct_size;    /* sizeof(nmod_params), for ABI evolution */
  int32_t device;         /* HIP device ordinal */
  void*   stream;         /* hipStream_t to launch on (NULL = default stream) */
  int32_t memspace;       /* NMOD_MEM_*: where sig/off/run_id/out pointers live */
  int32_t dtype;          /* NMOD_DTYPE_* */
  int32_t tests;          /* NMOD_TEST_* mask */
  int32_t method;         /* NMOD_METHOD_* */
  int32_t nb;             /* --neighborPvalues (NanoMod.py:357), window = 2*nb+1 */
  int32_t want_mstd;      /* --mstd: also mean / std (ddof=0) per group (myDetect.py:437-438) */
  double  weights_dif;    /* --WeightsDif (NanoMod.py:358; weights myDetect.py:396-400) */
  int64_t stride0;        /* >0: fixed-stride layout, off0 may be NULL: n0 = stride0 for every position */
  int64_t stride1;        /* >0: same for group 2 */
  int32_t max_n0;         /* upper bound of samples per position in group 1 (0 = unknown: the library
                             reduces off0 on the device and synchronises once to read it) */
  int32_t max_n1;         /* same for group 2 */
  void*   timer;          /* optional nmod_evtimer handle: HIP events are recorded around each kernel */
  int32_t flags;          /* NMOD_FLAG_* (0 = the reference's numbers bit for bit wherever it defines them) */
  int32_t reserved;       /* 0 */
} nmod_params;

/* tests == NMOD_TEST_KS only (a mode the reference never runs: getKStest always computes all three tests): report D as the
 * correctly rounded exact rational max|c0*n1 - c1*n0| / (n0*n1) instead of ks_2samp's float form max|fl(c0/n0) - fl(c1/n1)|
 * (they differ by <= 2 ulp, <= 4.5e-16 absolute; p-values agree to ~1e-15 relative).  Skips the pass that evaluates the float
 * form at the pooled points reaching the integer maximum (~10 % of the KS-only kernel).  Ignored with any other test in the mask. */
#define NMOD_FLAG_KS_RATIONAL_D 1
/* One extra pass over the samples (float32 / float64 input; ~0.2 ms per GB) that sets NMOD_STATUS_NONFINITE for every position
 * holding a NaN or an infinite sample, in every mode.  Without it the bit comes from the Welch moments alone (free, whenever they
 * are computed): that catches every NaN and -inf, and +inf except in the group a kernel form sorts with +inf pads. */
#define NMOD_FLAG_CHECK_FINITE 2
/* Kernel-form switches for A/B measurements and parity tests; the numbers are the same with and without them.  Event-like
 * rows (every sample on the milli-unit grid, a position's samples within a window of 2 048 milli-units apart from a few
 * outliers) are taken by counting forms of K1 instead of sorting forms when a device-side probe finds a size class of the
 * batch event-like.  NMOD_FLAG_NO_COUNTING keeps every position on the sorting forms;
 * NMOD_FLAG_NO_COUNT_WIDE only those outside the 256-capacity class. */
#define NMOD_FLAG_NO_COUNTING 4
#define NMOD_FLAG_NO_COUNT_WIDE 8
/* NMOD_MEM_HOST with NMOD_DTYPE_F64: the threads that fill the pinned bounce slots write a chunk as int16 milli-units where every
 * sample of it is k / 1000.0 with |k| <= 32 767 (what NanoMod's stored events are: myRefBaseSignalAnnotation.py:1108) — 2 bytes
 * per sample over PCIe instead of 8, the chunk then runs as NMOD_DTYPE_I16_MILLI; any other chunk is sent as float64.  The rank
 * statistics are the same bit for bit, the Welch moments come from exact integer sums instead of the two-pass float64 sums
 * (both within 1e-11 of the reference's t).  This flag sends every chunk as float64 (A/B, parity tests). */
#define NMOD_FLAG_NO_HOST_NARROW 16

/* Caller-allocated SoA outputs, npos elements each; a NULL member is skipped.
 * One (stat, p) pair per test = the tuples getKStest returns
 * (myDetect.py:363) after the m_min_float / m_max_float clamps (:317-325),
 * then the combined pair appended by combin_pvalues (:373-377,403-406). */
typedef struct nmod_out {
  double* mwu_u;   double* mwu_p;     /* myDetect.py:331-333 */
  double* t_t;     double* t_p;       /* myDetect.py:335-337 */
  double* ks_d;    double* ks_p;      /* myDetect.py:341-343 */
  double* comb_st; double* comb_p;    /* myDetect.py:379-414 (not written when method == KS) */
  double* mean0;   double* std0;      /* myDetect.py:438 (want_mstd) */
  double* mean1;   double* std1;
  uint8_t* status;
} nmod_out;

int nmod_abi_version(void);
int nmod_device_count(void);                 /* number of HIP devices (0 if none) */
const char* nmod_strerror(int rc);

/* Device scratch needed by nmod_detect_batch for `npos` positions (bytes). */
int64_t nmod_workspace_bytes(const nmod_params* prm, int64_t npos);

/* Replaces the two hot loops of mtest2 (myDetect.py:427-436 per-position
 * getKStest, :443 combin_pvalues).  NMOD_MEM_DEVICE: all pointers are device
 * pointers, `workspace` must hold nmod_workspace_bytes(), everything is
 * enqueued on prm->stream and the call returns without synchronising — except
 * for one host round trip each when (i) max_n0 / max_n1 are unknown for CSR
 * inputs, (ii) the maxima allow groups beyond NMOD_MAX_GROUP (the scratch of the
 * large-position pass is sized from the classifier's totals and allocated
 * stream-ordered), (iii) dtype is NMOD_DTYPE_F64 (the probe's verdict).
 * NMOD_MEM_HOST (what a drop-in mtest2 hands over: everything on this path is host
 * memory in the reference, myDetect.py:416-445): pointers are host memory, workspace
 * may be NULL, prm->stream is not used and the call returns after the results are
 * back.  The batch is cut into chunks of positions that go through pinned bounce
 * slots (skipped for arrays that are already page-locked) and three library-owned
 * streams: the H2D copy of chunk k+1, K1 + K2 of chunk k and the D2H copy of chunk
 * k-1 overlap; the KS track stays on the device and ONE K3 runs over the whole
 * batch at the end, so chunk cuts never change a window.  Device footprint:
 * slots x (chunk + workspace + results) + 36 B per position, from the library's
 * pool; the pinned ring and the streams are cached per device until
 * nmod_trim_scratch().  See nmod_host_pipeline_config / nmod_last_host_stats. */
int nmod_detect_batch(const nmod_params* prm, int64_t npos,
                      const void* sig0, const int64_t* off0,
                      const void* sig1, const int64_t* off1,
                      const int32_t* run_id,
                      void* workspace, int64_t workspace_bytes,
                      nmod_out* out);

/* Tunables of the NMOD_MEM_HOST pipeline, process-wide; 0 keeps / restores the default.  chunk_bytes: sample bytes per
 * chunk (default min(64 MiB, batch / 32), at least 1 MiB; env NMOD_HOST_CHUNK_BYTES); slots: ring depth 2..8 (default 3; env
 * NMOD_HOST_SLOTS); threads: host threads filling a bounce slot (default 4, capped by the cgroup CPU quota; env
 * NMOD_HOST_THREADS); mode: 0 = copy straight from arrays that are page-locked, bounce the rest; 2 = always bounce. */
int nmod_host_pipeline_config(int64_t chunk_bytes, int32_t slots, int32_t threads, int32_t mode);

/* What the calling thread's last NMOD_MEM_HOST nmod_detect_batch did. */
typedef struct nmod_host_stats {
  int64_t chunks;            /* chunks the batch was cut into */
  int64_t slots;             /* ring depth used */
  int64_t copy_threads;      /* host threads that filled the bounce slots (1 when the input was page-locked) */
  int64_t pinned_input;      /* 1: sig0 / sig1 were page-locked and copied from where they are */
  int64_t chunk_positions;   /* positions of the largest chunk */
  int64_t device_bytes;      /* device memory held during the call (ring + per-batch tracks) */
  int64_t pinned_bytes;      /* pinned host ring */
  int64_t h2d_bytes;         /* bytes copied host -> device */
  int64_t d2h_bytes;         /* bytes copied device -> host */
  int64_t narrowed_chunks;   /* NMOD_DTYPE_F64: chunks sent as int16 milli-units (see NMOD_FLAG_NO_HOST_NARROW) */
} nmod_host_stats;
int nmod_last_host_stats(nmod_host_stats* st);

/* Which K1 form computed the positions of the calling thread's last nmod_detect_batch.  The forms produce the same numbers
 * (tests/test_gpu_parity.py runs batches through both and compares); which one runs is decided on the device — size classes,
 * and for the counting forms a probe per class plus a per-position check — so the split is a property of the data the caller
 * can only learn here.  NMOD_MEM_DEVICE: the counters are reduced on request from facts the call left in `workspace` (one
 * small kernel on the call's stream, one synchronisation): ask before the workspace is reused or freed.  NMOD_MEM_HOST: they
 * were read back with the results.  NMOD_ERR_INVALID_ARG when the thread has made no such call. */
typedef struct nmod_dispatch_stats {
  int64_t positions;        /* positions of the batch */
  int64_t ks_rank;          /* ks_rank_kernel: KS only, the smaller group sorted */
  int64_t rank_hist;        /* rank_hist_kernel: all tests, both groups sorted (groups of similar size) */
  int64_t rank_hist_wide;   /* rank_hist_kernel, WIDE form: the smaller group sorted, the larger one streamed */
  int64_t rank_pair;        /* rank_pair_kernel: all tests, both groups beyond 256 samples and of different capacity */
  int64_t rank_count;       /* rank_count_kernel: the counting form of the 256-capacity class (event-like rows) */
  int64_t rank_count_wide;  /* rank_count_wide_kernel: the counting form for any coverage (event-like rows); rank_count_value_kernel's too */
  int64_t big;              /* big_rank_kernel / big_hist_kernel: a group beyond NMOD_MAX_GROUP */
  int64_t skipped;          /* no K1 form (NMOD_STATUS_EMPTY / NMOD_STATUS_TOO_LARGE) */
  int64_t count_tried;      /* positions of the classes whose probe let a counting form run */
  int64_t count_rejected;   /* ... that the counting form handed on to the class's sorting form (counted there above) */
  int64_t f64_redo;         /* NMOD_DTYPE_F64: positions done again on 64-bit keys (their first form counts them as well) */
  int64_t reserved[4];
} nmod_dispatch_stats;
int nmod_last_dispatch_stats(nmod_dispatch_stats* st);

/* "arch=gfx950 abi=3 ... | <translation unit>: NMOD_SKIP=0 NMOD_EXP=0 ..." — the value of every experiment macro
 * (phase-skip and variant switches of the kernel headers) in each translation unit of THIS binary.  A product build
 * reports NMOD_SKIP=0 NMOD_EXP=0 everywhere (tests/test_abi_and_host.py). */
const char* nmod_build_info(void);

/* KS statistic: ks_d is ks_2samp's own float form max|fl(c0/n0) - fl(c1/n1)| bit for bit in every mode (myDetect.py:341 ->
 * scipy 1.2.1), tests == NMOD_TEST_KS included: the kernels find the exact integer maximum of |c0*n1 - c1*n0| and evaluate the
 * float form for the pooled points that reach it. */

/* Name of the K1 kernel instance a position with n0 / n1 samples is dispatched to under prm's dtype / tests / method
 * (e.g. "ks_rank_kernel<16,16,f32>"), from the same size-class functions the dispatcher uses: what bench.py prints as
 * roofline.kernel and what the rocprofv3 kernel trace shows.  No device work. */
int nmod_describe_dispatch(const nmod_params* prm, int64_t n0, int64_t n1, char* buf, int32_t buflen);

/* Returns the slabs cached in the library's scratch pool of `device` to the driver (see the header comment), and frees the
 * pinned ring + streams the NMOD_MEM_HOST pipeline caches for it. */
int nmod_trim_scratch(int32_t device);

/* Replaces combin_pvalues / get_combin_pvalue on a whole KS track
 * (myDetect.py:373-414).  ks_d is only read when nb == 0 (:413). */
int nmod_combine_track(const nmod_params* prm, int64_t npos,
                       const double* ks_d, const double* ks_p, const int32_t* run_id,
                       double* comb_st, double* comb_p);

/* Replaces the down-sampling branch of getKStest (myDetect.py:345-361; taken when --coverages > 0 and a group exceeds the
 * threshold) for `nflag` positions of a HOST-resident CSR batch (prm->memspace == NMOD_MEM_HOST): `iters` (--downsampling, 100)
 * times, a group of position positions[i] with more than cov[i] samples is resampled WITH replacement to cov[i] samples
 * (np.random.choice(x, cov)), KS runs on each resample, and the (D, p) pair at index int(iters * quantile)
 * (--downsampling_quantile, 0.25) of the p-sorted resamples is written to ks_d[i] / ks_p[i].  The reference draws from numpy's
 * unseeded global generator; here the draws are a counter-based function of (seed, i, iteration, group, draw) on the device —
 * reproducible, statistically equivalent, not bit-comparable (SURVEY.md 8a row A3').  The resampled rows are materialised in
 * HBM chunk by chunk (2^27 samples) and go through the same KS kernel as everything else.  Synchronises. */
int nmod_downsample_ks(const nmod_params* prm, int64_t nflag, const void* sig0, const int64_t* off0, const void* sig1, const int64_t* off1,
                       const int64_t* positions, const int64_t* cov, int32_t iters, double quantile, uint64_t seed,
                       double* ks_d, double* ks_p);

/* Benchmark input generator (no reference counterpart).
 * Counter-based and integer-only, so tests restate it bit-exactly on the CPU:
 *   h = mix64(seed, group, pos, read);  s = sum of the four 16-bit fields of h;
 *   x = float(s - 131070) * (1/37837.2f)  [ + shift  at planted positions of group 1 ]
 * Fills sig[(pos - pos_begin) * n_per_pos + read] for pos in [pos_begin, pos_begin+npos).
 * A position is planted iff plant_period > 0 and (pos % plant_period) is 0, 1 or plant_period-1. */
int nmod_synth_fill(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos,
                    int32_t group, int32_t n_per_pos, int64_t plant_period, float plant_shift,
                    void* sig_out);

/* The same generator for ragged rows (BASELINE.json configs[4]): sample `read` of position pos_begin + i goes to
 * sig_out[off[i] + read], read < off[i+1] - off[i]; `off` is a DEVICE array of npos + 1 element offsets into sig_out.
 * float32 or int16 milli-unit output (prm->dtype). */
int nmod_synth_fill_csr(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos,
                        int32_t group, const int64_t* off, int64_t plant_period, float plant_shift,
                        void* sig_out);

/* Synthetic EVENT rows the way NanoMod stores them (benchmark input, no reference counterpart): every position has a signal
 * LEVEL shared by both groups (+-3 normalised units: the k-mer under the pore) and each read spreads around it with standard
 * deviation spread_milli / 1000 units; values sit on the 3-decimal grid (myRefBaseSignalAnnotation.py:1108 rounds norm_mean to
 * 3 decimals), so most of a position's samples tie with another one — the unit-variance rows of nmod_synth_fill tie ~11 times
 * per 200 v 200 position.  Integer-only up to the last quotient, so tests restate it bit for bit:
 *   level(pos) = (mix64(seed ^ 0xA5A5A5A5DEADBEEF, pos, 0, 0) >> 40) % 6001 - 3000                          [milli-units]
 *   z = (sum of the four 16-bit fields of mix64(seed, group, pos, read)) - 131070     (as nmod_synth_fill; sd 37837.2)
 *   k = level + floor((2 z spread_milli + 37837) / 75674)  [+ plant_shift_milli at planted positions of group 1]
 *   o = mix64(seed ^ 0x0DDBA11C0FFEE123, group, pos, read);  if ((o >> 20) % 1000 < outlier_permille) k = (o >> 32) % 10001 - 5000
 *       (a mis-segmented event: uniform over the +-5 unit clip range of the raw normalisation, myRefBaseSignalAnnotation.py:251-259)
 *   |k| <= 32767
 * int16 output: k.  float32 output: (float)((double)k / 1000.0), the float32 image of the stored 3-decimal value.
 * Rows: n_per_pos > 0 fixed stride (off ignored), else `off` = DEVICE array of npos + 1 element offsets into sig_out.
 * 0 <= spread_milli <= 8000, 0 <= outlier_permille <= 1000. */
int nmod_synth_fill_events(const nmod_params* prm, uint64_t seed, int64_t pos_begin, int64_t npos,
                           int32_t group, int32_t n_per_pos, const int64_t* off, int64_t plant_period,
                           int32_t plant_shift_milli, int32_t spread_milli, int32_t outlier_permille, void* sig_out);

/* HIP-event timer: records (start, stop) around every kernel the library
 * launches while prm->timer points to it; read it after synchronising. */
enum { NMOD_KERNEL_RANK_STATS = 0, NMOD_KERNEL_FINALIZE = 1, NMOD_KERNEL_COMBINE = 2,
       NMOD_KERNEL_SYNTH = 3, NMOD_KERNEL_COUNT = 4 };
int nmod_evtimer_create(int32_t capacity_per_kernel, void** timer);
int nmod_evtimer_reset(void* timer);
int nmod_evtimer_read(void* timer, int32_t kernel, double* total_ms, int32_t* launches);
int nmod_evtimer_destroy(void* timer);

/* Replaces save_test's table loop (myDetect.py:522-538) for array-shaped results: writes one line per
 * position, '%s %s %d %s %d %d %.3f %.3E %.3f %.3E %.3f %.3E' = chrom strand pos+1 base n0 n1 U pU t pt D pKS,
 * then ' %.3f %.3E' with the combined pair iff with_comb, then '\n'; non-finite values print as Python
 * prints them ('inf', '-inf', 'nan').  Host-only (no device work).  chrom_id[i] indexes chrom_names
 * (NUL-separated, n_chroms entries); strand[i] and base[i] are single characters.  Returns NMOD_OK or
 * NMOD_ERR_INVALID_ARG (also when the file cannot be opened). */
int nmod_write_sign_test(const char* path, int64_t npos, const int32_t* chrom_id, const char* chrom_names,
                         int32_t n_chroms, const char* strand, const int64_t* pos0, const char* base,
                         const int32_t* n0, const int32_t* n1, const double* mwu_u, const double* mwu_p,
                         const double* t_t, const double* t_p, const double* ks_d, const double* ks_p,
                         const double* comb_st, const double* comb_p, int32_t with_comb);

/* Test hook for the float64 -> int16 narrowing of the host-resident entry: out[i] = k where v[i] == k / 1000.0 with |k| <= 32 767;
 * returns 1 when every one of the n values narrowed, 0 when one refused (out is then unspecified), negative on bad arguments.
 * Host-only. */
int nmod_narrow_probe(const double* v, int64_t n, int16_t* out);

/* Test hook for the table writer's number formats: v[i] formatted as '%.3f' (sci = 0) or '%.3E' (sci = 1) the way
 * nmod_write_sign_test does, NUL-separated, into out (capacity cap bytes; at most 420 bytes per value). */
int nmod_format_probe(const double* v, int64_t n, int32_t sci, char* out, int64_t cap);

/* Replaces the ranking of the result records (myDetect.py:447-462): order_out[i] = index of the i-th record of
 * sorted(records, key = (key_primary, key_second, key_third)) — Python's stable tuple sort, ascending, -0.0 tied
 * with 0.0, NaN last — reversed as a whole when `descending` (rankUse == 'st': the reference reverses the sorted
 * list).  The reference's keys are (combined p, KS p, MWU p) or the three statistics.  Device radix sort (radix_sort.hpp),
 * three stable passes; synchronises before returning. */
int nmod_rank_order(const nmod_params* prm, int64_t npos, const double* key_primary, const double* key_second,
                    const double* key_third, int32_t descending, int32_t* order_out);

/* The stable ascending order of n signed 64-bit keys: order_out[i] = index of the i-th smallest key, equal keys in index
 * order (what numpy.argsort(kind='stable') returns).  keys / order_out: host or device memory by prm->memspace.  Behind the
 * grouping of events by (chrom, strand, position) that replaces getGenomeEvents' dict inserts for the simulation loops
 * (mySimulat2.py:127-171; nanomod_amd/simulate.py).  Synchronises before returning. */
int nmod_argsort_keys(const nmod_params* prm, int64_t n, const int64_t* keys, int32_t* order_out);

/* Replaces the window ranking of --RegionRankbyST 1 (myDetect.py:463-515) on array-shaped records in the
 * reference's record order (sorted (chrom, strand), ascending position).  strand_lo[i] / strand_hi[i]: index of the
 * first / last record of record i's (chrom, strand); value[i]: the p-value or statistic the ranking uses
 * (record[sorted_ind][use_pind]); w: the window half-width AFTER the reference's in-place increment (window + 1);
 * movesize: 1 when WindOvlp == 1, else w; na: base filter ('\0' = none).  Writes the indices of the ranked window
 * centres to ranked_out (capacity npos) and their number to *n_ranked.  Host memory only; synchronises. */
int nmod_region_rank(const nmod_params* prm, int64_t npos, const int32_t* strand_lo, const int32_t* strand_hi,
                     const int64_t* pos, const char* base, const double* value, int32_t w, int32_t movesize,
                     char na, double percentile, int32_t wind_ovlp, int32_t* ranked_out, int64_t* n_ranked);

/* ---- position shards across the GPUs of a node without any host framework (BASELINE.json north_star: "an RCCL
 * all-gather over xGMI to reassemble the per-base p-value track").  The reference has no counterpart (one CPU process).  One
 * process (or thread) per GPU computes a contiguous block of positions (+- nb recomputed neighbours) with
 * nmod_detect_batch and calls nmod_allgather_tracks: rank r's block of `block_len` doubles of every track lands at
 * full[t] + r * block_len on every rank (blocks of equal length: pad the last one).  The library does not link RCCL: it binds
 * librccl.so at the first call — the copy the process has already loaded if there is one (a PyTorch process: torch's), else
 * librccl.so.1 from the loader path — and returns NMOD_ERR_NO_RCCL when there is none.
 *   rank 0:      nmod_comm_unique_id(id)  -> hand the NMOD_COMM_ID_BYTES bytes to the other ranks (file, socket, MPI, ...)
 *   every rank:  nmod_comm_init_rank(id, nranks, rank, device, &comm)           (collective: all ranks must call it)
 *                nmod_allgather_tracks(comm, stream, block_len, ntracks, local, full)   enqueued on `stream`, no synchronisation
 *                nmod_comm_destroy(comm) */
#define NMOD_COMM_ID_BYTES 128
#define NMOD_ERR_NO_RCCL (-6)      /* librccl.so could not be bound */
#define NMOD_ERR_RCCL (-7)         /* an RCCL call failed; see nmod_strerror */
typedef struct nmod_comm nmod_comm;
int nmod_comm_unique_id(void* id_out);
int nmod_comm_init_rank(const void* unique_id, int32_t nranks, int32_t rank, int32_t device, nmod_comm** comm_out);
int nmod_allgather_tracks(nmod_comm* comm, void* stream, int64_t block_len, int32_t ntracks,
                          const double* const* local, double* const* full);
int nmod_comm_destroy(nmod_comm* comm);

/* Lane-permutation self test of the wave primitives the sort is built from
 * (runs tiny kernels; returns NMOD_OK or the number of the first failing primitive). */
int nmod_selftest(int32_t device);

#ifdef __cplusplus
}
#endif
#endif /* NANOMOD_HIP_H */
