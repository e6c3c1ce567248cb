/* nanomod_amd._hostwalk — host-side helper of detect.build_csr (CPython C API; no device code).
 *
 * The reference keeps a read group as dict[(chrom, strand)][pos] -> list of numpy.float64 (myDetect.py:124, :569-572);
 * mtest2's drop-in has to turn millions of those small per-position sequences into one CSR array.  numpy.concatenate
 * spends ~1.5 us per row on that; this walk spends ~0.1 us: C-contiguous float64 ndarrays are copied, anything else
 * goes through the sequence protocol and float().
 *
 *   flatten(rows, out) -> number of values written
 *     rows: list or tuple of per-position sequences (ndarray, list, tuple, ...)
 *     out:  writable C-contiguous float64 buffer with room for all of them (ValueError otherwise)
 *   filter_coverage(norm, base, min_cov), join_strand(norm0, norm1, base0, base1): mfilter_coverage's inner loop and
 *     mtest2's loop header for one (chrom, strand), one C pass over the dicts each (see below)
 *   join_strand_plan(norm0, norm1, base0, base1) + copy_plan(plan, out0, start0, out1, start1) (round 6): the same join in two
 *     steps — first the positions and sizes of every (chrom, strand), then the rows of all of them straight into ONE pair of CSR
 *     arrays (no per-strand arrays to concatenate: that copy was 0.17 s of build_csr's 0.23 s at 460 000 x 200 v 200), as float64
 *     or — where every sample is k / 1000.0 with |k| <= 32 767, what stored events are — narrowed to int16 milli-units on the way
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <math.h>

/* x == k / 1000.0 with |k| <= 32 767?  Division-free: q = fl(k / 1000) by one Newton step on k * fl(1 / 1000), correctly rounded
 * for every such k (the same test as csrc/host_pipeline.hpp's narrow_f64_to_i16; tests/test_abi_and_host.py checks both exhaustively) */
static inline int narrow_one(double x, short* out) {
  if (!(fabs(x) <= 33.0)) return 0;
  const double k = nearbyint(x * 1000.0);
  const double q0 = k * 0.001;
  const double q = fma(fma(-q0, 1000.0, k), 0.001, q0);
  *out = (short)(int)k;
  return q == x && fabs(k) <= 32767.0;
}
static int narrow_scalar(const double* src, short* dst, Py_ssize_t n) {
  for (Py_ssize_t i = 0; i < n; ++i) if (!narrow_one(src[i], dst + i)) return 0;
  return 1;
}
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2,fma"))) static int narrow_avx2(const double* src, short* dst, Py_ssize_t n) {
  const __m256d k1000 = _mm256_set1_pd(1000.0), r = _mm256_set1_pd(0.001), lim = _mm256_set1_pd(32767.0);
  const __m256d absmask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffll));
  __m256d good = _mm256_castsi256_pd(_mm256_set1_epi64x(-1));
  Py_ssize_t i = 0;
  for (; i + 8 <= n; i += 8) {
    __m128i kk[2];
    for (int h = 0; h < 2; ++h) {
      const __m256d x = _mm256_loadu_pd(src + i + 4 * h);
      const __m256d k = _mm256_round_pd(_mm256_mul_pd(x, k1000), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
      const __m256d q0 = _mm256_mul_pd(k, r);
      const __m256d q = _mm256_fmadd_pd(_mm256_fnmadd_pd(q0, k1000, k), r, q0);
      good = _mm256_and_pd(good, _mm256_and_pd(_mm256_cmp_pd(q, x, _CMP_EQ_OQ), _mm256_cmp_pd(_mm256_and_pd(k, absmask), lim, _CMP_LE_OQ)));
      kk[h] = _mm256_cvtpd_epi32(k);
    }
    _mm_storeu_si128((__m128i*)(dst + i), _mm_packs_epi32(kk[0], kk[1]));
  }
  if (i < n) {                                           /* the last < 8 samples: the same vector code on a zero-padded copy (rows of ~20 samples
                                                            would otherwise spend their time in libm's scalar fma) */
    double pad[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    short out8[8];
    memcpy(pad, src + i, (size_t)(n - i) * 8);
    __m128i kk[2];
    for (int h = 0; h < 2; ++h) {
      const __m256d x = _mm256_loadu_pd(pad + 4 * h);
      const __m256d k = _mm256_round_pd(_mm256_mul_pd(x, k1000), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
      const __m256d q0 = _mm256_mul_pd(k, r);
      const __m256d q = _mm256_fmadd_pd(_mm256_fnmadd_pd(q0, k1000, k), r, q0);
      good = _mm256_and_pd(good, _mm256_and_pd(_mm256_cmp_pd(q, x, _CMP_EQ_OQ), _mm256_cmp_pd(_mm256_and_pd(k, absmask), lim, _CMP_LE_OQ)));
      kk[h] = _mm256_cvtpd_epi32(k);
    }
    _mm_storeu_si128((__m128i*)out8, _mm_packs_epi32(kk[0], kk[1]));
    memcpy(dst + i, out8, (size_t)(n - i) * 2);
  }
  return _mm256_movemask_pd(good) == 0xf;
}
#endif
static int narrow_row(const double* src, short* dst, Py_ssize_t n) {
#if defined(__x86_64__)
  static int have = -1;
  if (have < 0) have = (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) ? 1 : 0;
  if (have) return narrow_avx2(src, dst, n);
#endif
  return narrow_scalar(src, dst, n);
}

static int put_generic(PyObject* row, double* dst, Py_ssize_t room, Py_ssize_t* wrote) {
  PyObject* fast = PySequence_Fast(row, "rows must be sequences");
  if (!fast) return -1;
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  if (n > room) { Py_DECREF(fast); PyErr_SetString(PyExc_ValueError, "out is too small"); return -1; }
  PyObject** items = PySequence_Fast_ITEMS(fast);
  for (Py_ssize_t i = 0; i < n; ++i) {
    const double v = PyFloat_AsDouble(items[i]);            /* numpy.float64 is a float; ints and 0-d arrays convert */
    if (v == -1.0 && PyErr_Occurred()) { Py_DECREF(fast); return -1; }
    dst[i] = v;
  }
  Py_DECREF(fast);
  *wrote = n;
  return 0;
}

static PyObject* flatten(PyObject* self, PyObject* args) {
  PyObject* rows; PyObject* out_obj; Py_buffer out;
  (void)self;
  if (!PyArg_ParseTuple(args, "OO", &rows, &out_obj)) return NULL;
  if (PyObject_GetBuffer(out_obj, &out, PyBUF_WRITABLE | PyBUF_FORMAT | PyBUF_ND) < 0) return NULL;   /* (with the format: "w*" hides the element type) */
  PyObject* fast_rows = PySequence_Fast(rows, "rows must be a list or tuple");
  if (!fast_rows) { PyBuffer_Release(&out); return NULL; }
  if (out.itemsize != 8 || out.len % 8 != 0 || out.ndim != 1 || (out.format && strcmp(out.format, "d") != 0)) {
    Py_DECREF(fast_rows); PyBuffer_Release(&out);
    PyErr_SetString(PyExc_ValueError, "out must be a float64 buffer"); return NULL;
  }
  double* dst = (double*)out.buf;
  const Py_ssize_t cap = out.len / 8;
  Py_ssize_t at = 0;
  const Py_ssize_t nrows = PySequence_Fast_GET_SIZE(fast_rows);
  PyObject** r = PySequence_Fast_ITEMS(fast_rows);
  for (Py_ssize_t i = 0; i < nrows; ++i) {
    Py_ssize_t wrote = 0;
    /* the rows are separate heap objects: the walk is a chain of cache misses unless the headers and the data of the
     * rows ahead are requested early */
    if (i + 16 < nrows) __builtin_prefetch(r[i + 16]);
    if (i + 8 < nrows) {
      PyObject* o = r[i + 8];
      if (PyList_CheckExact(o)) __builtin_prefetch(((PyListObject*)o)->ob_item);
      else if (PyArray_Check(o)) { __builtin_prefetch(PyArray_DATA((PyArrayObject*)o)); __builtin_prefetch(PyArray_DIMS((PyArrayObject*)o)); }
    }
    if (PyList_CheckExact(r[i]) || PyTuple_CheckExact(r[i])) {
      if (put_generic(r[i], dst + at, cap - at, &wrote) < 0) goto fail;
    } else if (PyArray_Check(r[i]) && PyArray_NDIM((PyArrayObject*)r[i]) != 1) {
      /* len(row) of a 2-D row counts its first axis: the caller's sample counts and this walk would disagree */
      PyErr_SetString(PyExc_ValueError, "every per-position row must be one-dimensional"); goto fail;
    } else if (PyArray_Check(r[i]) && PyArray_TYPE((PyArrayObject*)r[i]) == NPY_DOUBLE && PyArray_IS_C_CONTIGUOUS((PyArrayObject*)r[i])) {
      PyArrayObject* a = (PyArrayObject*)r[i];               /* (the buffer protocol costs ~0.8 us per array: numpy fills in a format string) */
      wrote = (Py_ssize_t)PyArray_SIZE(a);
      if (wrote > cap - at) { PyErr_SetString(PyExc_ValueError, "out is too small"); goto fail; }
      memcpy(dst + at, PyArray_DATA(a), (size_t)wrote * 8);
    } else {
      if (put_generic(r[i], dst + at, cap - at, &wrote) < 0) goto fail;
    }
    at += wrote;
  }
  Py_DECREF(fast_rows); PyBuffer_Release(&out);
  return PyLong_FromSsize_t(at);
fail:
  Py_DECREF(fast_rows); PyBuffer_Release(&out);
  return NULL;
}


/* ---------------------------------------------------------------------------------------------------------------
 * The reference's per-strand structure: dict[pos] -> row for each dataset, dict[pos] -> base beside it
 * (myDetect.py:569-572).  Both functions below read the dicts in STORAGE order with PyDict_Next — sequential memory, no
 * hashing; the reference fills them position by position, so storage order is ascending position for any real input
 * (checked; anything else is sorted first).
 */
static Py_ssize_t row_len(PyObject* row) {
  if (PyList_CheckExact(row)) return PyList_GET_SIZE(row);
  if (PyTuple_CheckExact(row)) return PyTuple_GET_SIZE(row);
  if (PyArray_Check(row)) {
    PyArrayObject* a = (PyArrayObject*)row;
    if (PyArray_NDIM(a) != 1) { PyErr_SetString(PyExc_ValueError, "every per-position row must be one-dimensional"); return -1; }
    return (Py_ssize_t)PyArray_DIM(a, 0);
  }
  return PyObject_Length(row);
}

/* filter_coverage(norm, base, min_cov) -> number of positions deleted.  mfilter_coverage's inner loop
 * (myDetect.py:304-309) for one (chrom, strand): positions with fewer than min_cov samples leave both dicts. */
static PyObject* filter_coverage(PyObject* self, PyObject* args) {
  PyObject *norm, *base; long long min_cov;
  (void)self;
  if (!PyArg_ParseTuple(args, "O!O!L", &PyDict_Type, &norm, &PyDict_Type, &base, &min_cov)) return NULL;
  Py_ssize_t it = 0, ndel = 0, cap = 0;
  PyObject *k, *v, **dead = NULL;
  while (PyDict_Next(norm, &it, &k, &v)) {
    const Py_ssize_t n = row_len(v);
    if (n < 0) { free(dead); return NULL; }
    if (n < min_cov) {
      if (ndel == cap) { cap = cap ? 2 * cap : 1024; PyObject** nd = (PyObject**)realloc(dead, (size_t)cap * sizeof(*dead)); if (!nd) { free(dead); return PyErr_NoMemory(); } dead = nd; }
      Py_INCREF(k); dead[ndel++] = k;
    }
  }
  int failed = 0;
  for (Py_ssize_t i = 0; i < ndel; ++i) {
    if (!failed && (PyDict_DelItem(norm, dead[i]) < 0 || PyDict_DelItem(base, dead[i]) < 0)) failed = 1;    /* KeyError on base: as the reference's del */
    Py_DECREF(dead[i]);
  }
  free(dead);
  if (failed) return NULL;
  return PyLong_FromSsize_t(ndel);
}

typedef struct { long long pos; PyObject* row; } ent_t;
static int ent_cmp(const void* a, const void* b) {
  const long long x = ((const ent_t*)a)->pos, y = ((const ent_t*)b)->pos;
  return x < y ? -1 : (x > y ? 1 : 0);
}
/* the (position, value) pairs of a dict, ascending by position; NULL + exception on a key that is not an integer */
static ent_t* dict_entries(PyObject* d, Py_ssize_t* n_out) {
  const Py_ssize_t n = PyDict_Size(d);
  ent_t* e = (ent_t*)malloc((size_t)(n ? n : 1) * sizeof(ent_t));
  if (!e) { PyErr_NoMemory(); return NULL; }
  Py_ssize_t it = 0, i = 0;
  PyObject *k, *v;
  int sorted = 1;
  while (PyDict_Next(d, &it, &k, &v)) {
    long long p;
    if (PyLong_CheckExact(k)) p = PyLong_AsLongLong(k);
    else { PyObject* ix = PyNumber_Index(k); if (!ix) { free(e); return NULL; } p = PyLong_AsLongLong(ix); Py_DECREF(ix); }
    if (p == -1 && PyErr_Occurred()) { free(e); return NULL; }
    if (i > 0 && p <= e[i - 1].pos) sorted = 0;
    e[i].pos = p; e[i].row = v; ++i;
  }
  if (!sorted) qsort(e, (size_t)i, sizeof(ent_t), ent_cmp);
  *n_out = i;
  return e;
}

static int copy_row(PyObject* row, double* dst, Py_ssize_t n) {
  if (PyArray_Check(row) && PyArray_TYPE((PyArrayObject*)row) == NPY_DOUBLE && PyArray_IS_C_CONTIGUOUS((PyArrayObject*)row)) {
    memcpy(dst, PyArray_DATA((PyArrayObject*)row), (size_t)n * 8);
    return 0;
  }
  Py_ssize_t wrote = 0;
  if (put_generic(row, dst, n, &wrote) < 0) return -1;
  if (wrote != n) { PyErr_SetString(PyExc_ValueError, "a position changed its number of samples while it was read"); return -1; }
  return 0;
}

/* The copy phase of join_strand on several threads.  The calling thread HOLDS the GIL while it waits, so no other Python code runs
 * and no row can change; the helper threads touch no Python API that needs the GIL — only macros that read object memory (type
 * flags, list item pointers, the double inside a float object, an ndarray's data pointer).  A row they cannot take that way (a
 * tuple, an ndarray of another dtype, a list holding something that is not a float) is left for the caller (`slow`). */
#include <pthread.h>
typedef struct {
  const ent_t* e; const Py_ssize_t* ix; const int* nn; const long long* start; double* dst;
  Py_ssize_t lo, hi; unsigned char* slow;
  short* dst16; volatile int* refused;       /* dst16 != NULL: the rows are narrowed to int16 milli-units; *refused = 1 when a sample is not one */
} copy_job_t;

static void* copy_worker(void* arg) {
  copy_job_t* jb = (copy_job_t*)arg;
  for (Py_ssize_t j = jb->lo; j < jb->hi; ++j) {
    if (j + 16 < jb->hi) __builtin_prefetch(jb->e[jb->ix[j + 16]].row);
    if (j + 8 < jb->hi) {
      PyObject* o = jb->e[jb->ix[j + 8]].row;
      if (PyList_CheckExact(o)) __builtin_prefetch(((PyListObject*)o)->ob_item);
      else if (PyArray_Check(o)) __builtin_prefetch(PyArray_DATA((PyArrayObject*)o));
    }
    PyObject* row = jb->e[jb->ix[j]].row;
    const int n = jb->nn[j];
    if (jb->dst16) {                                       /* narrowing copy */
      if (*jb->refused) return NULL;
      short* d16 = jb->dst16 + jb->start[j];
      if (PyArray_Check(row) && PyArray_TYPE((PyArrayObject*)row) == NPY_DOUBLE && PyArray_IS_C_CONTIGUOUS((PyArrayObject*)row) &&
          PyArray_NDIM((PyArrayObject*)row) == 1 && PyArray_DIM((PyArrayObject*)row, 0) == n) {
        if (!narrow_row((const double*)PyArray_DATA((PyArrayObject*)row), d16, n)) { *jb->refused = 1; return NULL; }
      } else if (PyList_CheckExact(row) && PyList_GET_SIZE(row) == n) {
        PyObject** items = ((PyListObject*)row)->ob_item;
        double buf[256];
        for (int i0 = 0; i0 < n && !jb->slow[j]; i0 += 256) {       /* 256 values at a time through the vector code */
          const int m = n - i0 < 256 ? n - i0 : 256;
          for (int i = 0; i < m; ++i) {
            PyObject* it = items[i0 + i];
            if (!PyFloat_Check(it)) { jb->slow[j] = 1; break; }
            buf[i] = PyFloat_AS_DOUBLE(it);
          }
          if (!jb->slow[j] && !narrow_row(buf, d16 + i0, m)) { *jb->refused = 1; return NULL; }
        }
      } else {
        jb->slow[j] = 1;
      }
      continue;
    }
    double* d = jb->dst + jb->start[j];
    if (PyArray_Check(row) && PyArray_TYPE((PyArrayObject*)row) == NPY_DOUBLE && PyArray_IS_C_CONTIGUOUS((PyArrayObject*)row) &&
        PyArray_NDIM((PyArrayObject*)row) == 1 && PyArray_DIM((PyArrayObject*)row, 0) == n) {
      memcpy(d, PyArray_DATA((PyArrayObject*)row), (size_t)n * 8);
    } else if (PyList_CheckExact(row) && PyList_GET_SIZE(row) == n) {
      PyObject** items = ((PyListObject*)row)->ob_item;
      int ok = 1;
      for (int i = 0; i < n; ++i) {
        PyObject* it = items[i];
        if (!PyFloat_Check(it)) { ok = 0; break; }         /* numpy.float64 is a float subclass */
        d[i] = PyFloat_AS_DOUBLE(it);
      }
      if (!ok) jb->slow[j] = 1;
    } else {
      jb->slow[j] = 1;
    }
  }
  return NULL;
}

static int host_threads(void) {
  const char* s = getenv("NMOD_HOSTWALK_THREADS");
  int t = s ? atoi(s) : 8;      /* (capped by the cgroup CPU quota below; 4 until round 5: 0.25 s of a 0.39 s mtest2 at 460 000 x 200 v 200 was this copy) */
  long q = -1, per = -1;
  FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) { char buf[64]; if (fscanf(f, "%63s %ld", buf, &per) == 2 && strcmp(buf, "max") != 0) q = atol(buf); fclose(f); }
  if (q > 0 && per > 0 && q / per < t) t = (int)(q / per);
  return t < 1 ? 1 : (t > 16 ? 16 : t);
}

/* rows of one group -> dst (CSR order; dst16 != NULL: narrowed to int16 milli-units instead); 0, 1 = a sample refused the narrowing
 * (dst16 is then unspecified), or -1 with an exception set */
static int copy_rows(const ent_t* e, const Py_ssize_t* ix, const int* nn, Py_ssize_t n, double* dst, short* dst16) {
  volatile int refused = 0;
  long long* start = (long long*)malloc((size_t)(n ? n : 1) * sizeof(long long));
  unsigned char* slow = (unsigned char*)calloc((size_t)(n ? n : 1), 1);
  if (!start || !slow) { free(start); free(slow); PyErr_NoMemory(); return -1; }
  long long at = 0;
  for (Py_ssize_t j = 0; j < n; ++j) { start[j] = at; at += nn[j]; }
  int T = host_threads();
  if (n < 4096) T = 1;
  copy_job_t jobs[16]; pthread_t th[16]; int started[16];
  long long per = at / T + 1, acc = 0; Py_ssize_t lo = 0; int used = 0;
  for (Py_ssize_t j = 0; j < n && used < T - 1; ++j) {           /* ranges of about equal sample counts */
    acc += nn[j];
    if (acc >= per) { jobs[used] = (copy_job_t){e, ix, nn, start, dst, lo, j + 1, slow, dst16, &refused}; lo = j + 1; acc = 0; ++used; }
  }
  jobs[used] = (copy_job_t){e, ix, nn, start, dst, lo, n, slow, dst16, &refused}; ++used;
  for (int t = 1; t < used; ++t) started[t] = pthread_create(&th[t], NULL, copy_worker, &jobs[t]) == 0;
  copy_worker(&jobs[0]);
  for (int t = 1; t < used; ++t) { if (started[t]) pthread_join(th[t], NULL); else copy_worker(&jobs[t]); }
  int rc = 0;
  double* tmp = NULL;
  for (Py_ssize_t j = 0; j < n && rc == 0 && !refused; ++j) {
    if (!slow[j]) continue;                                /* under the GIL: sequence protocol, float() */
    if (!dst16) { rc = copy_row(e[ix[j]].row, dst + start[j], nn[j]); continue; }
    double* t2 = (double*)realloc(tmp, (size_t)(nn[j] ? nn[j] : 1) * 8);
    if (!t2) { PyErr_NoMemory(); rc = -1; break; }
    tmp = t2;
    rc = copy_row(e[ix[j]].row, tmp, nn[j]);
    if (rc == 0 && !narrow_scalar(tmp, dst16 + start[j], nn[j])) refused = 1;
  }
  free(tmp); free(start); free(slow);
  return rc < 0 ? -1 : (refused ? 1 : 0);
}

/* join_strand(norm0, norm1, base0, base1) -> (pos int64[n], n0 int32[n], n1 int32[n], sig0 float64[sum n0], sig1 float64[sum n1],
 *                                             bases list[n] (group 2's, as mtest2 records them), mismatch list of indices,
 *                                             base_codes uint32[n]: the code point of a one-character base, else 0)
 * The loop header of mtest2 for one (chrom, strand) (myDetect.py:427-436): the positions both datasets hold, ascending, their
 * rows flattened into two CSR sample arrays, and the indices where the two datasets disagree about the base (:432-434). */
/* what join_strand_plan leaves for copy_plan: the joined entries of one (chrom, strand) — rows BORROWED from the two dicts, which the
 * plan keeps alive (the caller must not change them between the two calls: build_csr does not) */
typedef struct { ent_t *e0, *e1; Py_ssize_t *i0, *i1; PyObject *n0_a, *n1_a, *d0, *d1; Py_ssize_t n; } plan_t;
static void plan_free(PyObject* cap) {
  plan_t* p = (plan_t*)PyCapsule_GetPointer(cap, "nanomod_amd.join_plan");
  if (!p) return;
  free(p->e0); free(p->e1); free(p->i0); free(p->i1);
  Py_XDECREF(p->n0_a); Py_XDECREF(p->n1_a); Py_XDECREF(p->d0); Py_XDECREF(p->d1);
  free(p);
}

static PyObject* join_impl(PyObject* args, int defer) {
  PyObject *d0, *d1, *b0, *b1;
  if (!PyArg_ParseTuple(args, "O!O!O!O!", &PyDict_Type, &d0, &PyDict_Type, &d1, &PyDict_Type, &b0, &PyDict_Type, &b1)) return NULL;
  Py_ssize_t m0 = 0, m1 = 0, mb0 = 0, mb1 = 0;
  ent_t *e0 = NULL, *e1 = NULL, *eb0 = NULL, *eb1 = NULL;
  PyObject *pos_a = NULL, *n0_a = NULL, *n1_a = NULL, *s0_a = NULL, *s1_a = NULL, *bases = NULL, *mism = NULL, *ret = NULL, *codes_a = NULL;
  Py_ssize_t *i0 = NULL, *i1 = NULL;
  if (!(e0 = dict_entries(d0, &m0)) || !(e1 = dict_entries(d1, &m1)) || !(eb0 = dict_entries(b0, &mb0)) || !(eb1 = dict_entries(b1, &mb1))) goto done;
  /* merge join of the two ascending position lists */
  const Py_ssize_t cap = m0 < m1 ? m0 : m1;
  i0 = (Py_ssize_t*)malloc((size_t)(cap ? cap : 1) * sizeof(Py_ssize_t)); i1 = (Py_ssize_t*)malloc((size_t)(cap ? cap : 1) * sizeof(Py_ssize_t));
  if (!i0 || !i1) { PyErr_NoMemory(); goto done; }
  Py_ssize_t n = 0;
  for (Py_ssize_t a = 0, b = 0; a < m0 && b < m1;) {
    if (e0[a].pos < e1[b].pos) ++a;
    else if (e0[a].pos > e1[b].pos) ++b;
    else { i0[n] = a; i1[n] = b; ++n; ++a; ++b; }
  }
  npy_intp dn = (npy_intp)n;
  pos_a = PyArray_SimpleNew(1, &dn, NPY_INT64); n0_a = PyArray_SimpleNew(1, &dn, NPY_INT32); n1_a = PyArray_SimpleNew(1, &dn, NPY_INT32);
  bases = PyList_New(n); mism = PyList_New(0); codes_a = PyArray_SimpleNew(1, &dn, NPY_UINT32);
  if (!pos_a || !n0_a || !n1_a || !bases || !mism || !codes_a) goto done;
  npy_uint32* codes = (npy_uint32*)PyArray_DATA((PyArrayObject*)codes_a);
  long long* pos = (long long*)PyArray_DATA((PyArrayObject*)pos_a);
  int* n0 = (int*)PyArray_DATA((PyArrayObject*)n0_a); int* n1 = (int*)PyArray_DATA((PyArrayObject*)n1_a);
  long long t0 = 0, t1 = 0;
  Py_ssize_t jb0 = 0, jb1 = 0;                           /* cursors into the base dicts (the same ascending positions in any real input) */
  for (Py_ssize_t j = 0; j < n; ++j) {
    if (j + 8 < n) { __builtin_prefetch(e0[i0[j + 8]].row); __builtin_prefetch(e1[i1[j + 8]].row); }
    const long long p = e0[i0[j]].pos;
    /* the rows and bases are BORROWED from the dicts: nothing here may run Python code that could change them.  Exact lists /
     * tuples / arrays have a C-level length and exact str a C-level ==; anything else (a subclass with its own __len__ or
     * __eq__) sends the strand to the Python loop (detect._join_strand_py) through TypeError */
    {
      PyObject *r0 = e0[i0[j]].row, *r1 = e1[i1[j]].row;
      if (!((PyList_CheckExact(r0) || PyTuple_CheckExact(r0) || PyArray_CheckExact(r0)) &&
            (PyList_CheckExact(r1) || PyTuple_CheckExact(r1) || PyArray_CheckExact(r1)))) {
        PyErr_SetString(PyExc_TypeError, "join_strand: a row of a type the C walk does not take");
        goto done;
      }
    }
    const Py_ssize_t l0 = row_len(e0[i0[j]].row), l1 = row_len(e1[i1[j]].row);
    if (l0 < 0 || l1 < 0) goto done;
    if (l0 > 2147483647 || l1 > 2147483647) { PyErr_SetString(PyExc_OverflowError, "a position holds more than 2^31 samples"); goto done; }
    pos[j] = p; n0[j] = (int)l0; n1[j] = (int)l1; t0 += l0; t1 += l1;
    while (jb0 < mb0 && eb0[jb0].pos < p) ++jb0;
    while (jb1 < mb1 && eb1[jb1].pos < p) ++jb1;
    if (jb0 >= mb0 || eb0[jb0].pos != p || jb1 >= mb1 || eb1[jb1].pos != p) { PyObject* kp = PyLong_FromLongLong(p); if (kp) { PyErr_SetObject(PyExc_KeyError, kp); Py_DECREF(kp); } goto done; }   /* as base[sk][pk] would */
    PyObject *x0 = eb0[jb0].row, *x1 = eb1[jb1].row;
    if (!PyUnicode_CheckExact(x0) || !PyUnicode_CheckExact(x1)) {
      PyErr_SetString(PyExc_TypeError, "join_strand: a base of a type the C walk does not take");
      goto done;
    }
    Py_INCREF(x1); PyList_SET_ITEM(bases, j, x1);
    codes[j] = (PyUnicode_CheckExact(x1) && PyUnicode_GET_LENGTH(x1) == 1) ? (npy_uint32)PyUnicode_READ_CHAR(x1, 0) : 0u;
    if (x0 != x1) {
      const int eq = PyObject_RichCompareBool(x1, x0, Py_EQ);
      if (eq < 0) goto done;
      if (!eq) { PyObject* ix = PyLong_FromSsize_t(j); if (!ix || PyList_Append(mism, ix) < 0) { Py_XDECREF(ix); goto done; } Py_DECREF(ix); }
    }
  }
  if (defer) {                                           /* the rows are copied later, by copy_plan, into arrays the caller makes */
    plan_t* pl = (plan_t*)calloc(1, sizeof(plan_t));
    if (!pl) { PyErr_NoMemory(); goto done; }
    PyObject* cap = PyCapsule_New(pl, "nanomod_amd.join_plan", plan_free);
    if (!cap) { free(pl); goto done; }
    pl->e0 = e0; pl->e1 = e1; pl->i0 = i0; pl->i1 = i1; pl->n = n;
    pl->n0_a = n0_a; Py_INCREF(n0_a); pl->n1_a = n1_a; Py_INCREF(n1_a); pl->d0 = d0; Py_INCREF(d0); pl->d1 = d1; Py_INCREF(d1);
    e0 = e1 = NULL; i0 = i1 = NULL;                        /* (owned by the plan now) */
    ret = PyTuple_Pack(7, pos_a, n0_a, n1_a, cap, bases, mism, codes_a);
    Py_DECREF(cap);
    goto done;
  }
  npy_intp d0n = (npy_intp)t0, d1n = (npy_intp)t1;
  s0_a = PyArray_SimpleNew(1, &d0n, NPY_DOUBLE); s1_a = PyArray_SimpleNew(1, &d1n, NPY_DOUBLE);
  if (!s0_a || !s1_a) goto done;
  double* s0 = (double*)PyArray_DATA((PyArrayObject*)s0_a); double* s1 = (double*)PyArray_DATA((PyArrayObject*)s1_a);
  if (copy_rows(e0, i0, n0, n, s0, NULL) < 0 || copy_rows(e1, i1, n1, n, s1, NULL) < 0) goto done;   /* one group after the other */
  ret = PyTuple_Pack(8, pos_a, n0_a, n1_a, s0_a, s1_a, bases, mism, codes_a);
done:
  free(e0); free(e1); free(eb0); free(eb1); free(i0); free(i1);
  Py_XDECREF(pos_a); Py_XDECREF(n0_a); Py_XDECREF(n1_a); Py_XDECREF(s0_a); Py_XDECREF(s1_a); Py_XDECREF(bases); Py_XDECREF(mism); Py_XDECREF(codes_a);
  return ret;
}

static PyObject* join_strand(PyObject* self, PyObject* args) { (void)self; return join_impl(args, 0); }
static PyObject* join_strand_plan(PyObject* self, PyObject* args) { (void)self; return join_impl(args, 1); }

/* copy_plan(plan, out0, start0, out1, start1) -> True, or False when out is int16 and a sample is not k / 1000.0 with |k| <= 32 767
 * (the int16 arrays then hold nothing usable).  out0 / out1: writable C-contiguous 1-D arrays, both float64 or both int16, with room
 * for the strand's samples from element start0 / start1 on. */
static PyObject* copy_plan(PyObject* self, PyObject* args) {
  PyObject *cap, *o0, *o1; long long st0, st1;
  (void)self;
  if (!PyArg_ParseTuple(args, "OO!LO!L", &cap, &PyArray_Type, &o0, &st0, &PyArray_Type, &o1, &st1)) return NULL;
  plan_t* pl = (plan_t*)PyCapsule_GetPointer(cap, "nanomod_amd.join_plan");
  if (!pl) return NULL;
  PyArrayObject *a0 = (PyArrayObject*)o0, *a1 = (PyArrayObject*)o1;
  const int ty = PyArray_TYPE(a0);
  if ((ty != NPY_DOUBLE && ty != NPY_INT16) || PyArray_TYPE(a1) != ty || PyArray_NDIM(a0) != 1 || PyArray_NDIM(a1) != 1 ||
      !PyArray_IS_C_CONTIGUOUS(a0) || !PyArray_IS_C_CONTIGUOUS(a1) || !PyArray_ISWRITEABLE(a0) || !PyArray_ISWRITEABLE(a1)) {
    PyErr_SetString(PyExc_ValueError, "copy_plan: out0 / out1 must be writable contiguous 1-D arrays, both float64 or both int16"); return NULL;
  }
  const int* n0 = (const int*)PyArray_DATA((PyArrayObject*)pl->n0_a); const int* n1 = (const int*)PyArray_DATA((PyArrayObject*)pl->n1_a);
  long long t0 = 0, t1 = 0;
  for (Py_ssize_t j = 0; j < pl->n; ++j) { t0 += n0[j]; t1 += n1[j]; }
  if (st0 < 0 || st1 < 0 || st0 + t0 > (long long)PyArray_DIM(a0, 0) || st1 + t1 > (long long)PyArray_DIM(a1, 0)) {
    PyErr_SetString(PyExc_ValueError, "copy_plan: the strand's samples do not fit the output arrays"); return NULL;
  }
  int r0, r1 = 0;
  if (ty == NPY_DOUBLE) {
    r0 = copy_rows(pl->e0, pl->i0, n0, pl->n, (double*)PyArray_DATA(a0) + st0, NULL);
    if (r0 == 0) r1 = copy_rows(pl->e1, pl->i1, n1, pl->n, (double*)PyArray_DATA(a1) + st1, NULL);
  } else {
    r0 = copy_rows(pl->e0, pl->i0, n0, pl->n, NULL, (short*)PyArray_DATA(a0) + st0);
    if (r0 == 0) r1 = copy_rows(pl->e1, pl->i1, n1, pl->n, NULL, (short*)PyArray_DATA(a1) + st1);
  }
  if (r0 < 0 || r1 < 0) return NULL;
  if (r0 == 1 || r1 == 1) Py_RETURN_FALSE;
  Py_RETURN_TRUE;
}

static PyMethodDef methods[] = {
  {"flatten", flatten, METH_VARARGS, "flatten(rows, out): copy the values of every row into the float64 buffer `out`, in order"},
  {"filter_coverage", filter_coverage, METH_VARARGS, "filter_coverage(norm, base, min_cov): delete the positions with fewer than min_cov samples from both dicts"},
  {"join_strand", join_strand, METH_VARARGS, "join_strand(norm0, norm1, base0, base1): positions of both datasets, ascending, as CSR arrays"},
  {"join_strand_plan", join_strand_plan, METH_VARARGS, "join_strand_plan(norm0, norm1, base0, base1): the same join without the rows: (pos, n0, n1, plan, bases, mismatches, base codes)"},
  {"copy_plan", copy_plan, METH_VARARGS, "copy_plan(plan, out0, start0, out1, start1): the rows of a planned strand into float64 or (narrowed) int16 arrays; False: a sample refused int16"},
  {NULL, NULL, 0, NULL}
};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostwalk", "host-side row flattening for detect.build_csr", -1, methods,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__hostwalk(void) { import_array(); return PyModule_Create(&module); }
