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
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <string.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>

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
    double* d = jb->dst + jb->start[j];
    const int n = jb->nn[j];
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

/* rows of one group -> dst (CSR order); 0 or -1 with an exception set */
static int copy_rows(const ent_t* e, const Py_ssize_t* ix, const int* nn, Py_ssize_t n, double* dst) {
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
    if (acc >= per) { jobs[used] = (copy_job_t){e, ix, nn, start, dst, lo, j + 1, slow}; lo = j + 1; acc = 0; ++used; }
  }
  jobs[used] = (copy_job_t){e, ix, nn, start, dst, lo, n, slow}; ++used;
  for (int t = 1; t < used; ++t) started[t] = pthread_create(&th[t], NULL, copy_worker, &jobs[t]) == 0;
  copy_worker(&jobs[0]);
  for (int t = 1; t < used; ++t) { if (started[t]) pthread_join(th[t], NULL); else copy_worker(&jobs[t]); }
  int rc = 0;
  for (Py_ssize_t j = 0; j < n && rc == 0; ++j)
    if (slow[j]) rc = copy_row(e[ix[j]].row, dst + start[j], nn[j]);      /* under the GIL: sequence protocol, float() */
  free(start); free(slow);
  return rc;
}

/* join_strand(norm0, norm1, base0, base1) -> (pos int64[n], n0 int32[n], n1 int32[n], sig0 float64[sum n0], sig1 float64[sum n1],
 *                                             bases list[n] (group 2's, as mtest2 records them), mismatch list of indices,
 *                                             base_codes uint32[n]: the code point of a one-character base, else 0)
 * The loop header of mtest2 for one (chrom, strand) (myDetect.py:427-436): the positions both datasets hold, ascending, their
 * rows flattened into two CSR sample arrays, and the indices where the two datasets disagree about the base (:432-434). */
static PyObject* join_strand(PyObject* self, PyObject* args) {
  PyObject *d0, *d1, *b0, *b1;
  (void)self;
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
  npy_intp d0n = (npy_intp)t0, d1n = (npy_intp)t1;
  s0_a = PyArray_SimpleNew(1, &d0n, NPY_DOUBLE); s1_a = PyArray_SimpleNew(1, &d1n, NPY_DOUBLE);
  if (!s0_a || !s1_a) goto done;
  double* s0 = (double*)PyArray_DATA((PyArrayObject*)s0_a); double* s1 = (double*)PyArray_DATA((PyArrayObject*)s1_a);
  if (copy_rows(e0, i0, n0, n, s0) < 0 || copy_rows(e1, i1, n1, n, s1) < 0) goto done;   /* one group after the other */
  ret = PyTuple_Pack(8, pos_a, n0_a, n1_a, s0_a, s1_a, bases, mism, codes_a);
done:
  free(e0); free(e1); free(eb0); free(eb1); free(i0); free(i1);
  Py_XDECREF(pos_a); Py_XDECREF(n0_a); Py_XDECREF(n1_a); Py_XDECREF(s0_a); Py_XDECREF(s1_a); Py_XDECREF(bases); Py_XDECREF(mism); Py_XDECREF(codes_a);
  return ret;
}

static PyMethodDef methods[] = {
  {"flatten", flatten, METH_VARARGS, "flatten(rows, out): copy the values of every row into the float64 buffer `out`, in order"},
  {"filter_coverage", filter_coverage, METH_VARARGS, "filter_coverage(norm, base, min_cov): delete the positions with fewer than min_cov samples from both dicts"},
  {"join_strand", join_strand, METH_VARARGS, "join_strand(norm0, norm1, base0, base1): positions of both datasets, ascending, as CSR arrays"},
  {NULL, NULL, 0, NULL}
};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostwalk", "host-side row flattening for detect.build_csr", -1, methods,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__hostwalk(void) { import_array(); return PyModule_Create(&module); }
