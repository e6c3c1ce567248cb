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

static PyMethodDef methods[] = {
  {"flatten", flatten, METH_VARARGS, "flatten(rows, out): copy the values of every row into the float64 buffer `out`, in order"},
  {NULL, NULL, 0, NULL}
};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_hostwalk", "host-side row flattening for detect.build_csr", -1, methods,
                                    NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__hostwalk(void) { import_array(); return PyModule_Create(&module); }
