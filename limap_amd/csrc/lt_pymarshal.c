/* lt_pymarshal.c -- CPython helper for the Python mirror (limap_amd/triangulation.py): turns the
 * `matches` argument of TriangulateImage -- dict[int -> ndarray (K,2) int32] -- into the pointer / row-count
 * arrays of lt_triangulate_image_rows and calls it.  Getting the data pointer of 20 numpy arrays from Python
 * costs ~22 us per image (1.1 us per `.ctypes.data`), as much as the native buffering of the image's 10^5
 * rows; through the buffer protocol it is ~1 us.  Host-side marshalling only: no arithmetic, and the Python
 * path remains (arrays that are not C-contiguous int32 (K,2) make this return None and the caller converts
 * them the slow way, like pybind11's Eigen::MatrixXi caster does by copy).
 * Built by `make` next to liblimap_amd.so as limap_amd/_lt_pymarshal.so; optional at run time. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

typedef int (*rows_fn)(void *ctx, int img_id, int n_nb, const int32_t *nb_ids, const int32_t *const *rows,
                       const int64_t *n_rows);

#define MAX_NB 256

/* triangulate_image_rows(fn_addr: int, ctx_addr: int, img_id: int, matches: dict) -> rc (int) | None */
static PyObject *py_triangulate_image_rows(PyObject *self, PyObject *args) {
  unsigned long long fn_addr, ctx_addr;
  int img_id;
  PyObject *matches;
  if (!PyArg_ParseTuple(args, "KKiO!", &fn_addr, &ctx_addr, &img_id, &PyDict_Type, &matches)) return NULL;
  const Py_ssize_t n = PyDict_Size(matches);
  if (n > MAX_NB) Py_RETURN_NONE;
  int32_t nb[MAX_NB];
  const int32_t *rows[MAX_NB];
  int64_t cnt[MAX_NB];
  Py_buffer views[MAX_NB];
  Py_ssize_t pos = 0, k = 0;
  PyObject *key, *val;
  int ok = 1;
  while (ok && PyDict_Next(matches, &pos, &key, &val)) {
    long id = PyLong_AsLong(key);
    if (id == -1 && PyErr_Occurred()) {
      PyErr_Clear();
      ok = 0;
      break;
    }
    if (PyObject_GetBuffer(val, &views[k], PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) {
      PyErr_Clear();
      ok = 0;
      break;
    }
    const Py_buffer *v = &views[k];
    const char *f = v->format ? v->format : "";
    if (*f == '@' || *f == '=' || *f == '<') ++f;
    if (v->ndim != 2 || v->shape[1] != 2 || v->itemsize != 4 || !((f[0] == 'i' || f[0] == 'l') && f[1] == 0)) {
      PyBuffer_Release(&views[k]);
      ok = 0;
      break;
    }
    nb[k] = (int32_t)id;
    rows[k] = (const int32_t *)v->buf;
    cnt[k] = (int64_t)v->shape[0];
    ++k;
  }
  PyObject *ret = NULL;
  if (ok) {
    int rc = ((rows_fn)(uintptr_t)fn_addr)((void *)(uintptr_t)ctx_addr, img_id, (int)k, nb, rows, cnt);
    ret = PyLong_FromLong(rc);
  }
  for (Py_ssize_t i = 0; i < k; ++i) PyBuffer_Release(&views[i]);
  if (!ok) Py_RETURN_NONE;
  return ret;
}

static PyMethodDef methods[] = {
    {"triangulate_image_rows", py_triangulate_image_rows, METH_VARARGS,
     "triangulate_image_rows(fn_addr, ctx_addr, img_id, matches) -> rc, or None if an array needs converting"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_lt_pymarshal", NULL, -1, methods};
PyMODINIT_FUNC PyInit__lt_pymarshal(void) { return PyModule_Create(&moddef); }
