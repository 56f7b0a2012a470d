/* sbayes_amd._sbe_pyhost -- CPython extension of the drop-in host layer (plain C, no HIP, no NumPy headers: everything goes
 * through the buffer protocol).  What it is for: the host layer above the C ABI is per-MCMC-step Python, and tools/host_residual.py
 * shows it costs more than the device side of a step; a measurable part of it is glue that has no Python-level fix --
 *   addr(a)                    the buffer address of an array: ndarray.__array_interface__ builds a dict (0.9 us), .ctypes an
 *                              object (0.95 us); the engine wrappers need it for every array argument of every call
 *   subset_ids(...)            the ids sbe_counts_delta takes, from the samples' own arrays (sbe_host_helpers.h: sbeh_subset_ids)
 *   diff_rows(new, mirror, idx)   the bind cache's content compare (sbeh_diff_rows)
 *   touched_groups(gid_old, gid_new, n_groups_total, out) -> n
 * The same helpers are exported by the engine library (sbe_host_*); sbayes_amd/_fast.py falls back to those through ctypes when
 * this module is not built.  Nothing here touches the device. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include "sbe_host_helpers.h"

static PyObject* py_addr(PyObject* self, PyObject* obj) {
    Py_buffer v;
    if (PyObject_GetBuffer(obj, &v, PyBUF_STRIDED_RO) != 0) return NULL;
    PyObject* r = PyLong_FromVoidPtr(v.buf);
    PyBuffer_Release(&v);
    return r;
}

/* a C-contiguous buffer of 1-byte items (bool / uint8) or of `itemsize`-byte items with `ndim` dimensions; 0 on mismatch */
static int get_c(PyObject* obj, Py_buffer* v, int ndim, Py_ssize_t itemsize, int writable) {
    if (PyObject_GetBuffer(obj, v, (writable ? PyBUF_WRITABLE : 0) | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); return 0; }
    if (v->ndim != ndim || v->itemsize != itemsize) { PyBuffer_Release(v); return 0; }
    return 1;
}

/* subset_ids(objs, groups_new, groups_old, src_new, src_old, gid_new, gid_old, sid_new, sid_old) -> 0 / 1 / -1 like
 * sbeh_subset_ids, or -2 when an argument is not in ABI form (the caller converts and takes the ctypes route) */
static PyObject* py_subset_ids(PyObject* self, PyObject* args) {
    PyObject *objs, *gnew, *gold, *snew, *sold, *o_gn, *o_go, *o_sn, *o_so;
    if (!PyArg_ParseTuple(args, "OOOOOOOOO", &objs, &gnew, &gold, &snew, &sold, &o_gn, &o_go, &o_sn, &o_so)) return NULL;
    if (!PyList_Check(gnew) || !PyList_Check(gold) || PyList_GET_SIZE(gnew) != PyList_GET_SIZE(gold)) {
        PyErr_SetString(PyExc_TypeError, "groups_new / groups_old must be lists of equal length");
        return NULL;
    }
    const Py_ssize_t C = PyList_GET_SIZE(gnew);
    if (C < 1 || C > 254) { PyErr_SetString(PyExc_ValueError, "1..254 components"); return NULL; }
    Py_buffer vb[4 + 2 * 254 + 4];
    int nb = 0;
    long rc = -2;
    const uint8_t* pn[254]; const uint8_t* po[254]; int32_t ng[254];
    Py_buffer *b_objs, *b_sn, *b_so;
    if (!get_c(objs, &vb[nb], 1, 4, 0)) goto done;
    b_objs = &vb[nb++];
    if (!get_c(snew, &vb[nb], 3, 1, 0)) goto done;
    b_sn = &vb[nb++];
    if (sold == snew) b_so = b_sn;
    else { if (!get_c(sold, &vb[nb], 3, 1, 0)) goto done; b_so = &vb[nb++]; }
    const Py_ssize_t n = b_objs->shape[0], N = b_sn->shape[0], F = b_sn->shape[1], Cs = b_sn->shape[2];
    if (Cs != C || b_so->shape[0] != N || b_so->shape[1] != F || b_so->shape[2] != C) goto done;
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject* a = PyList_GET_ITEM(gnew, c);
        PyObject* b = PyList_GET_ITEM(gold, c);
        if (!get_c(a, &vb[nb], 2, 1, 0)) goto done;
        Py_buffer* va = &vb[nb++];
        if (va->shape[1] != N) goto done;
        pn[c] = (const uint8_t*)va->buf; ng[c] = (int32_t)va->shape[0];
        if (b == a) po[c] = pn[c];
        else {
            if (!get_c(b, &vb[nb], 2, 1, 0)) goto done;
            Py_buffer* vo = &vb[nb++];
            if (vo->shape[0] != va->shape[0] || vo->shape[1] != N) goto done;
            po[c] = (const uint8_t*)vo->buf;
        }
    }
    {
        Py_buffer *g1, *g2, *s1, *s2;
        if (!get_c(o_gn, &vb[nb], 2, 4, 1)) goto done;
        g1 = &vb[nb++];
        if (!get_c(o_go, &vb[nb], 2, 4, 1)) goto done;
        g2 = &vb[nb++];
        if (!get_c(o_sn, &vb[nb], 2, 1, 1)) goto done;
        s1 = &vb[nb++];
        if (o_so == o_sn) s2 = s1;
        else { if (!get_c(o_so, &vb[nb], 2, 1, 1)) goto done; s2 = &vb[nb++]; }
        if (g1->shape[0] != C || g1->shape[1] != n || g2->shape[0] != C || g2->shape[1] != n || s1->shape[0] != n || s1->shape[1] != F ||
            s2->shape[0] != n || s2->shape[1] != F) goto done;
        rc = sbeh_subset_ids((const int32_t*)b_objs->buf, (int)n, (int64_t)N, (int)F, (int)C, ng, pn, po, (const uint8_t*)b_sn->buf,
                             (const uint8_t*)b_so->buf, (int32_t*)g1->buf, (int32_t*)g2->buf, (uint8_t*)s1->buf, (uint8_t*)s2->buf);
    }
done:
    for (int i = 0; i < nb; ++i) PyBuffer_Release(&vb[i]);
    return PyLong_FromLong(rc);
}

/* diff_rows(new, mirror, idx_out) -> number of differing rows (copied into mirror, indices in idx_out), -2: not in ABI form */
static PyObject* py_diff_rows(PyObject* self, PyObject* args) {
    PyObject *a, *m, *idx;
    if (!PyArg_ParseTuple(args, "OOO", &a, &m, &idx)) return NULL;
    Py_buffer va, vm, vi;
    long long rc = -2;
    if (PyObject_GetBuffer(a, &va, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(m, &vm, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&va); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(idx, &vi, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&va); PyBuffer_Release(&vm); return PyLong_FromLong(-2); }
    if (va.ndim >= 1 && va.len == vm.len && va.shape[0] > 0 && vi.itemsize == 4 && vi.len >= 4 * va.shape[0])
        rc = sbeh_diff_rows(va.buf, vm.buf, (int64_t)va.shape[0], (int64_t)(va.len / va.shape[0]), (int32_t*)vi.buf);
    else if (va.ndim >= 1 && va.len == vm.len && va.shape[0] == 0) rc = 0;
    PyBuffer_Release(&va); PyBuffer_Release(&vm); PyBuffer_Release(&vi);
    return PyLong_FromLongLong(rc);
}

/* touched_groups(gid_old, gid_new, n_groups_total, touched_out) -> n_touched, -1 bad index, -2 not in ABI form */
static PyObject* py_touched_groups(PyObject* self, PyObject* args) {
    PyObject *go, *gn, *out;
    int gtot;
    if (!PyArg_ParseTuple(args, "OOiO", &go, &gn, &gtot, &out)) return NULL;
    Py_buffer a, b, o;
    long rc = -2;
    if (PyObject_GetBuffer(go, &a, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(gn, &b, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&a); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(out, &o, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&a); PyBuffer_Release(&b); return PyLong_FromLong(-2); }
    if (a.itemsize == 4 && b.itemsize == 4 && a.len == b.len && o.itemsize == 4 && gtot >= 0 && o.len >= 4 * (Py_ssize_t)gtot) {
        int32_t nt = 0;
        rc = sbeh_touched_groups((const int32_t*)a.buf, (const int32_t*)b.buf, (int64_t)(a.len / 4), gtot, (int32_t*)o.buf, &nt);
        if (rc == 0) rc = nt;
    }
    PyBuffer_Release(&a); PyBuffer_Release(&b); PyBuffer_Release(&o);
    return PyLong_FromLong(rc);
}

static PyMethodDef methods[] = {
    {"addr", py_addr, METH_O, "buffer address of an array (any strides), as int"},
    {"subset_ids", py_subset_ids, METH_VARARGS, "ids of the listed objects for sbe_counts_delta (sbeh_subset_ids)"},
    {"diff_rows", py_diff_rows, METH_VARARGS, "rows of `new` differing from `mirror`, copied into it (sbeh_diff_rows)"},
    {"touched_groups", py_touched_groups, METH_VARARGS, "sorted distinct group indices among two id arrays (sbeh_touched_groups)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_sbe_pyhost", "host-layer helpers of sbayes_amd (no device code)", -1, methods};

PyMODINIT_FUNC PyInit__sbe_pyhost(void) { return PyModule_Create(&moduledef); }
