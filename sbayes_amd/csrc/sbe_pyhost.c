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

/* a C-contiguous buffer with `ndim` dimensions of 1-byte items (bool / int8 / uint8) or of 4-byte INTEGERS (the ids: a float32
   array has the same item size and must not pass); 0 on mismatch */
static int get_c(PyObject* obj, Py_buffer* v, int ndim, Py_ssize_t itemsize, int writable) {
    if (PyObject_GetBuffer(obj, v, (writable ? PyBUF_WRITABLE : 0) | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); return 0; }
    int ok = v->ndim == ndim && v->itemsize == itemsize;
    if (ok && v->format) {
        const char* f = v->format;
        const char code = f[strlen(f) ? strlen(f) - 1 : 0];              /* (a byte-order prefix may precede the type code) */
        ok = itemsize == 1 ? (code == '?' || code == 'B' || code == 'b') : (code == 'i' || code == 'l' || code == 'I' || code == 'L');
    }
    if (!ok) { PyBuffer_Release(v); return 0; }
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

/* ---- the bind cache's tokens (sbayes_amd/binding.py: _token / _same) -----------------------------------------------------
 * scan(params, cached) -> (tokens, changed): for every state parameter (or plain array) of `params` its token
 * (array, version) -- `param.value` if it has one, else the object itself, made an ndarray; `param.version` or None -- and in
 * `changed` bit i set unless the token denotes what cached[i] = (array, version, private copy) | None recorded:
 *   versioned parameters   same ndarray OBJECT and equal version (an in-place edit through the parameter API bumps the
 *                          version, a copy-on-write edit creates a new ndarray; sbayes/sampling/state.py:34-61, 97-161, 340-350)
 *   unversioned arrays     the same object, recorded without a copy because it was frozen and owned its data, and still so;
 *                          otherwise compared by content against the private copy (the Python callback)
 * Thirteen parameters per bind and five binds per MCMC step: as Python this was 13 us per bind (tools/host_residual.py). */
static PyObject *s_value, *s_version, *s_flags, *s_writeable, *s_owndata;
static PyObject *g_ndarray = NULL, *g_asarray = NULL, *g_content_equal = NULL;

static PyObject* py_scan_setup(PyObject* self, PyObject* args) {
    PyObject *nd, *asarr, *ceq;
    if (!PyArg_ParseTuple(args, "OOO", &nd, &asarr, &ceq)) return NULL;
    Py_XDECREF(g_ndarray); Py_XDECREF(g_asarray); Py_XDECREF(g_content_equal);
    Py_INCREF(nd); Py_INCREF(asarr); Py_INCREF(ceq);
    g_ndarray = nd; g_asarray = asarr; g_content_equal = ceq;
    Py_RETURN_NONE;
}

/* 1 same, 0 differs, -1 error */
static int token_same(PyObject* value, PyObject* version, PyObject* cached) {
    if (cached == Py_None) return 0;
    if (!PyTuple_Check(cached) || PyTuple_GET_SIZE(cached) != 3) { PyErr_SetString(PyExc_TypeError, "cached entry must be a 3-tuple or None"); return -1; }
    PyObject* ref = PyTuple_GET_ITEM(cached, 0);
    PyObject* ref_version = PyTuple_GET_ITEM(cached, 1);
    PyObject* copy = PyTuple_GET_ITEM(cached, 2);
    if (version != Py_None && ref_version != Py_None) {
        if (value != ref) return 0;
        return PyObject_RichCompareBool(version, ref_version, Py_EQ);
    }
    if (value == ref && copy == Py_None) {              /* recorded frozen: still frozen and owning its data? */
        PyObject* flags = PyObject_GetAttr(value, s_flags);
        if (!flags) return -1;
        PyObject* w = PyObject_GetAttr(flags, s_writeable);
        PyObject* o = w ? PyObject_GetAttr(flags, s_owndata) : NULL;
        Py_DECREF(flags);
        if (!w || !o) { Py_XDECREF(w); Py_XDECREF(o); return -1; }
        const int wr = PyObject_IsTrue(w), own = PyObject_IsTrue(o);
        Py_DECREF(w); Py_DECREF(o);
        if (wr < 0 || own < 0) return -1;
        return !wr && own;
    }
    if (copy == Py_None) return 0;
    PyObject* r = PyObject_CallFunctionObjArgs(g_content_equal, value, copy, NULL);
    if (!r) return -1;
    const int eq = PyObject_IsTrue(r);
    Py_DECREF(r);
    return eq;
}

static PyObject* py_scan(PyObject* self, PyObject* args) {
    PyObject *params, *cached;
    if (!PyArg_ParseTuple(args, "OO", &params, &cached)) return NULL;
    if (!g_ndarray) { PyErr_SetString(PyExc_RuntimeError, "scan_setup() not called"); return NULL; }
    if (!PyList_Check(params) || !PyList_Check(cached) || PyList_GET_SIZE(params) != PyList_GET_SIZE(cached) || PyList_GET_SIZE(params) > 62) {
        PyErr_SetString(PyExc_TypeError, "scan(params, cached): two lists of equal length (at most 62)");
        return NULL;
    }
    const Py_ssize_t n = PyList_GET_SIZE(params);
    PyObject* tokens = PyList_New(n);
    if (!tokens) return NULL;
    unsigned long long changed = 0;
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* param = PyList_GET_ITEM(params, i);
        PyObject *value, *version;
        if ((PyObject*)Py_TYPE(param) == g_ndarray) {       /* a plain array (group matrix, concentration table): no version */
            value = param; Py_INCREF(value);
            version = Py_None; Py_INCREF(version);
        } else {
            value = PyObject_GetAttr(param, s_value);
            if (!value) {
                if (!PyErr_ExceptionMatches(PyExc_AttributeError)) goto fail;
                PyErr_Clear();
                value = param; Py_INCREF(value);
            }
            if ((PyObject*)Py_TYPE(value) != g_ndarray) {
                PyObject* conv = PyObject_CallFunctionObjArgs(g_asarray, value, NULL);
                Py_DECREF(value);
                if (!conv) goto fail;
                value = conv;
            }
            version = PyObject_GetAttr(param, s_version);
            if (!version) {
                if (!PyErr_ExceptionMatches(PyExc_AttributeError)) { Py_DECREF(value); goto fail; }
                PyErr_Clear();
                version = Py_None; Py_INCREF(version);
            }
        }
        const int same = token_same(value, version, PyList_GET_ITEM(cached, i));
        PyObject* tok = same < 0 ? NULL : PyTuple_Pack(2, value, version);
        Py_DECREF(value); Py_DECREF(version);
        if (!tok) goto fail;
        PyList_SET_ITEM(tokens, i, tok);
        if (!same) changed |= 1ull << i;
    }
    {
        PyObject* r = Py_BuildValue("(NK)", tokens, changed);
        return r;
    }
fail:
    Py_DECREF(tokens);
    return NULL;
}

static PyMethodDef methods[] = {
    {"scan_setup", py_scan_setup, METH_VARARGS, "scan_setup(ndarray_type, asarray, content_equal)"},
    {"scan", py_scan, METH_VARARGS, "scan(params, cached) -> (tokens, changed bitmask): the bind cache's token comparison"},
    {"addr", py_addr, METH_O, "buffer address of an array (any strides), as int"},
    {"subset_ids", py_subset_ids, METH_VARARGS, "ids of the listed objects for sbe_counts_delta (sbeh_subset_ids)"},
    {"diff_rows", py_diff_rows, METH_VARARGS, "rows of `new` differing from `mirror`, copied into it (sbeh_diff_rows)"},
    {"touched_groups", py_touched_groups, METH_VARARGS, "sorted distinct group indices among two id arrays (sbeh_touched_groups)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_sbe_pyhost", "host-layer helpers of sbayes_amd (no device code)", -1, methods};

PyMODINIT_FUNC PyInit__sbe_pyhost(void) {
    s_value = PyUnicode_InternFromString("value");
    s_version = PyUnicode_InternFromString("version");
    s_flags = PyUnicode_InternFromString("flags");
    s_writeable = PyUnicode_InternFromString("writeable");
    s_owndata = PyUnicode_InternFromString("owndata");
    if (!s_value || !s_version || !s_flags || !s_writeable || !s_owndata) return NULL;
    return PyModule_Create(&moduledef);
}
