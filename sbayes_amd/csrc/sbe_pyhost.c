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

/* diff_rows_among(new, mirror, cand_a, cand_b | None, idx_out) -> number of differing rows among the listed ones (copied into
 * mirror, indices ascending in idx_out); -1 bad index, -2 not in ABI form */
static PyObject* py_diff_rows_among(PyObject* self, PyObject* args) {
    PyObject *a, *m, *ca, *cb, *idx;
    if (!PyArg_ParseTuple(args, "OOOOO", &a, &m, &ca, &cb, &idx)) return NULL;
    Py_buffer va, vm, vi, v1, v2;
    long long rc = -2;
    int have2 = 0;
    if (PyObject_GetBuffer(a, &va, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(m, &vm, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&va); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(idx, &vi, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&va); PyBuffer_Release(&vm); return PyLong_FromLong(-2); }
    if (!get_c(ca, &v1, 1, 4, 0)) { PyBuffer_Release(&va); PyBuffer_Release(&vm); PyBuffer_Release(&vi); return PyLong_FromLong(-2); }
    if (cb != Py_None) {
        if (!get_c(cb, &v2, 1, 4, 0)) { PyBuffer_Release(&va); PyBuffer_Release(&vm); PyBuffer_Release(&vi); PyBuffer_Release(&v1); return PyLong_FromLong(-2); }
        have2 = 1;
    }
    const Py_ssize_t na = v1.shape[0], nb = have2 ? v2.shape[0] : 0;
    if (va.ndim >= 1 && va.len == vm.len && va.shape[0] > 0 && vi.itemsize == 4 && vi.len >= 4 * (na + nb))
        rc = sbeh_diff_rows_among(va.buf, vm.buf, (int64_t)va.shape[0], (int64_t)(va.len / va.shape[0]), (const int32_t*)v1.buf, (int64_t)na,
                                  have2 ? (const int32_t*)v2.buf : NULL, (int64_t)nb, (int32_t*)vi.buf);
    PyBuffer_Release(&va); PyBuffer_Release(&vm); PyBuffer_Release(&vi); PyBuffer_Release(&v1);
    if (have2) PyBuffer_Release(&v2);
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

/* ---- trusted classes: their read-only properties are read where they live --------------------------------------------------------
 * A bind walks thirteen parameters: sample.clusters / .weights / .source / .feature_counts, every parameter's .value, every
 * confounder prior's concentration_array(sample) -- Python-level properties and one method that return an instance attribute
 * (sbayes/sampling/state.py:30-32, 578-592; sbayes/model/prior.py:325-354 for a static prior), 0.4-1 us each where they run, four
 * binds per MCMC step.  For objects of EXACTLY the classes registered here (trust_setup: sbayes_amd.state's, and the reference's
 * when the source of those properties is the revision mirrored -- sbayes_amd/patch.py) the attribute is taken from the instance
 * dictionary instead; any other class, or an instance without that entry, goes through the public name as before. */
static PyObject *s_shared, *s_resolve_sharing, *s__value, *s_group_versions, *s_copy_method;
static PyObject *g_trust_samples = NULL, *g_trust_params = NULL, *g_trust_conf_priors = NULL, *g_trust_counts = NULL;
static PyObject *s__clusters, *s__weights, *s__source, *s__feature_counts, *s__concentration_array, *s_any_dynamic_priors_attr;

static PyObject* py_trust_setup(PyObject* self, PyObject* args) {
    PyObject *a, *b, *c, *d;
    if (!PyArg_ParseTuple(args, "O!O!O!O!", &PyTuple_Type, &a, &PyTuple_Type, &b, &PyTuple_Type, &c, &PyTuple_Type, &d)) return NULL;
    Py_XDECREF(g_trust_samples); Py_XDECREF(g_trust_params); Py_XDECREF(g_trust_conf_priors); Py_XDECREF(g_trust_counts);
    Py_INCREF(a); Py_INCREF(b); Py_INCREF(c); Py_INCREF(d);
    g_trust_samples = a; g_trust_params = b; g_trust_conf_priors = c; g_trust_counts = d;
    Py_RETURN_NONE;
}

static inline int type_in(PyObject* types, PyObject* obj) {
    if (!types) return 0;
    const Py_ssize_t n = PyTuple_GET_SIZE(types);
    for (Py_ssize_t i = 0; i < n; ++i) if ((PyObject*)Py_TYPE(obj) == PyTuple_GET_ITEM(types, i)) return 1;
    return 0;
}

/* new reference to obj.__dict__[name], or NULL without an exception */
static inline PyObject* inst_get(PyObject* obj, PyObject* name) {
    PyObject** dp = _PyObject_GetDictPtr(obj);
    if (!dp || !*dp) return NULL;
    PyObject* v = PyDict_GetItemWithError(*dp, name);
    if (!v) { PyErr_Clear(); return NULL; }
    Py_INCREF(v);
    return v;
}

/* param.value (new reference) */
static inline PyObject* param_value(PyObject* param) {
    if (type_in(g_trust_params, param)) {
        PyObject* v = inst_get(param, s__value);
        if (v) return v;
    }
    return PyObject_GetAttr(param, s_value);
}

/* sample.<public> (new reference) */
static inline PyObject* sample_attr(PyObject* sample, PyObject* private_name, PyObject* public_name) {
    if (type_in(g_trust_samples, sample)) {
        PyObject* v = inst_get(sample, private_name);
        if (v) return v;
    }
    return PyObject_GetAttr(sample, public_name);
}

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


/* ---- FeatureCounts.add_changes for a difference given as rows (sbayes/sampling/counts.py:77, :93 -> state.py:340-350) --------
 * add_rows_many(nodes, off, touched, rows) -> bounds [C + 1] | None
 * `touched` (int32, ascending global group indices) and `rows` (float32 [k, F, S]) are what sbe_counts_delta returned; component
 * c owns the rows whose index lies in [off[c], off[c + 1]).  Per node exactly what add_changes_rows (sbayes_amd/state.py, and the
 * method patch.install gives the reference's class) does -- resolve_sharing() if shared, value[group] += row, version += 1,
 * group_versions[group] = version where the row is not all zero -- as ONE call instead of three Python method calls with two
 * fancy-index operations and a reduction each (60 us per MCMC step at the south_america shape).  None: a node that is not in that
 * form (the caller keeps the Python route). */

static int off_at(const Py_buffer* v, Py_ssize_t i, long long* out) {
    if (v->itemsize == 8) { *out = ((const long long*)v->buf)[i]; return 1; }
    if (v->itemsize == 4) { *out = ((const int*)v->buf)[i]; return 1; }
    return 0;
}

static PyObject* py_add_rows_many(PyObject* self, PyObject* args) {
    PyObject *nodes, *off, *touched, *rows;
    if (!PyArg_ParseTuple(args, "OOOO", &nodes, &off, &touched, &rows)) return NULL;
    if (!PyList_Check(nodes)) Py_RETURN_NONE;
    const Py_ssize_t C = PyList_GET_SIZE(nodes);
    Py_buffer vo, vt, vr;
    if (PyObject_GetBuffer(off, &vo, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); Py_RETURN_NONE; }
    if (PyObject_GetBuffer(touched, &vt, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); PyBuffer_Release(&vo); Py_RETURN_NONE; }
    if (PyObject_GetBuffer(rows, &vr, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); PyBuffer_Release(&vo); PyBuffer_Release(&vt); Py_RETURN_NONE; }
    PyObject* result = NULL;
    int unsupported = 0;
    const Py_ssize_t k = vt.ndim == 1 ? vt.shape[0] : -1;
    if (vo.ndim != 1 || vo.shape[0] < C + 1 || (vo.itemsize != 8 && vo.itemsize != 4) || k < 0 || vt.itemsize != 4 || vr.itemsize != 4 || !vr.format || vr.format[strlen(vr.format) - 1] != 'f' ||
        (k > 0 && (vr.ndim != 3 || vr.shape[0] != k))) unsupported = 1;
    const Py_ssize_t row_elems = (!unsupported && k > 0) ? vr.shape[1] * vr.shape[2] : 0;
    const int* t = (const int*)vt.buf;
    const float* r = (const float*)vr.buf;
    if (unsupported) { result = Py_None; Py_INCREF(result); goto done; }
    /* component c owns touched[bounds[c] .. bounds[c + 1]) */
    result = PyList_New(C + 1);
    if (!result) goto done;
    {
        Py_ssize_t pos = 0;
        for (Py_ssize_t c = 0; c <= C; ++c) {
            long long o = 0; off_at(&vo, c, &o);
            while (pos < k && t[pos] < o) ++pos;
            PyObject* b = PyLong_FromSsize_t(pos);
            if (!b) { Py_CLEAR(result); goto done; }
            PyList_SET_ITEM(result, c, b);
        }
    }
    /* pass 1: EVERY node must be in the expected form -- attributes, buffer dtype / contiguity / shape, group indices in range --
       BEFORE anything is changed (a node in another form sends the whole call down the Python route; no component is
       ever left half-applied).  resolve_sharing() replaces _value by a copy of the same form, so the check holds for pass 2. */
    for (Py_ssize_t c = 0; c < C && !unsupported; ++c) {
        PyObject* node = PyList_GET_ITEM(nodes, c);
        const Py_ssize_t lo = PyLong_AsSsize_t(PyList_GET_ITEM(result, c)), hi = PyLong_AsSsize_t(PyList_GET_ITEM(result, c + 1));
        long long o = 0; off_at(&vo, c, &o);
        PyObject* val = PyObject_GetAttr(node, s__value);
        PyObject* gv = val ? PyObject_GetAttr(node, s_group_versions) : NULL;
        PyObject* ver = gv ? PyObject_GetAttr(node, s_version) : NULL;
        if (!val || !gv || !ver || !PyLong_Check(ver) || !PyObject_HasAttr(node, s_shared) || !PyObject_HasAttr(val, s_flags)) { PyErr_Clear(); unsupported = 1; }
        if (!unsupported) {
            Py_buffer bv, bg;
            if (PyObject_GetBuffer(val, &bv, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); unsupported = 1; }
            else {
                if (PyObject_GetBuffer(gv, &bg, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT | PyBUF_WRITABLE) != 0) { PyErr_Clear(); unsupported = 1; }
                else {
                    const char gf = bg.format ? bg.format[strlen(bg.format) - 1] : 0;
                    const int form_ok = bv.ndim == 3 && bv.itemsize == 4 && bv.format && bv.format[strlen(bv.format) - 1] == 'f' &&
                                        (hi == lo || bv.shape[1] * bv.shape[2] == row_elems) && bg.ndim == 1 && bg.itemsize == 8 &&
                                        bg.shape[0] == bv.shape[0] && (gf == 'd' || gf == 'l' || gf == 'q');
                    if (!form_ok) unsupported = 1;
                    for (Py_ssize_t i = lo; i < hi && !unsupported; ++i) {
                        const long long g = (long long)t[i] - o;
                        if (g < 0 || g >= bv.shape[0]) unsupported = 1;
                    }
                    PyBuffer_Release(&bg);
                }
                PyBuffer_Release(&bv);
            }
        }
        Py_XDECREF(val); Py_XDECREF(gv); Py_XDECREF(ver);
    }
    if (unsupported) { Py_CLEAR(result); result = Py_None; Py_INCREF(result); goto done; }
    /* pass 2: the update (state.py:340-350 per node) */
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject* node = PyList_GET_ITEM(nodes, c);
        const Py_ssize_t lo = PyLong_AsSsize_t(PyList_GET_ITEM(result, c)), hi = PyLong_AsSsize_t(PyList_GET_ITEM(result, c + 1));
        long long o = 0; off_at(&vo, c, &o);
        PyObject* shared = PyObject_GetAttr(node, s_shared);
        if (!shared) { Py_CLEAR(result); goto done; }
        const int is_shared = PyObject_IsTrue(shared);
        Py_DECREF(shared);
        if (is_shared < 0) { Py_CLEAR(result); goto done; }
        if (is_shared) {
            if (type_in(g_trust_counts, node)) {
                /* GroupedParameters.resolve_sharing (state.py:166-168 over :82-84) of a registered count class:
                   group_versions = group_versions.copy(); _value = _value.copy(); shared = False */
                PyObject* gv = PyObject_GetAttr(node, s_group_versions);
                PyObject* gv2 = gv ? PyObject_CallMethodNoArgs(gv, s_copy_method) : NULL;
                PyObject* v0 = gv2 ? PyObject_GetAttr(node, s__value) : NULL;
                PyObject* v2 = v0 ? PyObject_CallMethodNoArgs(v0, s_copy_method) : NULL;
                const int bad = !v2 || PyObject_SetAttr(node, s_group_versions, gv2) != 0 || PyObject_SetAttr(node, s__value, v2) != 0 ||
                                PyObject_SetAttr(node, s_shared, Py_False) != 0;
                Py_XDECREF(gv); Py_XDECREF(gv2); Py_XDECREF(v0); Py_XDECREF(v2);
                if (bad) { Py_CLEAR(result); goto done; }
            } else {
                PyObject* rr = PyObject_CallMethodNoArgs(node, s_resolve_sharing);
                if (!rr) { Py_CLEAR(result); goto done; }
                Py_DECREF(rr);
            }
        }
        PyObject* ver = PyObject_GetAttr(node, s_version);
        if (!ver) { Py_CLEAR(result); goto done; }
        const long long version = PyLong_AsLongLong(ver) + 1;
        Py_DECREF(ver);
        PyObject* val = PyObject_GetAttr(node, s__value);
        if (!val) { Py_CLEAR(result); goto done; }
        int ok = 1;
        if (hi > lo) {
            PyObject* gv = PyObject_GetAttr(node, s_group_versions);
            Py_buffer bv, bg;
            ok = gv && PyObject_GetBuffer(val, &bv, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0;    /* (kept read-only between edits: written through anyway, like the flag toggle of the Python form) */
            if (ok && PyObject_GetBuffer(gv, &bg, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT | PyBUF_WRITABLE) != 0) { PyBuffer_Release(&bv); ok = 0; }
            if (ok) {
                const char gf = bg.format ? bg.format[strlen(bg.format) - 1] : 0;
                /* (re-checked: resolve_sharing is a Python method and may be anything) */
                const int form_ok = bv.ndim == 3 && bv.itemsize == 4 && bv.format && bv.format[strlen(bv.format) - 1] == 'f' &&
                                    bv.shape[1] * bv.shape[2] == row_elems && bg.ndim == 1 && bg.itemsize == 8 && bg.shape[0] == bv.shape[0] &&
                                    (gf == 'd' || gf == 'l' || gf == 'q');
                if (form_ok) {
                    for (Py_ssize_t i = lo; i < hi && ok; ++i) {
                        const long long g = (long long)t[i] - o;
                        if (g < 0 || g >= bv.shape[0]) { PyErr_SetString(PyExc_IndexError, "add_rows_many: group index out of range"); ok = 0; break; }
                        float* dst = (float*)bv.buf + g * row_elems;
                        const float* src = r + i * row_elems;
                        int nz = 0;
                        for (Py_ssize_t e = 0; e < row_elems; ++e) { dst[e] += src[e]; nz |= src[e] != 0.0f; }
                        if (nz) { if (gf == 'd') ((double*)bg.buf)[g] = (double)version; else ((long long*)bg.buf)[g] = version; }
                    }
                } else {
                    PyErr_SetString(PyExc_TypeError, "add_rows_many: count node changed form between the check and the update");
                    ok = 0;
                }
                PyBuffer_Release(&bv); PyBuffer_Release(&bg);
            } else if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "add_rows_many: count arrays are not plain C-contiguous buffers");
            Py_XDECREF(gv);
        }
        /* _value.flags.writeable = False, as add_changes leaves it (state.py:347): the copy resolve_sharing() made is writeable */
        if (ok) {
            PyObject* flags = PyObject_GetAttr(val, s_flags);
            if (!flags || PyObject_SetAttr(flags, s_writeable, Py_False) != 0) ok = 0;
            Py_XDECREF(flags);
        }
        Py_DECREF(val);
        if (!ok) { Py_CLEAR(result); goto done; }
        PyObject* nv = PyLong_FromLongLong(version);
        if (!nv || PyObject_SetAttr(node, s_version, nv) != 0) { Py_XDECREF(nv); Py_CLEAR(result); goto done; }
        Py_DECREF(nv);
    }
done:
    PyBuffer_Release(&vo); PyBuffer_Release(&vt); PyBuffer_Release(&vr);
    return result;
}

/* copy_rows(dst, src, idx): dst[idx] = src[idx] for two C-contiguous arrays of one shape and item size (rows = everything
 * behind the first axis); idx int32.  The bind cache's mirrors follow the sample this way (binding.counts_followed). */
static PyObject* py_copy_rows(PyObject* self, PyObject* args) {
    PyObject *dst, *src, *idx;
    if (!PyArg_ParseTuple(args, "OOO", &dst, &src, &idx)) return NULL;
    Py_buffer vd, vs, vi;
    long rc = -2;
    if (PyObject_GetBuffer(dst, &vd, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(src, &vs, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&vd); return PyLong_FromLong(-2); }
    if (PyObject_GetBuffer(idx, &vi, PyBUF_C_CONTIGUOUS) != 0) { PyErr_Clear(); PyBuffer_Release(&vd); PyBuffer_Release(&vs); return PyLong_FromLong(-2); }
    if (vd.ndim >= 1 && vd.ndim == vs.ndim && vd.len == vs.len && vd.itemsize == vs.itemsize && vd.shape[0] == vs.shape[0] && vd.shape[0] > 0 && vi.itemsize == 4) {
        const Py_ssize_t row = vd.len / vd.shape[0], n = vi.len / 4;
        const int* ix = (const int*)vi.buf;
        rc = 0;
        for (Py_ssize_t i = 0; i < n; ++i) {
            if (ix[i] < 0 || ix[i] >= vd.shape[0]) { rc = -1; break; }
            memcpy((char*)vd.buf + (Py_ssize_t)ix[i] * row, (const char*)vs.buf + (Py_ssize_t)ix[i] * row, (size_t)row);
        }
    }
    PyBuffer_Release(&vd); PyBuffer_Release(&vs); PyBuffer_Release(&vi);
    return PyLong_FromLong(rc);
}

/* ---- binding._bind_slot in native code -----------------------------------------------------------------------------------------
 * bind_slot(eng, model, sample, slot, with_source) -> the stale components (a set), or NotImplemented when the engine keeps no
 * bind cache (test doubles: the Python form sends everything).  A line-by-line transcription of the Python function (kept in
 * binding.py as the reference form and the fallback: SBAYES_AMD_NO_PYHOST): the same comparisons, the same engine calls with the
 * same arguments in the same order, the same entry / mirror dictionaries.  What it saves is the interpreter: ~90 bytecode lines
 * per bind, five binds per MCMC step (tools/host_residual.py: 100 us of a 450 us host layer at the south_america shape).
 * The content helpers stay Python callables handed over by bind_setup: _send_counts, _send_source, _remember (unversioned arrays),
 * numpy.concatenate. */
static PyObject *g_send_counts = NULL, *g_send_source = NULL, *g_remember = NULL, *g_concatenate = NULL, *g_bind_py = NULL;
static int g_rows_with_probs = 1;
static PyObject *s_confounders, *s_clusters, *s_group_assignment, *s_prior, *s_prior_confounding_effects, *s_prior_cluster_effect,
                *s_concentration_array, *s_feature_counts, *s_weights, *s_source, *s__bound, *s__bound_conc, *s__mirror, *s__touch,
                *s_set_concentration, *s_set_groups, *s_set_slot_delta, *s_set_counts_rows, *s_set_source_rows, *s_set_weights,
                *k_groups, *k_counts, *k_weights, *k_source, *k_stale, *k_lh_all, *k_update_probs,
                *k_groups_component, *k_count_idx, *k_count_rows, *k_source_objects, *k_source_rows;

/* (shared with update_counts further down: np.empty and the dtypes, set by update_counts_setup) */
static PyObject *g_np_empty = NULL, *g_dt_i32 = NULL, *g_dt_u8 = NULL, *g_get_engine = NULL, *g_note_lineage = NULL, *g_source_followed = NULL,
                *g_apply_rows = NULL;
static PyObject *s_counts_delta, *s_group_offsets, *k_follow_slot, *k_update_source;

static PyObject* py_update_counts_setup(PyObject* self, PyObject* args) {
    PyObject *e, *di, *du, *ge, *nl, *sf, *ar;
    if (!PyArg_ParseTuple(args, "OOOOOOO", &e, &di, &du, &ge, &nl, &sf, &ar)) return NULL;
    Py_XDECREF(g_np_empty); Py_XDECREF(g_dt_i32); Py_XDECREF(g_dt_u8); Py_XDECREF(g_get_engine); Py_XDECREF(g_note_lineage); Py_XDECREF(g_source_followed); Py_XDECREF(g_apply_rows);
    Py_INCREF(e); Py_INCREF(di); Py_INCREF(du); Py_INCREF(ge); Py_INCREF(nl); Py_INCREF(sf); Py_INCREF(ar);
    g_np_empty = e; g_dt_i32 = di; g_dt_u8 = du; g_get_engine = ge; g_note_lineage = nl; g_source_followed = sf; g_apply_rows = ar;
    Py_RETURN_NONE;
}

/* np.empty((a, b), dtype) / np.empty((a,), dtype): new reference */
static PyObject* np_empty2(Py_ssize_t a, Py_ssize_t b, PyObject* dtype) {
    PyObject* shape = b < 0 ? Py_BuildValue("(n)", a) : Py_BuildValue("(nn)", a, b);
    if (!shape) return NULL;
    PyObject* r = PyObject_CallFunctionObjArgs(g_np_empty, shape, dtype, NULL);
    Py_DECREF(shape);
    return r;
}

static PyObject* py_bind_setup(PyObject* self, PyObject* args) {
    PyObject *sc, *ss, *rm, *cc, *bp;
    int rwp;
    if (!PyArg_ParseTuple(args, "OOOOpO", &sc, &ss, &rm, &cc, &rwp, &bp)) return NULL;
    Py_XDECREF(g_send_counts); Py_XDECREF(g_send_source); Py_XDECREF(g_remember); Py_XDECREF(g_concatenate); Py_XDECREF(g_bind_py);
    Py_INCREF(sc); Py_INCREF(ss); Py_INCREF(rm); Py_INCREF(cc); Py_INCREF(bp);
    g_send_counts = sc; g_send_source = ss; g_remember = rm; g_concatenate = cc; g_rows_with_probs = rwp; g_bind_py = bp;
    Py_RETURN_NONE;
}

/* _remember(token): (array, version, None) for a versioned parameter -- the common case, built here -- else the Python helper
   (frozen arrays: identity; others: a private copy) */
static PyObject* remember_token(PyObject* tok) {
    PyObject* version = PyTuple_GET_ITEM(tok, 1);
    if (version != Py_None) return PyTuple_Pack(3, PyTuple_GET_ITEM(tok, 0), version, Py_None);
    return PyObject_CallFunctionObjArgs(g_remember, tok, NULL);
}

/* one token (value, version) of a parameter and whether it denotes `cached`; -1 on error */
static int token_of(PyObject* param, PyObject* cached, PyObject** tok_out) {
    PyObject *value, *version;
    if ((PyObject*)Py_TYPE(param) == g_ndarray) {
        value = param; Py_INCREF(value);
        version = Py_None; Py_INCREF(version);
    } else {
        value = type_in(g_trust_params, param) ? inst_get(param, s__value) : NULL;
        if (!value) value = PyObject_GetAttr(param, s_value);
        if (!value) {
            if (!PyErr_ExceptionMatches(PyExc_AttributeError)) return -1;
            PyErr_Clear();
            value = param; Py_INCREF(value);
        }
        if ((PyObject*)Py_TYPE(value) != g_ndarray) {
            PyObject* conv = PyObject_CallFunctionObjArgs(g_asarray, value, NULL);
            Py_DECREF(value);
            if (!conv) return -1;
            value = conv;
        }
        version = PyObject_GetAttr(param, s_version);
        if (!version) {
            if (!PyErr_ExceptionMatches(PyExc_AttributeError)) { Py_DECREF(value); return -1; }
            PyErr_Clear();
            version = Py_None; Py_INCREF(version);
        }
    }
    const int same = token_same(value, version, cached);
    *tok_out = same < 0 ? NULL : PyTuple_Pack(2, value, version);
    Py_DECREF(value); Py_DECREF(version);
    if (!*tok_out) return -1;
    return same;
}

#define BIND_MAXC 16
/* an engine without a bind cache (test doubles), or more components than the fixed arrays hold: the Python form */
static PyObject* bind_python_form(PyObject* eng, PyObject* model, PyObject* sample, PyObject* slot, int with_source) {
    return PyObject_CallFunctionObjArgs(g_bind_py, eng, model, sample, slot, with_source ? Py_True : Py_False, NULL);
}

static PyObject* py_bind_slot(PyObject* self, PyObject* args, PyObject* kwargs) {
    PyObject *eng, *model, *sample, *slot;
    int with_source = 0;
    static char* kwlist[] = {"eng", "model", "sample", "slot", "with_source", NULL};
    if (!PyArg_ParseTupleAndKeywords(args, kwargs, "OOOO|p", kwlist, &eng, &model, &sample, &slot, &with_source)) return NULL;
    if (!g_send_counts || !g_ndarray) { PyErr_SetString(PyExc_RuntimeError, "bind_setup() / scan_setup() not called"); return NULL; }
    PyObject* cache = PyObject_GetAttr(eng, s__bound);
    if (!cache) { PyErr_Clear(); return bind_python_form(eng, model, sample, slot, with_source); }
    if (!PyDict_Check(cache)) { Py_DECREF(cache); return bind_python_form(eng, model, sample, slot, with_source); }
    /* every owned reference of the function lives in `own` and is released at `out` */
    PyObject* own[64 + 24 * BIND_MAXC]; int n_own = 0;
#define OWN(x) (own[n_own++] = (x))
    PyObject* result = NULL;
    int in_send = 0;
    PyObject *tokens[3 * BIND_MAXC + 2]; int n_tok = 0;
    memset(tokens, 0, sizeof tokens);
    OWN(cache);
    PyObject* confounders = OWN(PyObject_GetAttr(sample, s_confounders));
    if (!confounders) goto out;
    PyObject* conf_names = OWN(PySequence_List(confounders));
    if (!conf_names) goto out;
    const Py_ssize_t C = 1 + PyList_GET_SIZE(conf_names);
    if (C > BIND_MAXC) { result = bind_python_form(eng, model, sample, slot, with_source); goto out; }
    PyObject* old = PyDict_GetItemWithError(cache, slot);                 /* borrowed */
    int old_is_new = 0;
    if (!old) {
        if (PyErr_Occurred()) goto out;
        old = OWN(PyDict_New());
        if (!old) goto out;
        old_is_new = 1;
        PyObject* g = OWN(PyList_New(C)); PyObject* cn = OWN(PyList_New(C)); PyObject* st = OWN(PySet_New(NULL));
        if (!g || !cn || !st) goto out;
        for (Py_ssize_t c = 0; c < C; ++c) {
            Py_INCREF(Py_None); PyList_SET_ITEM(g, c, Py_None);
            Py_INCREF(Py_None); PyList_SET_ITEM(cn, c, Py_None);
            PyObject* ci = PyLong_FromSsize_t(c);
            if (!ci || PySet_Add(st, ci) != 0) { Py_XDECREF(ci); goto out; }
            Py_DECREF(ci);
        }
        if (PyDict_SetItem(old, k_groups, g) || PyDict_SetItem(old, k_counts, cn) || PyDict_SetItem(old, k_weights, Py_None) ||
            PyDict_SetItem(old, k_source, Py_None) || PyDict_SetItem(old, k_stale, st) || PyDict_SetItem(old, k_lh_all, Py_None)) goto out;
    } else if (!PyDict_Check(old)) { PyErr_SetString(PyExc_TypeError, "bind entry is not a dict"); goto out; }
    else { Py_INCREF(old); OWN(old); }          /* (the engine's setters drop the slot's entry from the cache while it is still read here) */
    PyObject* old_groups = PyDict_GetItem(old, k_groups);                 /* borrowed */
    PyObject* old_counts = PyDict_GetItem(old, k_counts);
    PyObject* old_stale = PyDict_GetItem(old, k_stale);
    if (!old_groups || !old_counts || !old_stale || !PyList_Check(old_groups) || !PyList_Check(old_counts) ||
        PyList_GET_SIZE(old_groups) != C || PyList_GET_SIZE(old_counts) != C) { PyErr_SetString(PyExc_TypeError, "malformed bind entry"); goto out; }
    /* ---- what differs? ---- */
    unsigned long long changed = 0;
    PyObject *prior = NULL, *conf_priors = NULL, *feature_counts = NULL, *bound_conc = NULL;
    const int have_model = model != Py_None;
    {
        PyObject* p = OWN(sample_attr(sample, s__clusters, s_clusters));
        if (!p) goto out;
        int same = token_of(p, PyList_GET_ITEM(old_groups, 0), &tokens[n_tok]);
        if (same < 0) goto out;
        OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        for (Py_ssize_t c = 1; c < C; ++c) {
            PyObject* conf = OWN(PyObject_GetItem(confounders, PyList_GET_ITEM(conf_names, c - 1)));
            if (!conf) goto out;
            PyObject* ga = OWN(PyObject_GetAttr(conf, s_group_assignment));
            if (!ga) goto out;
            same = token_of(ga, PyList_GET_ITEM(old_groups, c), &tokens[n_tok]);
            if (same < 0) goto out;
            OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        }
    }
    if (have_model) {
        prior = OWN(PyObject_GetAttr(model, s_prior));
        if (!prior) goto out;
        conf_priors = OWN(PyObject_GetAttr(prior, s_prior_confounding_effects));
        feature_counts = conf_priors ? OWN(sample_attr(sample, s__feature_counts, s_feature_counts)) : NULL;
        bound_conc = feature_counts ? OWN(PyObject_GetAttr(eng, s__bound_conc)) : NULL;
        if (!bound_conc) goto out;
        for (Py_ssize_t c = 0; c < C; ++c) {
            PyObject* arr;
            if (c == 0) {
                PyObject* pce = OWN(PyObject_GetAttr(prior, s_prior_cluster_effect));
                if (!pce) goto out;
                arr = OWN(PyObject_GetAttr(pce, s_concentration_array));
            } else {
                PyObject* cp = OWN(PyObject_GetItem(conf_priors, PyList_GET_ITEM(conf_names, c - 1)));
                if (!cp) goto out;
                arr = NULL;
                if (type_in(g_trust_conf_priors, cp)) {               /* a static prior's table is an instance attribute (prior.py:325-354) */
                    PyObject* dyn = inst_get(cp, s_any_dynamic_priors_attr);
                    if (dyn == Py_False) arr = inst_get(cp, s__concentration_array);
                    Py_XDECREF(dyn);
                }
                if (!arr) arr = PyObject_CallMethodObjArgs(cp, s_concentration_array, sample, NULL);
                OWN(arr);
            }
            if (!arr) goto out;
            PyObject* ci = PyLong_FromSsize_t(c);
            if (!ci) goto out;
            PyObject* cached = PyObject_GetItem(bound_conc, ci);                /* eng._bound_conc.get(c) */
            Py_DECREF(ci);
            if (!cached) { if (!PyErr_ExceptionMatches(PyExc_KeyError)) goto out; PyErr_Clear(); cached = Py_None; Py_INCREF(cached); }
            OWN(cached);
            const int same = token_of(arr, cached, &tokens[n_tok]);
            if (same < 0) goto out;
            OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        }
        for (Py_ssize_t c = 0; c < C; ++c) {
            PyObject* node = OWN(PyObject_GetItem(feature_counts, c == 0 ? s_clusters : PyList_GET_ITEM(conf_names, c - 1)));
            if (!node) goto out;
            const int same = token_of(node, PyList_GET_ITEM(old_counts, c), &tokens[n_tok]);
            if (same < 0) goto out;
            OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        }
    }
    const int at = have_model ? 3 * (int)C : (int)C;
    {
        PyObject* w = OWN(sample_attr(sample, s__weights, s_weights));
        if (!w) goto out;
        PyObject* ow = PyDict_GetItem(old, k_weights);
        int same = token_of(w, ow ? ow : Py_None, &tokens[n_tok]);
        if (same < 0) goto out;
        OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        if (with_source) {
            PyObject* src = OWN(sample_attr(sample, s__source, s_source));
            if (!src) goto out;
            PyObject* os = PyDict_GetItem(old, k_source);
            same = token_of(src, os ? os : Py_None, &tokens[n_tok]);
            if (same < 0) goto out;
            OWN(tokens[n_tok]); if (!same) changed |= 1ull << n_tok; ++n_tok;
        }
    }
    if (!changed) {
        if (old_is_new && PyDict_SetItem(cache, slot, old) != 0) goto out;      /* (a first bind of an empty state: keep the entry) */
        result = old_stale; Py_INCREF(result);
        goto out;
    }
    /* ---- send the differences ---- */
    in_send = 1;
    {
        PyObject* mirror_map = PyObject_GetAttr(eng, s__mirror);
        const int has_mirror = mirror_map != NULL;
        if (!has_mirror) PyErr_Clear(); else OWN(mirror_map);
        PyObject* mirrors = NULL;
        if (has_mirror) {
            mirrors = PyDict_GetItemWithError(mirror_map, slot);                /* borrowed */
            if (!mirrors && PyErr_Occurred()) goto out;
            if (mirrors) { Py_INCREF(mirrors); OWN(mirrors); }
        }
        if (!mirrors) {
            mirrors = OWN(PyDict_New());
            PyObject* mc = mirrors ? OWN(PyList_New(C)) : NULL;
            if (!mc) goto out;
            for (Py_ssize_t c = 0; c < C; ++c) { Py_INCREF(Py_None); PyList_SET_ITEM(mc, c, Py_None); }
            if (PyDict_SetItem(mirrors, k_counts, mc) || PyDict_SetItem(mirrors, k_source, Py_None)) goto out;
        }
        PyObject* mirror_counts = PyDict_GetItem(mirrors, k_counts);            /* borrowed list */
        if (!mirror_counts || !PyList_Check(mirror_counts) || PyList_GET_SIZE(mirror_counts) != C) { PyErr_SetString(PyExc_TypeError, "malformed mirror entry"); goto out; }
        PyObject* nw = OWN(PyDict_New());
        PyObject* new_groups = nw ? OWN(PySequence_List(old_groups)) : NULL;
        PyObject* new_counts = new_groups ? OWN(PySequence_List(old_counts)) : NULL;
        PyObject* new_stale = new_counts ? OWN(PySet_New(old_stale)) : NULL;
        if (!new_stale) goto out;
        {
            PyObject* ow = PyDict_GetItem(old, k_weights); PyObject* os = PyDict_GetItem(old, k_source); PyObject* ol = PyDict_GetItem(old, k_lh_all);
            if (PyDict_SetItem(nw, k_groups, new_groups) || PyDict_SetItem(nw, k_counts, new_counts) || PyDict_SetItem(nw, k_weights, ow ? ow : Py_None) ||
                PyDict_SetItem(nw, k_source, os ? os : Py_None) || PyDict_SetItem(nw, k_stale, new_stale) || PyDict_SetItem(nw, k_lh_all, ol ? ol : Py_None)) goto out;
        }
        PyObject* pend_idx = OWN(PyList_New(0));
        PyObject* pend_rows = pend_idx ? OWN(PyList_New(0)) : NULL;
        PyObject* pending = pend_rows ? OWN(PyTuple_Pack(2, pend_idx, pend_rows)) : NULL;
        if (!pending) goto out;
        /* concentrations (drops every slot's entry: all tables depend on it) */
        int any_conc = 0;
        if (have_model) {
            for (Py_ssize_t c = 0; c < C; ++c) {
                if (!((changed >> (C + c)) & 1)) continue;
                any_conc = 1;
                PyObject* ci = OWN(PyLong_FromSsize_t(c));
                PyObject* r = ci ? PyObject_CallMethodObjArgs(eng, s_set_concentration, ci, PyTuple_GET_ITEM(tokens[C + c], 0), NULL) : NULL;
                if (!r) goto out;
                Py_DECREF(r);
            }
        }
        if (any_conc) {
            if (PySet_Clear(new_stale) != 0) goto out;
            for (Py_ssize_t c = 0; c < C; ++c) {
                PyObject* ci = PyLong_FromSsize_t(c);
                if (!ci || PySet_Add(new_stale, ci) != 0) { Py_XDECREF(ci); goto out; }
                Py_DECREF(ci);
            }
            if (PyDict_SetItem(nw, k_lh_all, Py_None) != 0) goto out;
        }
        /* group matrices */
        int n_groups_changed = 0;
        for (Py_ssize_t c = 0; c < C; ++c) n_groups_changed += (int)((changed >> c) & 1);
        const int has_delta = PyObject_HasAttr(eng, s_set_slot_delta);
        PyObject *groups_c = NULL, *groups_arr = NULL;                         /* groups_op = (c, array) */
        for (Py_ssize_t c = 0; c < C; ++c) {
            if (!((changed >> c) & 1)) continue;
            PyObject* ci = OWN(PyLong_FromSsize_t(c));
            if (!ci) goto out;
            if (n_groups_changed == 1 && has_delta) { groups_c = ci; groups_arr = PyTuple_GET_ITEM(tokens[c], 0); }
            else {
                PyObject* r = PyObject_CallMethodObjArgs(eng, s_set_groups, slot, ci, PyTuple_GET_ITEM(tokens[c], 0), NULL);
                if (!r) goto out;
                Py_DECREF(r);
            }
            PyObject* rem = remember_token(tokens[c]);
            if (!rem || PyList_SetItem(new_groups, c, rem) != 0) goto out;      /* (SetItem steals rem) */
        }
        PyObject* was_stale = OWN(PySet_New(new_stale));
        if (!was_stale) goto out;
        /* counts */
        int by_rows[BIND_MAXC], n_by_rows = 0;
        /* Native form of the loop below (binding._send_counts per component, then np.concatenate): when EVERY component whose
           count parameter changed is a plain C-contiguous float32 [G, F, S] array with a mirror of the same shape, the rows that
           differ are found (and copied into the mirrors) here and handed on as ONE pair (int32 global group indices, float32
           rows) -- the same two arrays the Python helpers build.  Any other form: the loop below, unchanged. */
        int counts_native = 0;
        if (have_model && g_np_empty && PyObject_HasAttr(eng, s_set_counts_rows)) {
            int32_t* diff[BIND_MAXC]; Py_ssize_t kdiff[BIND_MAXC]; int comp[BIND_MAXC]; int n_comp = 0;
            Py_ssize_t K = 0, FS = -1, Fd = -1, Sd = -1;
            int eligible = 1, any = 0;
            for (Py_ssize_t c = 0; c < C && eligible; ++c) {
                if (!((changed >> (2 * C + c)) & 1)) continue;
                any = 1;
                PyObject* arr = PyTuple_GET_ITEM(tokens[2 * C + c], 0);
                PyObject* mir = PyList_GET_ITEM(mirror_counts, c);
                if ((PyObject*)Py_TYPE(arr) != g_ndarray || mir == Py_None) { eligible = 0; break; }
                Py_buffer a, m;
                if (PyObject_GetBuffer(arr, &a, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); eligible = 0; break; }
                if (PyObject_GetBuffer(mir, &m, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT | PyBUF_WRITABLE) != 0) { PyErr_Clear(); PyBuffer_Release(&a); eligible = 0; break; }
                const char fa = a.format ? a.format[strlen(a.format) ? strlen(a.format) - 1 : 0] : 0, fm = m.format ? m.format[strlen(m.format) ? strlen(m.format) - 1 : 0] : 0;
                if (!(a.ndim == 3 && m.ndim == 3 && a.itemsize == 4 && m.itemsize == 4 && fa == 'f' && fm == 'f' && a.shape[0] == m.shape[0] && a.shape[1] == m.shape[1] &&
                      a.shape[2] == m.shape[2] && a.shape[0] > 0 && a.shape[1] * a.shape[2] > 0 && (FS < 0 || (a.shape[1] == Fd && a.shape[2] == Sd)))) eligible = 0;
                if (eligible) { Fd = a.shape[1]; Sd = a.shape[2]; FS = Fd * Sd; }
                PyBuffer_Release(&a); PyBuffer_Release(&m);
            }
            if (eligible && any) {
                counts_native = 1;
                PyObject* offs = OWN(PyObject_GetAttr(eng, s_group_offsets));
                Py_buffer vo;
                if (!offs || PyObject_GetBuffer(offs, &vo, PyBUF_C_CONTIGUOUS) != 0) goto out;
                int fail = !(vo.ndim == 1 && vo.shape[0] >= C + 1 && (vo.itemsize == 8 || vo.itemsize == 4));
                for (Py_ssize_t c = 0; c < C && !fail; ++c) {
                    if (!((changed >> (2 * C + c)) & 1)) continue;
                    PyObject* arr = PyTuple_GET_ITEM(tokens[2 * C + c], 0);
                    PyObject* mir = PyList_GET_ITEM(mirror_counts, c);
                    Py_buffer a, m;
                    if (PyObject_GetBuffer(arr, &a, PyBUF_C_CONTIGUOUS) != 0) { fail = 1; break; }
                    if (PyObject_GetBuffer(mir, &m, PyBUF_C_CONTIGUOUS | PyBUF_WRITABLE) != 0) { PyBuffer_Release(&a); fail = 1; break; }
                    int32_t* d = (int32_t*)malloc((size_t)a.shape[0] * sizeof(int32_t));
                    const int64_t k = d ? sbeh_diff_rows(a.buf, m.buf, (int64_t)a.shape[0], (int64_t)(a.len / a.shape[0]), d) : -1;
                    PyBuffer_Release(&a); PyBuffer_Release(&m);
                    if (k < 0) { free(d); if (!PyErr_Occurred()) PyErr_NoMemory(); fail = 1; break; }
                    PyObject* rem = remember_token(tokens[2 * C + c]);
                    if (!rem || PyList_SetItem(new_counts, c, rem) != 0) { free(d); fail = 1; break; }
                    if (k > 0) {
                        diff[n_comp] = d; kdiff[n_comp] = (Py_ssize_t)k; comp[n_comp++] = (int)c; K += (Py_ssize_t)k;
                        by_rows[n_by_rows++] = (int)c;
                        if (PyDict_SetItem(nw, k_lh_all, Py_None) != 0) { fail = 1; break; }
                    } else free(d);
                }
                if (!fail && K > 0) {
                    PyObject* idx = OWN(np_empty2(K, -1, g_dt_i32));
                    PyObject* shape = idx ? OWN(Py_BuildValue("(nnn)", K, Fd, Sd)) : NULL;
                    PyObject* dt_f32 = shape ? OWN(PyObject_GetAttrString(PyTuple_GET_ITEM(tokens[2 * C + comp[0]], 0), "dtype")) : NULL;
                    PyObject* rr = dt_f32 ? OWN(PyObject_CallFunctionObjArgs(g_np_empty, shape, dt_f32, NULL)) : NULL;
                    Py_buffer vi, vr;
                    if (!rr || PyObject_GetBuffer(idx, &vi, PyBUF_C_CONTIGUOUS | PyBUF_WRITABLE) != 0) fail = 1;
                    else if (PyObject_GetBuffer(rr, &vr, PyBUF_C_CONTIGUOUS | PyBUF_WRITABLE) != 0) { PyBuffer_Release(&vi); fail = 1; }
                    else {
                        if (vi.len != K * 4 || vr.len != K * FS * 4) { PyErr_SetString(PyExc_TypeError, "bind_slot: np.empty scratch of another size"); fail = 1; }
                        Py_ssize_t j = 0;
                        for (int i = 0; i < n_comp && !fail; ++i) {
                            long long o = 0; off_at(&vo, comp[i], &o);
                            Py_buffer a;
                            if (PyObject_GetBuffer(PyTuple_GET_ITEM(tokens[2 * C + comp[i]], 0), &a, PyBUF_C_CONTIGUOUS) != 0) { fail = 1; break; }
                            for (Py_ssize_t q = 0; q < kdiff[i]; ++q, ++j) {
                                ((int32_t*)vi.buf)[j] = (int32_t)(o + diff[i][q]);
                                memcpy((char*)vr.buf + j * FS * 4, (const char*)a.buf + (Py_ssize_t)diff[i][q] * FS * 4, (size_t)FS * 4);
                            }
                            PyBuffer_Release(&a);
                        }
                        PyBuffer_Release(&vi); PyBuffer_Release(&vr);
                    }
                    if (!fail && (PyList_Append(pend_idx, idx) != 0 || PyList_Append(pend_rows, rr) != 0)) fail = 1;
                }
                for (int i = 0; i < n_comp; ++i) free(diff[i]);
                PyBuffer_Release(&vo);
                if (fail) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "bind_slot: native count rows"); goto out; }
            }
        }
        if (have_model && !counts_native) {
            for (Py_ssize_t c = 0; c < C; ++c) {
                if (!((changed >> (2 * C + c)) & 1)) continue;
                const Py_ssize_t n_pending = PyList_GET_SIZE(pend_idx);
                PyObject* ci = OWN(PyLong_FromSsize_t(c));
                if (!ci) goto out;
                PyObject* r = OWN(PyObject_CallFunctionObjArgs(g_send_counts, eng, slot, ci, PyTuple_GET_ITEM(tokens[2 * C + c], 0),
                                                               PyList_GET_ITEM(mirror_counts, c), pending, NULL));
                if (!r) goto out;
                if (!PyTuple_Check(r) || PyTuple_GET_SIZE(r) != 2) { PyErr_SetString(PyExc_TypeError, "_send_counts must return a pair"); goto out; }
                PyObject* m = PyTuple_GET_ITEM(r, 0); Py_INCREF(m);
                if (PyList_SetItem(mirror_counts, c, m) != 0) goto out;
                const int ch = PyObject_IsTrue(PyTuple_GET_ITEM(r, 1));
                if (ch < 0) goto out;
                PyObject* rem = remember_token(tokens[2 * C + c]);
                if (!rem || PyList_SetItem(new_counts, c, rem) != 0) goto out;
                if (ch) {
                    if (PyList_GET_SIZE(pend_idx) > n_pending) by_rows[n_by_rows++] = (int)c;
                    else if (PySet_Add(new_stale, ci) != 0) goto out;              /* (the component went up whole) */
                    if (PyDict_SetItem(nw, k_lh_all, Py_None) != 0) goto out;
                }
            }
        }
        PyObject *rows_idx = NULL, *rows_rows = NULL; int rows_fused = 0;
        if (PyList_GET_SIZE(pend_idx) > 0) {
            rows_fused = g_rows_with_probs;
            for (int i = 0; i < n_by_rows && rows_fused; ++i) {
                PyObject* ci = PyLong_FromLong(by_rows[i]);
                if (!ci) goto out;
                const int in = PySet_Contains(was_stale, ci);
                Py_DECREF(ci);
                if (in < 0) goto out;
                if (in) rows_fused = 0;
            }
            if (PyList_GET_SIZE(pend_idx) == 1) {
                rows_idx = PyList_GET_ITEM(pend_idx, 0); rows_rows = PyList_GET_ITEM(pend_rows, 0);
            } else {
                rows_idx = OWN(PyObject_CallFunctionObjArgs(g_concatenate, pend_idx, NULL));
                rows_rows = rows_idx ? OWN(PyObject_CallFunctionObjArgs(g_concatenate, pend_rows, NULL)) : NULL;
                if (!rows_rows) goto out;
            }
            if (!rows_fused) {
                for (int i = 0; i < n_by_rows; ++i) {
                    PyObject* ci = PyLong_FromLong(by_rows[i]);
                    if (!ci || PySet_Add(new_stale, ci) != 0) { Py_XDECREF(ci); goto out; }
                    Py_DECREF(ci);
                }
            }
        }
        /* source */
        PyObject* source_op = NULL;                                             /* (rows, new[rows]) or None */
        if (with_source && ((changed >> (at + 1)) & 1)) {
            PyObject* os = PyDict_GetItem(old, k_source);                       /* what the mirror's content is known to equal, or None */
            PyObject* r = OWN(PyObject_CallFunctionObjArgs(g_send_source, eng, slot, tokens[at + 1], mirrors, os ? os : Py_None, NULL));
            if (!r) goto out;
            if (!PyTuple_Check(r) || PyTuple_GET_SIZE(r) != 2) { PyErr_SetString(PyExc_TypeError, "_send_source must return a pair"); goto out; }
            if (PyDict_SetItem(mirrors, k_source, PyTuple_GET_ITEM(r, 0)) != 0) goto out;
            if (PyTuple_GET_ITEM(r, 1) != Py_None) source_op = PyTuple_GET_ITEM(r, 1);
            PyObject* rem = OWN(remember_token(tokens[at + 1]));
            if (!rem || PyDict_SetItem(nw, k_source, rem) != 0) goto out;
        }
        const int n_ops = (groups_c != NULL) + (rows_idx != NULL) + (source_op != NULL);
        if (n_ops >= 2) {
            PyObject* kw = OWN(PyDict_New());
            PyObject* zero = kw ? OWN(PyLong_FromLong(0)) : NULL;
            PyObject* meth = zero ? OWN(PyObject_GetAttr(eng, s_set_slot_delta)) : NULL;
            PyObject* pos = meth ? OWN(PyTuple_Pack(1, slot)) : NULL;
            if (!pos) goto out;
            if (PyDict_SetItem(kw, k_groups_component, groups_c ? groups_c : zero) || PyDict_SetItem(kw, k_groups, groups_arr ? groups_arr : Py_None) ||
                PyDict_SetItem(kw, k_count_idx, rows_idx ? rows_idx : Py_None) || PyDict_SetItem(kw, k_count_rows, rows_rows ? rows_rows : Py_None) ||
                PyDict_SetItem(kw, k_update_probs, (rows_idx && rows_fused) ? Py_True : Py_False) ||
                PyDict_SetItem(kw, k_source_objects, source_op ? PyTuple_GET_ITEM(source_op, 0) : Py_None) ||
                PyDict_SetItem(kw, k_source_rows, source_op ? PyTuple_GET_ITEM(source_op, 1) : Py_None)) goto out;
            PyObject* r = PyObject_Call(meth, pos, kw);
            if (!r) goto out;
            Py_DECREF(r);
        } else {
            if (groups_c) {
                PyObject* r = PyObject_CallMethodObjArgs(eng, s_set_groups, slot, groups_c, groups_arr, NULL);
                if (!r) goto out;
                Py_DECREF(r);
            }
            if (rows_idx) {
                PyObject* r;
                if (rows_fused) {
                    PyObject* meth = OWN(PyObject_GetAttr(eng, s_set_counts_rows));
                    PyObject* pos = meth ? OWN(PyTuple_Pack(3, slot, rows_idx, rows_rows)) : NULL;
                    PyObject* kw = pos ? OWN(PyDict_New()) : NULL;
                    if (!kw || PyDict_SetItem(kw, k_update_probs, Py_True) != 0) goto out;
                    r = PyObject_Call(meth, pos, kw);
                } else r = PyObject_CallMethodObjArgs(eng, s_set_counts_rows, slot, rows_idx, rows_rows, NULL);
                if (!r) goto out;
                Py_DECREF(r);
            }
            if (source_op) {
                PyObject* r = PyObject_CallMethodObjArgs(eng, s_set_source_rows, slot, PyTuple_GET_ITEM(source_op, 0), PyTuple_GET_ITEM(source_op, 1), NULL);
                if (!r) goto out;
                Py_DECREF(r);
            }
        }
        if ((changed >> at) & 1) {
            PyObject* r = PyObject_CallMethodObjArgs(eng, s_set_weights, slot, PyTuple_GET_ITEM(tokens[at], 0), NULL);
            if (!r) goto out;
            Py_DECREF(r);
            PyObject* rem = OWN(remember_token(tokens[at]));
            if (!rem || PyDict_SetItem(nw, k_weights, rem) != 0) goto out;
        }
        if (any_conc) {
            for (Py_ssize_t c = 0; c < C; ++c) {
                if (!((changed >> (C + c)) & 1)) continue;
                PyObject* ci = OWN(PyLong_FromSsize_t(c));
                PyObject* rem = ci ? OWN(remember_token(tokens[C + c])) : NULL;
                if (!rem || PyObject_SetItem(bound_conc, ci, rem) != 0) goto out;
            }
        }
        if (PyDict_SetItem(cache, slot, nw) != 0) goto out;                     /* (the setters above dropped the slot's entry) */
        if (has_mirror && PyDict_SetItem(mirror_map, slot, mirrors) != 0) goto out;
        result = new_stale; Py_INCREF(result);
    }
out:
    if (!result && in_send) {
        /* diff_rows copies the differing rows INTO the host mirror before anything is sent: a setter that raises in between
           would leave mirror and entry claiming rows the device never received -- forget the slot */
        PyObject *et, *ev, *tb;
        PyErr_Fetch(&et, &ev, &tb);
        PyObject* r = PyObject_CallMethodObjArgs(eng, s__touch, slot, NULL);
        if (!r) PyErr_Clear(); else Py_DECREF(r);
        PyErr_Restore(et, ev, tb);
    }
    for (int i = 0; i < n_own; ++i) Py_XDECREF(own[i]);
    return result;
#undef OWN
}

/* ---- binding.counts_followed in native code ---------------------------------------------------------------------------------------
 * counts_followed(eng, entry, mirrors, nodes, off, touched, bounds, probs_rebuilt, source_rows | None, source_value | None, slot)
 * After a call that left the slot holding the sample's new counts (Engine.counts_delta(follow_slot=...), the Gibbs proposals'
 * `follow` forms) and the host's own add_changes: the touched components' mirror rows and tokens are brought to the sample's
 * counts, components whose probability rows were not rebuilt join the stale set, the source mirror takes the listed rows, and the
 * entry / mirrors go back into the engine's cache (the engine call dropped them). */
static PyObject* py_counts_followed(PyObject* self, PyObject* args) {
    PyObject *eng, *entry, *mirrors, *nodes, *off, *touched, *bounds, *source_rows, *source_value, *slot;
    int probs_rebuilt;
    if (!PyArg_ParseTuple(args, "OOOOOOOpOOO", &eng, &entry, &mirrors, &nodes, &off, &touched, &bounds, &probs_rebuilt, &source_rows, &source_value, &slot)) return NULL;
    if (!PyDict_Check(entry) || !PyDict_Check(mirrors) || !PyList_Check(nodes) || !PyList_Check(bounds)) { PyErr_SetString(PyExc_TypeError, "counts_followed: entry / mirrors / nodes / bounds"); return NULL; }
    const Py_ssize_t C = PyList_GET_SIZE(nodes);
    PyObject* mcounts = PyDict_GetItem(mirrors, k_counts);
    PyObject* ecounts = PyDict_GetItem(entry, k_counts);
    PyObject* stale = PyDict_GetItem(entry, k_stale);
    if (!mcounts || !ecounts || !stale || !PyList_Check(mcounts) || !PyList_Check(ecounts) || PyList_GET_SIZE(mcounts) != C || PyList_GET_SIZE(ecounts) != C ||
        PyList_GET_SIZE(bounds) != C + 1) { PyErr_SetString(PyExc_TypeError, "counts_followed: malformed entry"); return NULL; }
    if (source_rows != Py_None) {                       /* mirrors["source"][rows] = sample.source.value[rows]; entry["source"] = None */
        PyObject* ms = PyDict_GetItem(mirrors, k_source);
        Py_buffer vd, vs, vi;
        if (!ms || PyObject_GetBuffer(ms, &vd, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "counts_followed: no source mirror"); return NULL; }
        if (PyObject_GetBuffer(source_value, &vs, PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&vd); return NULL; }
        if (PyObject_GetBuffer(source_rows, &vi, PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&vd); PyBuffer_Release(&vs); return NULL; }
        int ok = vd.ndim >= 1 && vd.len == vs.len && vd.itemsize == 1 && vs.itemsize == 1 && vd.shape[0] == vs.shape[0] && vd.shape[0] > 0 &&
                 (vi.itemsize == 4 || vi.itemsize == 8);
        if (ok) {
            const Py_ssize_t row = vd.len / vd.shape[0], n = vi.len / vi.itemsize;
            for (Py_ssize_t i = 0; i < n && ok; ++i) {
                const long long ix = vi.itemsize == 4 ? (long long)((const int*)vi.buf)[i] : ((const long long*)vi.buf)[i];
                if (ix < 0 || ix >= vd.shape[0]) { ok = 0; break; }
                const unsigned char* sp = (const unsigned char*)vs.buf + (Py_ssize_t)ix * row;
                unsigned char* dp = (unsigned char*)vd.buf + (Py_ssize_t)ix * row;
                for (Py_ssize_t e = 0; e < row; ++e) dp[e] = sp[e] != 0;          /* (np.asarray(..., dtype=bool) of a non-bool byte array) */
            }
        }
        PyBuffer_Release(&vd); PyBuffer_Release(&vs); PyBuffer_Release(&vi);
        if (!ok) { PyErr_SetString(PyExc_ValueError, "counts_followed: source mirror and source differ in form, or a row index is out of range"); return NULL; }
        if (PyDict_SetItem(entry, k_source, Py_None) != 0) return NULL;
    }
    Py_buffer vo, vt;
    if (PyObject_GetBuffer(off, &vo, PyBUF_C_CONTIGUOUS) != 0) return NULL;
    if (PyObject_GetBuffer(touched, &vt, PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&vo); return NULL; }
    int fail = !(vo.ndim == 1 && vo.shape[0] >= C + 1 && (vo.itemsize == 8 || vo.itemsize == 4) && vt.itemsize == 4);
    if (fail) PyErr_SetString(PyExc_TypeError, "counts_followed: off / touched");
    for (Py_ssize_t c = 0; c < C && !fail; ++c) {
        const Py_ssize_t lo = PyLong_AsSsize_t(PyList_GET_ITEM(bounds, c)), hi = PyLong_AsSsize_t(PyList_GET_ITEM(bounds, c + 1));
        if (hi <= lo) continue;
        if (lo < 0 || hi > vt.len / 4) { PyErr_SetString(PyExc_IndexError, "counts_followed: bounds"); fail = 1; break; }
        long long o = 0; off_at(&vo, c, &o);
        PyObject* node = PyList_GET_ITEM(nodes, c);
        PyObject* tok = NULL;
        if (token_of(node, Py_None, &tok) < 0) { fail = 1; break; }          /* (value, version) of the count parameter */
        PyObject* mirror = PyList_GET_ITEM(mcounts, c);
        Py_buffer vd, vs;
        if (PyObject_GetBuffer(mirror, &vd, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { Py_DECREF(tok); fail = 1; break; }
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(tok, 0), &vs, PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&vd); Py_DECREF(tok); fail = 1; break; }
        if (vd.ndim >= 1 && vd.len == vs.len && vd.itemsize == vs.itemsize && vd.shape[0] == vs.shape[0] && vd.shape[0] > 0) {
            const Py_ssize_t row = vd.len / vd.shape[0];
            const int* t = (const int*)vt.buf;
            for (Py_ssize_t i = lo; i < hi; ++i) {
                const long long g = (long long)t[i] - o;
                if (g < 0 || g >= vd.shape[0]) { PyErr_SetString(PyExc_IndexError, "counts_followed: group index out of range"); fail = 1; break; }
                memcpy((char*)vd.buf + g * row, (const char*)vs.buf + g * row, (size_t)row);
            }
        } else { PyErr_SetString(PyExc_ValueError, "counts_followed: mirror and counts differ in form"); fail = 1; }
        PyBuffer_Release(&vd); PyBuffer_Release(&vs);
        if (!fail) {
            PyObject* rem = remember_token(tok);
            if (!rem || PyList_SetItem(ecounts, c, rem) != 0) fail = 1;
            if (!fail && !probs_rebuilt) {
                PyObject* ci = PyLong_FromSsize_t(c);
                if (!ci || PySet_Add(stale, ci) != 0) fail = 1;
                Py_XDECREF(ci);
            }
        }
        Py_DECREF(tok);
    }
    PyBuffer_Release(&vo); PyBuffer_Release(&vt);
    if (fail) return NULL;
    if (PyDict_SetItem(entry, k_lh_all, Py_None) != 0) return NULL;
    PyObject* cache = PyObject_GetAttr(eng, s__bound);
    PyObject* mirror_map = cache ? PyObject_GetAttr(eng, s__mirror) : NULL;
    const int bad = !mirror_map || PyObject_SetItem(cache, slot, entry) != 0 || PyObject_SetItem(mirror_map, slot, mirrors) != 0;
    Py_XDECREF(cache); Py_XDECREF(mirror_map);
    if (bad) return NULL;
    Py_RETURN_NONE;
}

/* ---- the CacheNode protocol of the reference's sample cache (sbayes/sampling/state.py:215-321) in native code --------------------------
 * Every cached quantity of a step goes through it -- is_outdated(), what_changed(), edit() / set_up_to_date() -- ten is_outdated, four
 * what_changed and six set_up_to_date calls per MCMC step, 2-8 us each where they run (cold interpreter paths between 400 KB
 * NumPy operations: tools/host_residual.py).  The functions below do exactly what those methods do, attribute by attribute, for
 * nodes whose class is one of the REGISTERED ones (node_setup: the reference's CacheNode when its methods' source digests are the
 * mirrored ones -- sbayes_amd/patch.py -- and sbayes_amd.state.CacheNode); for any other node they call the node's own method. */
static PyObject *g_plain_nodes = NULL, *g_grouped = NULL, *g_empty_i64 = NULL, *g_node_version_props = NULL;
static PyObject *s_update_value;
static PyObject *s_inputs, *s_input_idx, *s_cached_version, *s_cached_group_versions, *s_copy, *s_is_outdated, *s_what_changed,
                *s_set_up_to_date, *s_ahead_of, *s_cache, *s_group_likelihoods, *s_sum, *s_any_dynamic_priors, *s_n_groups, *s_caching,
                *k_universal_counts, *k_counts_key, *k_weights_key, *k_source_key;

static PyObject* py_node_setup(PyObject* self, PyObject* args) {
    PyObject *plain, *grouped, *empty;
    if (!PyArg_ParseTuple(args, "O!O!O", &PyTuple_Type, &plain, &PyTuple_Type, &grouped, &empty)) return NULL;
    /* the `version` property objects of the registered classes (what a subclass that does not override it inherits) */
    PyObject* props = PyTuple_New(PyTuple_GET_SIZE(plain));
    if (!props) return NULL;
    for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(plain); ++i) {
        PyObject* cls = PyTuple_GET_ITEM(plain, i);
        PyObject* d = PyType_Check(cls) ? _PyType_Lookup((PyTypeObject*)cls, s_version) : NULL;
        if (!d) d = Py_None;
        Py_INCREF(d);
        PyTuple_SET_ITEM(props, i, d);
    }
    Py_XDECREF(g_plain_nodes); Py_XDECREF(g_grouped); Py_XDECREF(g_empty_i64); Py_XDECREF(g_node_version_props);
    Py_INCREF(plain); Py_INCREF(grouped); Py_INCREF(empty);
    g_plain_nodes = plain; g_grouped = grouped; g_empty_i64 = empty; g_node_version_props = props;
    Py_RETURN_NONE;
}

static int node_is_plain(PyObject* cache) {
    if (!g_plain_nodes) return 0;
    for (Py_ssize_t i = 0; i < PyTuple_GET_SIZE(g_plain_nodes); ++i)
        if ((PyObject*)Py_TYPE(cache) == PyTuple_GET_ITEM(g_plain_nodes, i)) return 1;
    return 0;
}

/* tuple(inpt.version for inpt in cache.inputs.values())  (state.py:294-297); new reference */
static PyObject* node_version(PyObject* cache) {
    PyObject* inputs = PyObject_GetAttr(cache, s_inputs);
    if (!inputs) return NULL;
    const Py_ssize_t n = PyObject_Length(inputs);
    PyObject* it = n >= 0 ? PyObject_GetIter(inputs) : NULL;
    PyObject* tup = it ? PyTuple_New(n) : NULL;
    Py_ssize_t i = 0;
    int ok = tup != NULL;
    PyObject* key;
    while (ok && (key = PyIter_Next(it)) != NULL) {
        PyObject* inpt = PyObject_GetItem(inputs, key);
        Py_DECREF(key);
        PyObject* v = NULL;
        if (inpt) {
            /* an input that is itself a cache node whose `version` is the registered classes' property (HasComponents inherits
               it): the same tuple, built here instead of by the property's generator */
            PyObject* desc = g_node_version_props ? _PyType_Lookup(Py_TYPE(inpt), s_version) : NULL;      /* borrowed */
            int nested = 0;
            if (desc) for (Py_ssize_t j = 0; j < PyTuple_GET_SIZE(g_node_version_props); ++j) if (desc == PyTuple_GET_ITEM(g_node_version_props, j)) nested = 1;
            v = nested ? node_version(inpt) : PyObject_GetAttr(inpt, s_version);
        }
        Py_XDECREF(inpt);
        if (!v || i >= n) { Py_XDECREF(v); ok = 0; break; }
        PyTuple_SET_ITEM(tup, i++, v);
    }
    if (ok && (PyErr_Occurred() || i != n)) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_RuntimeError, "cache node inputs changed size"); ok = 0; }
    Py_XDECREF(it); Py_DECREF(inputs);
    if (!ok) { Py_XDECREF(tup); return NULL; }
    return tup;
}

/* cache.is_outdated()  (state.py:232-233): 1 / 0 / -1 */
static int node_outdated(PyObject* cache) {
    if (!node_is_plain(cache)) {
        PyObject* r = PyObject_CallMethodNoArgs(cache, s_is_outdated);
        if (!r) return -1;
        const int b = PyObject_IsTrue(r);
        Py_DECREF(r);
        return b;
    }
    PyObject* ver = node_version(cache);
    PyObject* cv = ver ? PyObject_GetAttr(cache, s_cached_version) : NULL;
    const int b = cv ? PyObject_RichCompareBool(cv, ver, Py_NE) : -1;
    Py_XDECREF(ver); Py_XDECREF(cv);
    return b;
}

/* cache.ahead_of(key)  (state.py:235-237) */
static int node_ahead_of(PyObject* cache, PyObject* key) {
    if (!node_is_plain(cache)) {
        PyObject* r = PyObject_CallMethodOneArg(cache, s_ahead_of, key);
        if (!r) return -1;
        const int b = PyObject_IsTrue(r);
        Py_DECREF(r);
        return b;
    }
    int b = -1;
    PyObject* idx = PyObject_GetAttr(cache, s_input_idx);
    PyObject* i = idx ? PyObject_GetItem(idx, key) : NULL;
    PyObject* cv = i ? PyObject_GetAttr(cache, s_cached_version) : NULL;
    PyObject* cvi = cv ? PyObject_GetItem(cv, i) : NULL;
    PyObject* inputs = cvi ? PyObject_GetAttr(cache, s_inputs) : NULL;
    PyObject* inpt = inputs ? PyObject_GetItem(inputs, key) : NULL;
    PyObject* v = inpt ? PyObject_GetAttr(inpt, s_version) : NULL;
    if (v) b = PyObject_RichCompareBool(cvi, v, Py_NE);
    Py_XDECREF(idx); Py_XDECREF(i); Py_XDECREF(cv); Py_XDECREF(cvi); Py_XDECREF(inputs); Py_XDECREF(inpt); Py_XDECREF(v);
    return b;
}

/* cache.set_up_to_date()  (state.py:265-270): 0 / -1 */
static int node_commit(PyObject* cache) {
    if (!node_is_plain(cache)) {
        PyObject* r = PyObject_CallMethodNoArgs(cache, s_set_up_to_date);
        if (!r) return -1;
        Py_DECREF(r);
        return 0;
    }
    PyObject* ver = node_version(cache);
    if (!ver || PyObject_SetAttr(cache, s_cached_version, ver) != 0) { Py_XDECREF(ver); return -1; }
    Py_DECREF(ver);
    PyObject* inputs = PyObject_GetAttr(cache, s_inputs);
    PyObject* cgv = inputs ? PyObject_GetAttr(cache, s_cached_group_versions) : NULL;
    PyObject* it = cgv ? PyObject_GetIter(inputs) : NULL;
    int ok = it != NULL;
    PyObject* key;
    while (ok && (key = PyIter_Next(it)) != NULL) {
        PyObject* inpt = PyObject_GetItem(inputs, key);
        const int grouped = inpt ? PyObject_IsInstance(inpt, g_grouped) : -1;
        if (grouped < 0) ok = 0;
        else if (grouped) {
            PyObject* gv = PyObject_GetAttr(inpt, s_group_versions);
            PyObject* stamp = gv ? PyObject_CallMethodNoArgs(gv, s_copy) : NULL;
            PyObject* flags = stamp ? PyObject_GetAttr(stamp, s_flags) : NULL;
            if (!flags || PyObject_SetAttr(flags, s_writeable, Py_False) != 0 || PyObject_SetItem(cgv, key, stamp) != 0) ok = 0;
            Py_XDECREF(gv); Py_XDECREF(stamp); Py_XDECREF(flags);
        }
        Py_XDECREF(inpt); Py_DECREF(key);
    }
    if (ok && PyErr_Occurred()) ok = 0;
    Py_XDECREF(it); Py_XDECREF(cgv); Py_XDECREF(inputs);
    return ok ? 0 : -1;
}

/* cache.what_changed(key, caching)  (state.py:239-254) as a malloc'ed ascending index list: 0 ok (*idx, *n; caller frees), -1 error.
   Nodes of another class, and group-version arrays that are not two equal-length one-dimensional arrays of one 8-byte type, go
   through the node's own method (its result read through the buffer protocol). */
static int node_changed(PyObject* cache, PyObject* key, int caching, long long** idx_out, Py_ssize_t* n_out) {
    *idx_out = NULL; *n_out = 0;
    int form = 0;                                          /* 1: handled natively */
    if (node_is_plain(cache)) {
        PyObject* inputs = PyObject_GetAttr(cache, s_inputs);
        PyObject* inpt = inputs ? PyObject_GetItem(inputs, key) : NULL;
        Py_XDECREF(inputs);
        if (!inpt) return -1;
        const int grouped = PyObject_IsInstance(inpt, g_grouped);
        if (grouped < 0) { Py_DECREF(inpt); return -1; }
        if (!grouped) { Py_DECREF(inpt); PyErr_SetString(PyExc_ValueError, "Can only track what changed for GroupedParameters"); return -1; }
        if (!caching) {
            PyObject* ng = PyObject_GetAttr(inpt, s_n_groups);
            Py_DECREF(inpt);
            if (!ng) return -1;
            const Py_ssize_t n = PyLong_AsSsize_t(ng);
            Py_DECREF(ng);
            if (n < 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "n_groups"); return -1; }
            long long* idx = (long long*)malloc((size_t)(n ? n : 1) * sizeof(long long));
            if (!idx) { PyErr_NoMemory(); return -1; }
            for (Py_ssize_t i = 0; i < n; ++i) idx[i] = i;
            *idx_out = idx; *n_out = n;
            return 0;
        }
        PyObject* gv = PyObject_GetAttr(inpt, s_group_versions);
        Py_DECREF(inpt);
        PyObject* cgvs = gv ? PyObject_GetAttr(cache, s_cached_group_versions) : NULL;
        PyObject* cgv = cgvs ? PyObject_GetItem(cgvs, key) : NULL;
        Py_XDECREF(cgvs);
        if (!cgv) { Py_XDECREF(gv); return -1; }
        Py_buffer a, b;
        if (PyObject_GetBuffer(cgv, &a, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
            if (PyObject_GetBuffer(gv, &b, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
                const char fa = a.format ? a.format[strlen(a.format) ? strlen(a.format) - 1 : 0] : 0, fb = b.format ? b.format[strlen(b.format) ? strlen(b.format) - 1 : 0] : 0;
                if (a.ndim == 1 && b.ndim == 1 && a.itemsize == 8 && b.itemsize == 8 && a.shape[0] == b.shape[0] && fa == fb && (fa == 'd' || fa == 'l' || fa == 'q')) {
                    const Py_ssize_t n = a.shape[0];
                    long long* idx = (long long*)malloc((size_t)(n ? n : 1) * sizeof(long long));
                    if (!idx) { PyBuffer_Release(&a); PyBuffer_Release(&b); Py_DECREF(gv); Py_DECREF(cgv); PyErr_NoMemory(); return -1; }
                    Py_ssize_t k = 0;
                    if (fa == 'd') { const double *x = (const double*)a.buf, *y = (const double*)b.buf; for (Py_ssize_t i = 0; i < n; ++i) if (x[i] != y[i]) idx[k++] = i; }
                    else { const long long *x = (const long long*)a.buf, *y = (const long long*)b.buf; for (Py_ssize_t i = 0; i < n; ++i) if (x[i] != y[i]) idx[k++] = i; }
                    *idx_out = idx; *n_out = k;
                    form = 1;
                }
                PyBuffer_Release(&b);
            } else PyErr_Clear();
            PyBuffer_Release(&a);
        } else PyErr_Clear();
        Py_DECREF(gv); Py_DECREF(cgv);
        if (form) return 0;
    }
    /* the node's own method */
    PyObject* meth = PyObject_GetAttr(cache, s_what_changed);
    PyObject* pos = meth ? PyTuple_Pack(1, key) : NULL;
    PyObject* kw = pos ? PyDict_New() : NULL;
    PyObject* res = NULL;
    if (kw && PyDict_SetItem(kw, s_caching, caching ? Py_True : Py_False) == 0) res = PyObject_Call(meth, pos, kw);
    Py_XDECREF(meth); Py_XDECREF(pos); Py_XDECREF(kw);
    if (!res) return -1;
    Py_buffer r;
    int rc = -1;
    if (PyObject_GetBuffer(res, &r, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) == 0) {
        const char fr = r.format ? r.format[strlen(r.format) ? strlen(r.format) - 1 : 0] : 0;
        if (r.ndim == 1 && r.itemsize == 8 && (fr == 'l' || fr == 'q' || fr == 'L' || fr == 'Q')) {
            const Py_ssize_t n = r.shape[0];
            long long* idx = (long long*)malloc((size_t)(n ? n : 1) * sizeof(long long));
            if (idx) { memcpy(idx, r.buf, (size_t)n * sizeof(long long)); *idx_out = idx; *n_out = n; rc = 0; } else PyErr_NoMemory();
        } else PyErr_SetString(PyExc_TypeError, "what_changed() did not return a one-dimensional int64 array");
        PyBuffer_Release(&r);
    }
    Py_DECREF(res);
    return rc;
}

static PyObject* py_node_outdated(PyObject* self, PyObject* cache) {
    const int b = node_outdated(cache);
    if (b < 0) return NULL;
    return PyBool_FromLong(b);
}

/* cache.update_value(value)  (state.py:261-263) */
static PyObject* py_node_update_value(PyObject* self, PyObject* args) {
    PyObject *cache, *value;
    if (!PyArg_ParseTuple(args, "OO", &cache, &value)) return NULL;
    if (!node_is_plain(cache)) return PyObject_CallMethodOneArg(cache, s_update_value, value);
    if (PyObject_SetAttr(cache, s__value, value) != 0 || node_commit(cache) != 0) return NULL;
    Py_RETURN_NONE;
}

static PyObject* py_node_commit(PyObject* self, PyObject* cache) {
    if (node_commit(cache) != 0) return NULL;
    Py_RETURN_NONE;
}

/* node_changed(cache, key, caching) -> int64 ndarray: cache.what_changed(key, caching=caching) for ONE key */
static PyObject* py_node_changed(PyObject* self, PyObject* args) {
    PyObject *cache, *key;
    int caching = 1;
    if (!PyArg_ParseTuple(args, "OO|p", &cache, &key, &caching)) return NULL;
    if (!g_empty_i64) { PyErr_SetString(PyExc_RuntimeError, "node_setup() not called"); return NULL; }
    long long* idx; Py_ssize_t n;
    if (node_changed(cache, key, caching, &idx, &n) != 0) return NULL;
    PyObject* nn = PyLong_FromSsize_t(n);
    PyObject* out = nn ? PyObject_CallOneArg(g_empty_i64, nn) : NULL;
    Py_XDECREF(nn);
    if (out && n > 0) {
        Py_buffer v;
        if (PyObject_GetBuffer(out, &v, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { Py_CLEAR(out); }
        else {
            if (v.len == n * (Py_ssize_t)sizeof(long long)) memcpy(v.buf, idx, (size_t)v.len);
            else { PyErr_SetString(PyExc_TypeError, "node_setup: empty_i64(n) must return n int64 items"); Py_CLEAR(out); }
            PyBuffer_Release(&v);
        }
    }
    free(idx);
    return out;
}

/* a one-dimensional C-contiguous float64 buffer; 0 on mismatch (no exception left) */
static int get_f64_1d(PyObject* obj, Py_buffer* v, int writable) {
    if (PyObject_GetBuffer(obj, v, (writable ? PyBUF_WRITABLE : 0) | PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); return 0; }
    const char f = v->format ? v->format[strlen(v->format) ? strlen(v->format) - 1 : 0] : 0;
    if (v->ndim != 1 || v->itemsize != 8 || f != 'd') { PyBuffer_Release(v); return 0; }
    return 1;
}

/* ---- Likelihood.__call__ (sbayes/model/likelihood.py:47-101), caching path ------------------------------------------------------------
 * likelihood_call(lik, sample, names, off, all_fn) -> the log-likelihood | NotImplemented | None
 * Component after component exactly what compute_lh_clusters / compute_lh_confounder do (sbayes_amd/likelihood.py, the reference's
 * :65-101): the cached sum when the node is current; else  lh = cache.value;  changed = what_changed("counts", caching=not
 * (conf_prior.any_dynamic_priors and cache.ahead_of("universal_counts")));  lh[changed] = values;  set_up_to_date();  sum.  The
 * values of ALL components come from `all_fn(sample)` (one bind, one device call: Likelihood._all_group_logliks), asked at the
 * first component that has a changed group.  NotImplemented (nothing touched): a node of an unregistered class or a value array
 * that is not a plain float64 vector -- the Python form serves the call.  None: all_fn returned None (groups that overlap have no
 * resident form); nodes visited so far are left current or untouched, the Python form starts over and arrives at the same
 * state. */
static PyObject* py_likelihood_call(PyObject* self, PyObject* args) {
    PyObject *lik, *sample, *names, *off, *all_fn;
    if (!PyArg_ParseTuple(args, "OOO!OO", &lik, &sample, &PyList_Type, &names, &off, &all_fn)) return NULL;
    const Py_ssize_t C = PyList_GET_SIZE(names);
    if (C < 1 || C > BIND_MAXC || !g_plain_nodes) Py_RETURN_NOTIMPLEMENTED;
    PyObject* nodes[BIND_MAXC];
    PyObject* result = NULL;
    PyObject *sc = NULL, *gl = NULL, *lh_all = NULL, *total = NULL, *conf_priors = NULL;
    Py_buffer vo, vall;
    int have_off = 0, have_all = 0;
    Py_ssize_t n_nodes = 0;
    sc = PyObject_GetAttr(sample, s_cache);
    gl = sc ? PyObject_GetAttr(sc, s_group_likelihoods) : NULL;
    if (!gl) goto done;
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject* node = PyObject_GetItem(gl, PyList_GET_ITEM(names, c));
        if (!node) goto done;
        nodes[n_nodes++] = node;
        int ok = node_is_plain(node);
        if (ok) {
            PyObject* val = PyObject_GetAttr(node, s_value);
            if (!val) goto done;
            Py_buffer v;
            ok = get_f64_1d(val, &v, 1);
            if (ok) PyBuffer_Release(&v);
            Py_DECREF(val);
        }
        if (!ok) { result = Py_NotImplemented; Py_INCREF(result); goto done; }
    }
    if (PyObject_GetBuffer(off, &vo, PyBUF_C_CONTIGUOUS) != 0) goto done;
    have_off = 1;
    if (vo.ndim != 1 || vo.shape[0] < C + 1 || (vo.itemsize != 8 && vo.itemsize != 4)) { result = Py_NotImplemented; Py_INCREF(result); goto done; }
    total = PyFloat_FromDouble(0.0);
    if (!total) goto done;
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject* cache = nodes[c];
        const int outdated = node_outdated(cache);
        if (outdated < 0) goto done;
        if (outdated) {
            int caching = 1;
            if (c > 0) {
                if (!conf_priors) {
                    PyObject* prior = PyObject_GetAttr(lik, s_prior);
                    conf_priors = prior ? PyObject_GetAttr(prior, s_prior_confounding_effects) : NULL;
                    Py_XDECREF(prior);
                    if (!conf_priors) goto done;
                }
                PyObject* cp = PyObject_GetItem(conf_priors, PyList_GET_ITEM(names, c));
                PyObject* dyn = cp ? PyObject_GetAttr(cp, s_any_dynamic_priors) : NULL;
                Py_XDECREF(cp);
                if (!dyn) goto done;
                const int is_dyn = PyObject_IsTrue(dyn);
                Py_DECREF(dyn);
                if (is_dyn < 0) goto done;
                if (is_dyn) {
                    const int ahead = node_ahead_of(cache, k_universal_counts);
                    if (ahead < 0) goto done;
                    if (ahead) caching = 0;
                }
            }
            long long* idx; Py_ssize_t n;
            if (node_changed(cache, k_counts_key, caching, &idx, &n) != 0) goto done;
            if (n > 0) {
                if (!lh_all) {
                    lh_all = PyObject_CallOneArg(all_fn, sample);
                    if (!lh_all) { free(idx); goto done; }
                    if (lh_all == Py_None) { free(idx); result = Py_None; Py_INCREF(result); goto done; }
                    if (!get_f64_1d(lh_all, &vall, 0)) { free(idx); PyErr_SetString(PyExc_TypeError, "likelihood_call: all_fn must return a float64 vector"); goto done; }
                    have_all = 1;
                }
                PyObject* val = PyObject_GetAttr(cache, s_value);
                Py_buffer v;
                if (!val || !get_f64_1d(val, &v, 1)) { Py_XDECREF(val); free(idx); if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "likelihood_call: the node's value changed form"); goto done; }
                long long o = 0; off_at(&vo, c, &o);
                int bad = 0;
                for (Py_ssize_t i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= v.shape[0] || o + idx[i] < 0 || o + idx[i] >= vall.shape[0]) bad = 1;
                if (!bad) for (Py_ssize_t i = 0; i < n; ++i) ((double*)v.buf)[idx[i]] = ((const double*)vall.buf)[o + idx[i]];
                PyBuffer_Release(&v); Py_DECREF(val);
                if (bad) { free(idx); PyErr_SetString(PyExc_IndexError, "likelihood_call: group index out of range"); goto done; }
            }
            free(idx);
            if (node_commit(cache) != 0) goto done;
        }
        PyObject* val = PyObject_GetAttr(cache, s_value);
        PyObject* s = val ? PyObject_CallMethodNoArgs(val, s_sum) : NULL;
        Py_XDECREF(val);
        if (!s) goto done;
        PyObject* t2 = PyNumber_InPlaceAdd(total, s);
        Py_DECREF(s);
        if (!t2) goto done;
        Py_DECREF(total);
        total = t2;
    }
    result = total; total = NULL;
done:
    for (Py_ssize_t i = 0; i < n_nodes; ++i) Py_DECREF(nodes[i]);
    if (have_off) PyBuffer_Release(&vo);
    if (have_all) PyBuffer_Release(&vall);
    Py_XDECREF(sc); Py_XDECREF(gl); Py_XDECREF(lh_all); Py_XDECREF(total); Py_XDECREF(conf_priors);
    return result;
}

/* ---- SourcePrior.__call__'s cache update (sbayes/model/prior.py:596-609) -----------------------------------------------------------------
 * store_per_object(cache, n_objects, values, caching) -> None | NotImplemented
 *   per_object = cache.value
 *   if cache.ahead_of("weights"): per_object[:] = values           (the reference's changed = arange(n_objects))
 *   else: changed = cache.what_changed("source", caching); per_object[changed] = values[changed]
 *   cache.set_up_to_date()
 * `values`: a float64 vector [n_objects], or a callable returning one -- called only when something is listed, like the
 * reference computes its values only then.  NotImplemented (nothing touched): an unregistered node class or a value that is not a
 * plain float64 vector. */
static PyObject* py_store_per_object(PyObject* self, PyObject* args) {
    PyObject *cache, *values;
    Py_ssize_t n_objects;
    int caching = 1;
    if (!PyArg_ParseTuple(args, "OnO|p", &cache, &n_objects, &values, &caching)) return NULL;
    if (!node_is_plain(cache)) Py_RETURN_NOTIMPLEMENTED;
    PyObject* val = PyObject_GetAttr(cache, s_value);
    if (!val) return NULL;
    Py_buffer v;
    if (!get_f64_1d(val, &v, 1)) { Py_DECREF(val); Py_RETURN_NOTIMPLEMENTED; }
    PyObject* result = NULL;
    PyObject* got = NULL;
    long long* idx = NULL; Py_ssize_t n = 0;
    int all = node_ahead_of(cache, k_weights_key);
    if (all < 0) goto done;
    if (all) n = n_objects;
    else if (node_changed(cache, k_source_key, caching, &idx, &n) != 0) goto done;
    if (n > 0) {
        if (PyCallable_Check(values)) { got = PyObject_CallNoArgs(values); if (!got) goto done; }
        else { got = values; Py_INCREF(got); }
        Py_buffer w;
        if (!get_f64_1d(got, &w, 0)) { PyErr_SetString(PyExc_TypeError, "store_per_object: values must be a float64 vector"); goto done; }
        int bad = 0;
        if (all) {
            if (w.shape[0] != v.shape[0] || n_objects != v.shape[0]) bad = 1;
            else memcpy(v.buf, w.buf, (size_t)v.len);
        } else {
            for (Py_ssize_t i = 0; i < n; ++i) if (idx[i] < 0 || idx[i] >= v.shape[0] || idx[i] >= w.shape[0]) bad = 1;
            if (!bad) for (Py_ssize_t i = 0; i < n; ++i) ((double*)v.buf)[idx[i]] = ((const double*)w.buf)[idx[i]];
        }
        PyBuffer_Release(&w);
        if (bad) { PyErr_SetString(PyExc_IndexError, "store_per_object: values and the cached vector differ in length"); goto done; }
    }
    if (node_commit(cache) != 0) goto done;
    result = Py_None; Py_INCREF(result);
done:
    free(idx);
    PyBuffer_Release(&v); Py_DECREF(val); Py_XDECREF(got);
    return result;
}

/* ---- counts.update_feature_counts (sbayes/sampling/counts.py:55-95) in native code ------------------------------------------------------
 * update_counts(sample_old, sample_new, features, object_subset, follow_ok) -> sample_new.feature_counts | NotImplemented
 * A transcription of the Python function (sbayes_amd/counts.py, kept as the reference form and the fallback): the same ids
 * (sbeh_subset_ids), the same ONE engine call -- counts_delta(objs, gid_old, gid_new, sid_old, sid_new[, follow_slot=0,
 * update_probs=..., update_source=...]) -- the same add_changes per component (add_rows_many), the same bind-cache follow-up
 * (counts_followed, the source lineage notes).  NotImplemented (nothing touched, the Python form serves the call): a sample whose
 * arrays are not plain C-contiguous bool arrays, an object_subset that is neither a bool mask [N] nor a one-dimensional int32
 * array, an object listed twice or in several groups of one component (the reference's two-count difference applies). */
static PyObject* py_update_counts(PyObject* self, PyObject* args) {
    PyObject *s_old, *s_new, *features, *subset;
    int follow_ok = 0;
    if (!PyArg_ParseTuple(args, "OOOOp", &s_old, &s_new, &features, &subset, &follow_ok)) return NULL;
    if (!g_np_empty || !g_ndarray || !g_send_counts) { PyErr_SetString(PyExc_RuntimeError, "update_counts_setup() / bind_setup() / scan_setup() not called"); return NULL; }
    PyObject* own[96 + 12 * BIND_MAXC]; int n_own = 0;
#define OWN(x) (own[n_own++] = (x))
    Py_buffer vb[8 + 2 * BIND_MAXC]; int nb = 0;
    PyObject* result = NULL;
    int followed_started = 0;
    PyObject* eng = NULL;
    PyObject* counts = OWN(sample_attr(s_new, s__feature_counts, s_feature_counts));
    PyObject* conf_new = counts ? OWN(PyObject_GetAttr(s_new, s_confounders)) : NULL;
    PyObject* conf_old = conf_new ? OWN(PyObject_GetAttr(s_old, s_confounders)) : NULL;
    PyObject* conf_names = conf_old ? OWN(PySequence_List(conf_new)) : NULL;
    if (!conf_names) goto out;
    const Py_ssize_t C = 1 + PyList_GET_SIZE(conf_names);
    if (C > BIND_MAXC) goto unsupported;
    PyObject* names = OWN(PyList_New(C));
    if (!names) goto out;
    Py_INCREF(s_clusters); PyList_SET_ITEM(names, 0, s_clusters);
    for (Py_ssize_t c = 1; c < C; ++c) { PyObject* k = PyList_GET_ITEM(conf_names, c - 1); Py_INCREF(k); PyList_SET_ITEM(names, c, k); }
    /* the group matrices of both samples: plain C-contiguous bool [G_c, N] */
    const uint8_t* pn[BIND_MAXC]; const uint8_t* po[BIND_MAXC]; int32_t ng[BIND_MAXC];
    Py_ssize_t N = -1;
    PyObject* n_groups = OWN(PyList_New(C));
    if (!n_groups) goto out;
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject *gn, *go;
        if (c == 0) {
            PyObject* p = OWN(sample_attr(s_new, s__clusters, s_clusters));
            gn = p ? OWN(param_value(p)) : NULL;
            PyObject* q = gn ? OWN(sample_attr(s_old, s__clusters, s_clusters)) : NULL;
            go = q ? OWN(param_value(q)) : NULL;
        } else {
            PyObject* k = PyList_GET_ITEM(conf_names, c - 1);
            PyObject* p = OWN(PyObject_GetItem(conf_new, k));
            gn = p ? OWN(PyObject_GetAttr(p, s_group_assignment)) : NULL;
            PyObject* q = gn ? OWN(PyObject_GetItem(conf_old, k)) : NULL;
            go = q ? OWN(PyObject_GetAttr(q, s_group_assignment)) : NULL;
        }
        if (!go) goto out;
        if ((PyObject*)Py_TYPE(gn) != g_ndarray || (PyObject*)Py_TYPE(go) != g_ndarray) goto unsupported;
        if (!get_c(gn, &vb[nb], 2, 1, 0)) goto unsupported;
        Py_buffer* va = &vb[nb++];
        if (N < 0) N = va->shape[1];
        if (va->shape[1] != N) goto unsupported;
        pn[c] = (const uint8_t*)va->buf; ng[c] = (int32_t)va->shape[0];
        if (go == gn) po[c] = pn[c];
        else {
            if (!get_c(go, &vb[nb], 2, 1, 0)) goto unsupported;
            Py_buffer* vo = &vb[nb++];
            if (vo->shape[0] != va->shape[0] || vo->shape[1] != N) goto unsupported;
            po[c] = (const uint8_t*)vo->buf;
        }
        PyObject* g = PyLong_FromSsize_t(va->shape[0]);
        if (!g) goto out;
        PyList_SET_ITEM(n_groups, c, g);
    }
    /* source of both samples: plain C-contiguous bool [N, F, C] */
    PyObject* sp_new = OWN(sample_attr(s_new, s__source, s_source));
    PyObject* src_new = sp_new ? OWN(param_value(sp_new)) : NULL;
    PyObject* sp_old = src_new ? OWN(sample_attr(s_old, s__source, s_source)) : NULL;
    PyObject* src_old = sp_old ? OWN(param_value(sp_old)) : NULL;
    if (!src_old) goto out;
    if ((PyObject*)Py_TYPE(src_new) != g_ndarray || (PyObject*)Py_TYPE(src_old) != g_ndarray) goto unsupported;
    if (!get_c(src_new, &vb[nb], 3, 1, 0)) goto unsupported;
    Py_buffer* b_sn = &vb[nb++];
    Py_buffer* b_so = b_sn;
    if (src_old != src_new) { if (!get_c(src_old, &vb[nb], 3, 1, 0)) goto unsupported; b_so = &vb[nb++]; }
    const Py_ssize_t F = b_sn->shape[1];
    if (b_sn->shape[0] != N || b_sn->shape[2] != C || b_so->shape[0] != N || b_so->shape[1] != F || b_so->shape[2] != C) goto unsupported;
    /* the listed objects: a bool mask [N] (np.flatnonzero) or int32 indices */
    PyObject* objs = NULL;
    if ((PyObject*)Py_TYPE(subset) != g_ndarray) goto unsupported;
    {
        Py_buffer vs;
        if (PyObject_GetBuffer(subset, &vs, PyBUF_C_CONTIGUOUS | PyBUF_FORMAT) != 0) { PyErr_Clear(); goto unsupported; }
        const char f = vs.format ? vs.format[strlen(vs.format) ? strlen(vs.format) - 1 : 0] : 0;
        if (vs.ndim == 1 && vs.itemsize == 1 && f == '?' && vs.shape[0] == N) {
            Py_ssize_t k = 0;
            const uint8_t* m = (const uint8_t*)vs.buf;
            for (Py_ssize_t i = 0; i < N; ++i) k += m[i] != 0;
            objs = OWN(np_empty2(k, -1, g_dt_i32));
            Py_buffer vo;
            if (!objs || PyObject_GetBuffer(objs, &vo, PyBUF_WRITABLE | PyBUF_C_CONTIGUOUS) != 0) { PyBuffer_Release(&vs); goto out; }
            int32_t* o = (int32_t*)vo.buf;
            Py_ssize_t j = 0;
            if (vo.len == k * 4) for (Py_ssize_t i = 0; i < N; ++i) if (m[i]) o[j++] = (int32_t)i;
            PyBuffer_Release(&vo);
            PyBuffer_Release(&vs);
            if (j != k) { PyErr_SetString(PyExc_TypeError, "update_counts_setup: np.empty(n, int32) expected"); goto out; }
        } else if (vs.ndim == 1 && vs.itemsize == 4 && (f == 'i' || f == 'l')) {
            PyBuffer_Release(&vs);
            objs = subset;
        } else { PyBuffer_Release(&vs); goto unsupported; }
    }
    if (!get_c(objs, &vb[nb], 1, 4, 0)) goto unsupported;
    Py_buffer* b_objs = &vb[nb++];
    const Py_ssize_t n = b_objs->shape[0];
    /* ids of both samples for the listed objects */
    PyObject* gid_new = OWN(np_empty2(C, n, g_dt_i32));
    PyObject* gid_old = gid_new ? OWN(np_empty2(C, n, g_dt_i32)) : NULL;
    PyObject* sid_new = gid_old ? OWN(np_empty2(n, F, g_dt_u8)) : NULL;
    PyObject* sid_old = !sid_new ? NULL : (src_old == src_new ? sid_new : OWN(np_empty2(n, F, g_dt_u8)));
    if (!sid_old) goto out;
    {
        Py_buffer g1, g2, s1, s2;
        if (!get_c(gid_new, &g1, 2, 4, 1)) { PyErr_SetString(PyExc_TypeError, "update_counts: id scratch"); goto out; }
        if (!get_c(gid_old, &g2, 2, 4, 1)) { PyBuffer_Release(&g1); PyErr_SetString(PyExc_TypeError, "update_counts: id scratch"); goto out; }
        if (!get_c(sid_new, &s1, 2, 1, 1)) { PyBuffer_Release(&g1); PyBuffer_Release(&g2); PyErr_SetString(PyExc_TypeError, "update_counts: id scratch"); goto out; }
        int have_s2 = 0;
        if (sid_old != sid_new) { if (!get_c(sid_old, &s2, 2, 1, 1)) { PyBuffer_Release(&g1); PyBuffer_Release(&g2); PyBuffer_Release(&s1); PyErr_SetString(PyExc_TypeError, "update_counts: id scratch"); goto out; } have_s2 = 1; }
        const int rc = sbeh_subset_ids((const int32_t*)b_objs->buf, (int)n, (int64_t)N, (int)F, (int)C, ng, pn, po, (const uint8_t*)b_sn->buf, (const uint8_t*)b_so->buf,
                                       (int32_t*)g1.buf, (int32_t*)g2.buf, (uint8_t*)s1.buf, have_s2 ? (uint8_t*)s2.buf : (uint8_t*)s1.buf);
        PyBuffer_Release(&g1); PyBuffer_Release(&g2); PyBuffer_Release(&s1); if (have_s2) PyBuffer_Release(&s2);
        if (rc == 1) goto unsupported;                       /* (repeated objects / several groups: the two-count difference) */
        if (rc < 0) { PyErr_SetString(PyExc_ValueError, "object index out of range in object_subset"); goto out; }
    }
    eng = OWN(PyObject_CallFunctionObjArgs(g_get_engine, features, n_groups, NULL));
    if (!eng) goto out;
    PyObject* off = OWN(PyObject_GetAttr(eng, s_group_offsets));
    if (!off) goto out;
    PyObject* parent_tok = NULL;                              /* (array, version) of sample_old.source: binding.py, source lineage */
    if (token_of(sp_old, Py_None, &parent_tok) < 0) goto out;
    OWN(parent_tok);
    PyObject* slot0 = OWN(PyLong_FromLong(0));
    if (!slot0) goto out;
    /* a slot that holds the counts this difference is added to follows on the device (binding.counts_follow_plan) */
    PyObject *entry = NULL, *mirrors = NULL, *nodes = NULL;
    nodes = OWN(PyList_New(C));
    if (!nodes) goto out;
    for (Py_ssize_t c = 0; c < C; ++c) {
        PyObject* node = PyObject_GetItem(counts, PyList_GET_ITEM(names, c));
        if (!node) goto out;
        PyList_SET_ITEM(nodes, c, node);
    }
    if (follow_ok) {
        PyObject* bound = PyObject_GetAttr(eng, s__bound);
        if (!bound) PyErr_Clear();
        else {
            OWN(bound);
            PyObject* mm = PyObject_GetAttr(eng, s__mirror);
            if (!mm) PyErr_Clear();
            else {
                OWN(mm);
                PyObject* e = PyDict_Check(bound) ? PyDict_GetItemWithError(bound, slot0) : NULL;
                PyObject* m = (e && PyDict_Check(mm)) ? PyDict_GetItemWithError(mm, slot0) : NULL;
                if (PyErr_Occurred()) goto out;
                if (e && m && PyDict_Check(e) && PyDict_Check(m)) {
                    PyObject* ec = PyDict_GetItem(e, k_counts);
                    PyObject* mc = PyDict_GetItem(m, k_counts);
                    int all = ec && mc && PyList_Check(ec) && PyList_Check(mc) && PyList_GET_SIZE(ec) == C && PyList_GET_SIZE(mc) == C;
                    for (Py_ssize_t c = 0; c < C && all; ++c) {
                        PyObject* cached = PyList_GET_ITEM(ec, c);
                        if (cached == Py_None || PyList_GET_ITEM(mc, c) == Py_None) { all = 0; break; }
                        PyObject* tok = NULL;
                        const int same = token_of(PyList_GET_ITEM(nodes, c), cached, &tok);
                        Py_XDECREF(tok);
                        if (same < 0) goto out;
                        if (!same) all = 0;
                    }
                    if (all) { entry = e; mirrors = m; Py_INCREF(entry); Py_INCREF(mirrors); OWN(entry); OWN(mirrors); }
                }
            }
        }
    }
    PyObject *touched = NULL, *rows = NULL;
    int rebuild = 0, with_source = 0;
    PyObject* known = NULL;
    {
        PyObject* meth = OWN(PyObject_GetAttr(eng, s_counts_delta));
        PyObject* pos = meth ? OWN(PyTuple_Pack(5, objs, gid_old, gid_new, sid_old, sid_new)) : NULL;
        if (!pos) goto out;
        PyObject* r;
        if (!entry) r = PyObject_Call(meth, pos, NULL);
        else {
            PyObject* stale = PyDict_GetItem(entry, k_stale);
            const Py_ssize_t n_stale = stale ? PyObject_Length(stale) : -1;
            if (n_stale < 0) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_TypeError, "malformed bind entry"); goto out; }
            rebuild = n_stale == 0;                          /* (the probability rows are rebuilt along when no table of the slot is stale) */
            PyObject* ms = PyDict_GetItem(mirrors, k_source);
            if (ms && ms != Py_None) {                       /* the subset's new source rows are in the call anyway: a slot that has a source takes them */
                Py_buffer vm;
                if (PyObject_GetBuffer(ms, &vm, PyBUF_C_CONTIGUOUS) == 0) {
                    with_source = vm.ndim == 3 && vm.shape[0] == N && vm.shape[1] == F && vm.shape[2] == C;
                    PyBuffer_Release(&vm);
                } else PyErr_Clear();
            }
            known = PyDict_GetItem(entry, k_source);          /* borrowed; the entry is ours (OWN) */
            if (known) { Py_INCREF(known); OWN(known); }
            PyObject* kw = OWN(PyDict_New());
            if (!kw || PyDict_SetItem(kw, k_follow_slot, slot0) || PyDict_SetItem(kw, k_update_probs, rebuild ? Py_True : Py_False) ||
                PyDict_SetItem(kw, k_update_source, with_source ? Py_True : Py_False)) goto out;
            followed_started = 1;
            r = PyObject_Call(meth, pos, kw);
        }
        if (!r) goto out;
        OWN(r);
        if (!PyTuple_Check(r) || PyTuple_GET_SIZE(r) != 2) { PyErr_SetString(PyExc_TypeError, "counts_delta must return (touched, rows)"); goto out; }
        touched = PyTuple_GET_ITEM(r, 0); rows = PyTuple_GET_ITEM(r, 1);
    }
    /* where the two samples' sources differ, for the binds to come: this call's own contract (binding.py, source lineage) */
    PyObject* child_tok = NULL;
    if (!(entry && with_source)) {
        if (token_of(sp_new, Py_None, &child_tok) < 0) goto out;
        OWN(child_tok);
        PyObject* r = PyObject_CallFunctionObjArgs(g_note_lineage, parent_tok, child_tok, objs, NULL);
        if (!r) goto out;
        Py_DECREF(r);
    }
    /* the reference's add_changes(diff) per component, in its row form */
    PyObject* bounds;
    {
        PyObject* a = OWN(PyTuple_Pack(4, nodes, off, touched, rows));
        bounds = a ? py_add_rows_many(NULL, a) : NULL;
        if (!bounds) goto out;
        OWN(bounds);
        if (bounds == Py_None) {
            bounds = OWN(PyObject_CallFunctionObjArgs(g_apply_rows, counts, names, off, touched, rows, Py_True, NULL));
            if (!bounds) goto out;
        }
    }
    if (entry) {
        PyObject* a = OWN(Py_BuildValue("(OOOOOOOOOOO)", eng, entry, mirrors, nodes, off, touched, bounds, rebuild ? Py_True : Py_False,
                                        with_source ? objs : Py_None, with_source ? src_new : Py_None, slot0));
        PyObject* r = a ? py_counts_followed(NULL, a) : NULL;
        if (!r) goto out;
        Py_DECREF(r);
        if (with_source) {
            r = PyObject_CallFunctionObjArgs(g_source_followed, entry, mirrors, known ? known : Py_None, parent_tok, s_new, objs, NULL);
            if (!r) goto out;
            Py_DECREF(r);
        }
    }
    followed_started = 0;
    result = counts; Py_INCREF(result);
    goto out;
unsupported:
    result = Py_NotImplemented; Py_INCREF(result);
out:
    if (!result && followed_started && eng) {                 /* (half-updated mirrors must not come back into the cache) */
        PyObject *et, *ev, *tb;
        PyErr_Fetch(&et, &ev, &tb);
        PyObject* z = PyLong_FromLong(0);
        PyObject* r = z ? PyObject_CallMethodObjArgs(eng, s__touch, z, NULL) : NULL;
        if (!r) PyErr_Clear(); else Py_DECREF(r);
        Py_XDECREF(z);
        PyErr_Restore(et, ev, tb);
    }
    for (int i = 0; i < nb; ++i) PyBuffer_Release(&vb[i]);
    for (int i = 0; i < n_own; ++i) Py_XDECREF(own[i]);
    return result;
#undef OWN
}

static PyMethodDef methods[] = {
    {"scan_setup", py_scan_setup, METH_VARARGS, "scan_setup(ndarray_type, asarray, content_equal)"},
    {"scan", py_scan, METH_VARARGS, "scan(params, cached) -> (tokens, changed bitmask): the bind cache's token comparison"},
    {"addr", py_addr, METH_O, "buffer address of an array (any strides), as int"},
    {"subset_ids", py_subset_ids, METH_VARARGS, "ids of the listed objects for sbe_counts_delta (sbeh_subset_ids)"},
    {"diff_rows", py_diff_rows, METH_VARARGS, "rows of `new` differing from `mirror`, copied into it (sbeh_diff_rows)"},
    {"diff_rows_among", py_diff_rows_among, METH_VARARGS, "the listed rows of `new` differing from `mirror`, copied into it (sbeh_diff_rows_among)"},
    {"touched_groups", py_touched_groups, METH_VARARGS, "sorted distinct group indices among two id arrays (sbeh_touched_groups)"},
    {"bind_setup", py_bind_setup, METH_VARARGS, "bind_setup(_send_counts, _send_source, _remember, concatenate, rows_with_probs, python_bind_slot)"},
    {"bind_slot", (PyCFunction)(void (*)(void))py_bind_slot, METH_VARARGS | METH_KEYWORDS, "bind_slot(eng, model, sample, slot, with_source=False) -> stale set (binding._bind_slot in C)"},
    {"counts_followed", py_counts_followed, METH_VARARGS, "binding.counts_followed in C (mirrors and tokens follow the sample after a follow-slot call)"},
    {"add_rows_many", py_add_rows_many, METH_VARARGS, "FeatureCounts.add_changes of every component for a difference given as rows -> bounds | None"},
    {"node_setup", py_node_setup, METH_VARARGS, "node_setup(plain CacheNode classes, GroupedParameters classes, empty_i64)"},
    {"node_outdated", py_node_outdated, METH_O, "cache.is_outdated()"},
    {"node_commit", py_node_commit, METH_O, "cache.set_up_to_date()"},
    {"node_update_value", py_node_update_value, METH_VARARGS, "cache.update_value(value)"},
    {"node_changed", py_node_changed, METH_VARARGS, "cache.what_changed(key, caching=True) for one key -> int64 ndarray"},
    {"likelihood_call", py_likelihood_call, METH_VARARGS, "Likelihood.__call__'s caching path: likelihood_call(lik, sample, names, off, all_fn)"},
    {"store_per_object", py_store_per_object, METH_VARARGS, "SourcePrior.__call__'s cache update: store_per_object(cache, n_objects, values, caching=True)"},
    {"update_counts_setup", py_update_counts_setup, METH_VARARGS, "update_counts_setup(np.empty, int32 dtype, uint8 dtype, get_engine, note_source_lineage, _source_followed, apply_count_rows)"},
    {"update_counts", py_update_counts, METH_VARARGS, "counts.update_feature_counts in C: update_counts(sample_old, sample_new, features, object_subset, follow_ok)"},
    {"trust_setup", py_trust_setup, METH_VARARGS, "trust_setup(sample classes, parameter classes, confounder prior classes): exact classes whose read-only properties are read from the instance"},
    {"copy_rows", py_copy_rows, METH_VARARGS, "dst[idx] = src[idx] (rows of two same-shaped C-contiguous arrays)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_sbe_pyhost", "host-layer helpers of sbayes_amd (no device code)", -1, methods};

PyMODINIT_FUNC PyInit__sbe_pyhost(void) {
    s_value = PyUnicode_InternFromString("value");
    s_version = PyUnicode_InternFromString("version");
    s_flags = PyUnicode_InternFromString("flags");
    s_writeable = PyUnicode_InternFromString("writeable");
    s_owndata = PyUnicode_InternFromString("owndata");
    s_confounders = PyUnicode_InternFromString("confounders");
    s_clusters = PyUnicode_InternFromString("clusters");
    s_group_assignment = PyUnicode_InternFromString("group_assignment");
    s_prior = PyUnicode_InternFromString("prior");
    s_prior_confounding_effects = PyUnicode_InternFromString("prior_confounding_effects");
    s_prior_cluster_effect = PyUnicode_InternFromString("prior_cluster_effect");
    s_concentration_array = PyUnicode_InternFromString("concentration_array");
    s_feature_counts = PyUnicode_InternFromString("feature_counts");
    s_weights = PyUnicode_InternFromString("weights");
    s_source = PyUnicode_InternFromString("source");
    s__bound = PyUnicode_InternFromString("_bound");
    s__bound_conc = PyUnicode_InternFromString("_bound_conc");
    s__mirror = PyUnicode_InternFromString("_mirror");
    s_set_concentration = PyUnicode_InternFromString("set_concentration");
    s_set_groups = PyUnicode_InternFromString("set_groups");
    s_set_slot_delta = PyUnicode_InternFromString("set_slot_delta");
    s_set_counts_rows = PyUnicode_InternFromString("set_counts_rows");
    s_set_source_rows = PyUnicode_InternFromString("set_source_rows");
    s_set_weights = PyUnicode_InternFromString("set_weights");
    k_groups = PyUnicode_InternFromString("groups");
    k_counts = PyUnicode_InternFromString("counts");
    k_weights = PyUnicode_InternFromString("weights");
    k_source = PyUnicode_InternFromString("source");
    k_stale = PyUnicode_InternFromString("stale");
    k_lh_all = PyUnicode_InternFromString("lh_all");
    k_update_probs = PyUnicode_InternFromString("update_probs");
    k_groups_component = PyUnicode_InternFromString("groups_component");
    k_count_idx = PyUnicode_InternFromString("count_idx");
    k_count_rows = PyUnicode_InternFromString("count_rows");
    k_source_objects = PyUnicode_InternFromString("source_objects");
    k_source_rows = PyUnicode_InternFromString("source_rows");
    s__touch = PyUnicode_InternFromString("_touch");
    s_shared = PyUnicode_InternFromString("shared");
    s_resolve_sharing = PyUnicode_InternFromString("resolve_sharing");
    s__value = PyUnicode_InternFromString("_value");
    s_group_versions = PyUnicode_InternFromString("group_versions");
    s_update_value = PyUnicode_InternFromString("update_value");
    if (!s_update_value) return NULL;
    s_inputs = PyUnicode_InternFromString("inputs");
    s_input_idx = PyUnicode_InternFromString("input_idx");
    s_cached_version = PyUnicode_InternFromString("cached_version");
    s_cached_group_versions = PyUnicode_InternFromString("cached_group_versions");
    s_copy = PyUnicode_InternFromString("copy");
    s_is_outdated = PyUnicode_InternFromString("is_outdated");
    s_what_changed = PyUnicode_InternFromString("what_changed");
    s_set_up_to_date = PyUnicode_InternFromString("set_up_to_date");
    s_ahead_of = PyUnicode_InternFromString("ahead_of");
    s_cache = PyUnicode_InternFromString("cache");
    s_group_likelihoods = PyUnicode_InternFromString("group_likelihoods");
    s_sum = PyUnicode_InternFromString("sum");
    s_any_dynamic_priors = PyUnicode_InternFromString("any_dynamic_priors");
    s_n_groups = PyUnicode_InternFromString("n_groups");
    s_caching = PyUnicode_InternFromString("caching");
    k_universal_counts = PyUnicode_InternFromString("universal_counts");
    k_counts_key = PyUnicode_InternFromString("counts");
    k_weights_key = PyUnicode_InternFromString("weights");
    k_source_key = PyUnicode_InternFromString("source");
    s_copy_method = PyUnicode_InternFromString("copy");
    if (!s_copy_method) return NULL;
    s__clusters = PyUnicode_InternFromString("_clusters");
    s__weights = PyUnicode_InternFromString("_weights");
    s__source = PyUnicode_InternFromString("_source");
    s__feature_counts = PyUnicode_InternFromString("_feature_counts");
    s__concentration_array = PyUnicode_InternFromString("_concentration_array");
    s_any_dynamic_priors_attr = PyUnicode_InternFromString("any_dynamic_priors");
    if (!s__clusters || !s__weights || !s__source || !s__feature_counts || !s__concentration_array || !s_any_dynamic_priors_attr) return NULL;
    s_counts_delta = PyUnicode_InternFromString("counts_delta");
    s_group_offsets = PyUnicode_InternFromString("group_offsets");
    k_follow_slot = PyUnicode_InternFromString("follow_slot");
    k_update_source = PyUnicode_InternFromString("update_source");
    if (!s_counts_delta || !s_group_offsets || !k_follow_slot || !k_update_source) return NULL;
    if (!s_inputs || !s_input_idx || !s_cached_version || !s_cached_group_versions || !s_copy || !s_is_outdated || !s_what_changed || !s_set_up_to_date ||
        !s_ahead_of || !s_cache || !s_group_likelihoods || !s_sum || !s_any_dynamic_priors || !s_n_groups || !s_caching || !k_universal_counts || !k_counts_key ||
        !k_weights_key || !k_source_key) return NULL;
    if (!s_value || !s_version || !s_flags || !s_writeable || !s_owndata || !s_shared || !s_resolve_sharing || !s__value || !s_group_versions) return NULL;
    return PyModule_Create(&moduledef);
}
