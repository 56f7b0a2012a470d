"""Host-layer glue in native code: sbayes_amd._sbe_pyhost (csrc/sbe_pyhost.c, built by __graft_entry__.build() / build.sh
with gcc) where it is built, the same helpers through ctypes (include/sbe_engine.h: sbe_host_*) otherwise.  Same results
either way; the extension saves the per-argument and per-call overhead of the ctypes route (tools/host_residual.py: the
host layer above the C ABI is the largest in-scope share of a patched MCMC step).  No device code is reached from here."""
from __future__ import annotations

import ctypes as ct

import numpy as np

from . import _lib

import os

try:
    if os.environ.get("SBAYES_AMD_NO_PYHOST"):        # (tests: force the ctypes route)
        raise ImportError("disabled by SBAYES_AMD_NO_PYHOST")
    from . import _sbe_pyhost as _h
except ImportError:                                   # not built: the ctypes route below
    _h = None

HAVE_EXTENSION = _h is not None

if _h is not None:
    addr = _h.addr
else:
    def addr(a):
        """The buffer address of an array as a plain int (what a c_void_p argument takes)."""
        return a.__array_interface__["data"][0]


def _boolc(a):
    if type(a) is not np.ndarray or a.dtype != np.bool_ or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=bool)
    return a


def subset_ids(objs, groups_new, groups_old, src_new, src_old):
    """Group ids [C, n] of both samples and source ids [n, F] of both for the listed objects (int32, C-contiguous), in one
    pass over the samples' own arrays (sbe_host_helpers.h: sbeh_subset_ids).  Returns (gid_old, gid_new, sid_old, sid_new), or
    None when there is no single id per object and component (an object listed twice, or in several groups)."""
    C, n, F = len(groups_new), objs.size, src_new.shape[1]
    gid_new = np.empty((C, n), dtype=np.int32)
    gid_old = np.empty((C, n), dtype=np.int32)
    sid_new = np.empty((n, F), dtype=np.uint8)
    sid_old = sid_new if src_old is src_new else np.empty((n, F), dtype=np.uint8)
    rc = -2
    if _h is not None:
        rc = _h.subset_ids(objs, groups_new, groups_old, src_new, src_old, gid_new, gid_old, sid_new, sid_old)
    if rc == -2:                                      # not built, or an argument is not a C-contiguous bool array: convert
        objs = np.ascontiguousarray(objs, dtype=np.int32)
        gn = [_boolc(g) for g in groups_new]
        go = [gn[c] if groups_old[c] is groups_new[c] else _boolc(groups_old[c]) for c in range(C)]
        sn = _boolc(src_new)
        so = sn if src_old is src_new else _boolc(src_old)
        for c in range(C):
            if go[c].shape != gn[c].shape or gn[c].shape[1] != sn.shape[0]:
                raise ValueError("group matrices of the two samples differ in shape")
        if _h is not None:
            rc = _h.subset_ids(objs, gn, go, sn, so, gid_new, gid_old, sid_new, sid_old)
        if rc == -2:
            pn = (ct.c_void_p * C)(*[addr(g) for g in gn])
            po = (ct.c_void_p * C)(*[addr(g) for g in go])
            ng = (ct.c_int32 * C)(*[g.shape[0] for g in gn])
            rc = _lib.load().sbe_host_subset_ids(addr(objs), n, sn.shape[0], F, C, ng, pn, po, addr(sn), addr(so),
                                                 addr(gid_new), addr(gid_old), addr(sid_new), addr(sid_old))
    if rc < 0:
        raise ValueError("object index out of range in object_subset")
    return None if rc else (gid_old, gid_new, sid_old, sid_new)


def diff_rows(new, mirror):
    """Indices (int32, ascending) of the rows of `new` that differ bytewise from `mirror` (same shape and dtype, `mirror`
    C-contiguous and writeable); those rows are copied INTO `mirror` (sbeh_diff_rows)."""
    n_rows = new.shape[0]
    if n_rows == 0 or new.size == 0:
        return np.zeros(0, dtype=np.int32)
    if not new.flags.c_contiguous:
        new = np.ascontiguousarray(new)
    idx = np.empty(n_rows, dtype=np.int32)
    k = _h.diff_rows(new, mirror, idx) if _h is not None else -2
    if k == -2:
        k = _lib.load().sbe_host_diff_rows(addr(new), addr(mirror), n_rows, new.nbytes // n_rows, addr(idx))
    if k < 0:
        raise ValueError("diff_rows: `new` and `mirror` differ in size")
    return idx[:k]


def diff_rows_among(new, mirror, cand_a, cand_b=None):
    """diff_rows restricted to the rows listed in `cand_a` (and `cand_b`): int32 index arrays, any order, repeats allowed --
    for a caller that KNOWS every other row equals the mirror's (binding.py: source lineage).  Ascending int32 result."""
    n_rows = new.shape[0]
    if n_rows == 0 or new.size == 0:
        return np.zeros(0, dtype=np.int32)
    if not new.flags.c_contiguous:
        new = np.ascontiguousarray(new)
    k = -2
    if _h is not None:
        idx = np.empty(cand_a.size + (cand_b.size if cand_b is not None else 0), dtype=np.int32)
        k = _h.diff_rows_among(new, mirror, cand_a, cand_b, idx)
        if k >= 0:
            return idx[:k]
    if k == -2:                                       # not built, or the candidates are not int32: the same with NumPy
        cand = np.asarray(cand_a).reshape(-1) if cand_b is None else np.concatenate([np.asarray(cand_a).reshape(-1), np.asarray(cand_b).reshape(-1)])
        cand = np.unique(cand.astype(np.int64))
        if cand.size == 0:
            return np.zeros(0, dtype=np.int32)
        if cand.size and (cand[0] < 0 or cand[-1] >= n_rows):
            raise ValueError("diff_rows_among: row index out of range")
        differs = (new[cand] != mirror[cand]).reshape(cand.size, -1).any(axis=1)
        rows = cand[differs].astype(np.int32)
        mirror[rows] = new[rows]
        return rows
    raise ValueError("diff_rows_among: row index out of range, or `new` and `mirror` differ in size")


def touched_groups(gid_old, gid_new, n_groups_total):
    """Sorted distinct group indices >= 0 among two int32 id arrays of equal size (np.union1d without the -1s)."""
    touched = np.empty(n_groups_total, dtype=np.int32)
    k = _h.touched_groups(gid_old, gid_new, n_groups_total, touched) if _h is not None else -2
    if k == -2:
        nt = ct.c_int32(0)
        k = _lib.load().sbe_host_touched_groups(addr(gid_old), addr(gid_new), gid_old.size, n_groups_total, addr(touched), ct.byref(nt))
        k = nt.value if k == 0 else -1
    if k < 0:
        raise ValueError("group index out of range in gid_old / gid_new")
    return touched[:k]


# ---- the sample cache's node protocol (sbayes/sampling/state.py:215-321) -------------------------------------------------------
# is_outdated / what_changed / set_up_to_date in native code for the node classes registered here (csrc/sbe_pyhost.c: node_*);
# any other node class is served by its own methods, from inside the same functions.  Registered: sbayes_amd.state's classes
# (at import) and the reference's (patch.install, when their source is the revision mirrored).
_NODE_PLAIN, _NODE_GROUPED = [], []


def register_node_classes(plain, grouped):
    """`plain`: a CacheNode class whose is_outdated / ahead_of / what_changed / set_up_to_date / edit / version / value are the
    reference's (exact class, not subclasses); `grouped`: its GroupedParameters class (isinstance)."""
    if plain not in _NODE_PLAIN:
        _NODE_PLAIN.append(plain)
    if grouped not in _NODE_GROUPED:
        _NODE_GROUPED.append(grouped)
    if _h is not None:
        _h.node_setup(tuple(_NODE_PLAIN), tuple(_NODE_GROUPED), _empty_i64)


def unregister_node_classes(plain, grouped):
    if plain in _NODE_PLAIN:
        _NODE_PLAIN.remove(plain)
    if grouped in _NODE_GROUPED:
        _NODE_GROUPED.remove(grouped)
    if _h is not None:
        _h.node_setup(tuple(_NODE_PLAIN), tuple(_NODE_GROUPED), _empty_i64)


def _empty_i64(n):
    return np.empty(n, dtype=np.int64)


if _h is not None:
    _h.node_setup((), (), _empty_i64)
    node_outdated = _h.node_outdated
    node_changed = _h.node_changed
    node_commit = _h.node_commit
    node_update_value = _h.node_update_value
else:
    def node_update_value(cache, value):
        cache.update_value(value)

    def node_outdated(cache):
        return cache.is_outdated()

    def node_changed(cache, key, caching=True):
        return cache.what_changed(key, caching=caching)

    def node_commit(cache):
        cache.set_up_to_date()


# ---- classes whose read-only properties the native bind reads from the instance (csrc/sbe_pyhost.c: trust_setup) ----------------
_TRUSTED = ([], [], [], [])     # sample, parameter, confounder prior classes; count classes with the plain resolve_sharing (exact classes)


def register_trusted(samples=(), params=(), conf_priors=(), counts=()):
    for have, new in zip(_TRUSTED, (samples, params, conf_priors, counts)):
        for cls in new:
            if cls not in have:
                have.append(cls)
    if _h is not None:
        _h.trust_setup(*(tuple(x) for x in _TRUSTED))


def unregister_trusted(samples=(), params=(), conf_priors=(), counts=()):
    for have, gone in zip(_TRUSTED, (samples, params, conf_priors, counts)):
        for cls in gone:
            if cls in have:
                have.remove(cls)
    if _h is not None:
        _h.trust_setup(*(tuple(x) for x in _TRUSTED))
