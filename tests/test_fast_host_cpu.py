"""sbayes_amd._fast / the CPython extension sbayes_amd._sbe_pyhost (csrc/sbe_pyhost.c): the native glue of the host layer
gives what the Python / NumPy forms it replaces give -- addresses, subset ids (against the reference expressions of
counts.py:21-27), row diffs, touched groups, and the bind cache's token comparison (binding._token / _same)."""
import ctypes as ct
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import _fast, _lib, binding
from sbayes_amd import state as st

REPO = Path(__file__).resolve().parent.parent
ext = pytest.mark.skipif(not _fast.HAVE_EXTENSION, reason="sbayes_amd._sbe_pyhost not built (python __graft_entry__.py)")


def test_the_extension_is_built_here():
    """build() compiles it; the suite runs the native route, not the fallback."""
    assert _fast.HAVE_EXTENSION or os.environ.get("SBAYES_AMD_NO_PYHOST")


def test_the_ctypes_route_gives_the_same_host_layer():
    """Without the extension (SBAYES_AMD_NO_PYHOST=1: what a box without a C compiler gets) the same helpers run through
    the engine library's sbe_host_* exports and the bind cache compares its tokens in Python: this file's checks and the host
    logic tests pass unchanged."""
    if os.environ.get("SBAYES_AMD_NO_PYHOST"):
        pytest.skip("already on the ctypes route")
    env = dict(os.environ, SBAYES_AMD_NO_PYHOST="1")
    res = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", str(REPO / "tests" / "test_fast_host_cpu.py"),
                          str(REPO / "tests" / "test_host_logic_cpu.py"), str(REPO / "tests" / "test_dynamic_priors_cpu.py")],
                         cwd=str(REPO), env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-2000:]
    assert "passed" in res.stdout


@ext
def test_addr_is_the_buffer_address_for_any_layout():
    rng = np.random.default_rng(0)
    a = rng.random((7, 5, 3))
    for v in (a, a[2:], a[:, 1:4, 2], a.T, a[::-1], np.zeros((0, 4)), a.astype(np.float32), a > 0.5):
        assert _fast.addr(v) == v.__array_interface__["data"][0]
    a.setflags(write=False)                                  # frozen arrays (the reference freezes its parameters)
    assert _fast.addr(a) == a.ctypes.data
    with pytest.raises(TypeError):
        _fast.addr([1, 2, 3])


def _reference_ids(groups, source, objs, offset):
    sub = groups[:, objs]
    gid = np.where(sub.any(axis=0), sub.argmax(axis=0) + offset, -1).astype(np.int32)
    return gid


@pytest.mark.parametrize("seed", range(6))
def test_subset_ids_against_numpy_and_the_c_abi(seed):
    rng = np.random.default_rng(seed)
    N, F, C = int(rng.integers(5, 80)), int(rng.integers(1, 40)), int(rng.integers(1, 5))
    G = [int(rng.integers(1, 6)) for _ in range(C)]

    def groups():
        out = []
        for g in G:
            a = rng.integers(0, g + 1, size=N)
            out.append(a[None, :] == np.arange(g)[:, None])
        return out
    g_new, g_old = groups(), groups()
    for c in range(1, C):                                    # confounders: the same matrix object in both samples
        g_old[c] = g_new[c]

    def source():
        pick = rng.integers(0, C + 1, size=(N, F))
        return pick[..., None] == np.arange(C)
    s_new, s_old = source(), source()
    objs = np.sort(rng.choice(N, size=int(rng.integers(1, N + 1)), replace=False)).astype(np.int32)
    off = np.concatenate([[0], np.cumsum(G)])
    got = _fast.subset_ids(objs, g_new, g_old, s_new, s_old)
    assert got is not None
    gid_old, gid_new, sid_old, sid_new = got
    for c in range(C):
        assert np.array_equal(gid_new[c], _reference_ids(g_new[c], s_new, objs, off[c]))
        assert np.array_equal(gid_old[c], _reference_ids(g_old[c], s_old, objs, off[c]))
    for sid, s in ((sid_new, s_new), (sid_old, s_old)):
        want = np.where(s[objs].any(-1), s[objs].argmax(-1), 255).astype(np.uint8)
        assert np.array_equal(sid, want)
    # the same source object in both samples: one id array serves both
    again = _fast.subset_ids(objs, g_new, g_old, s_new, s_new)
    assert again[2] is again[3] and np.array_equal(again[3], sid_new)
    # the exported C-ABI helper (what a C caller binds) agrees
    lib = _lib.load()
    pn = (ct.c_void_p * C)(*[g.ctypes.data for g in g_new])
    po = (ct.c_void_p * C)(*[g.ctypes.data for g in g_old])
    ng = (ct.c_int32 * C)(*G)
    o = [np.empty((C, objs.size), np.int32), np.empty((C, objs.size), np.int32), np.empty((objs.size, F), np.uint8),
         np.empty((objs.size, F), np.uint8)]
    assert lib.sbe_host_subset_ids(objs.ctypes.data, objs.size, N, F, C, ng, pn, po, s_new.ctypes.data, s_old.ctypes.data,
                                   *[a.ctypes.data for a in o]) == 0
    assert np.array_equal(o[0], gid_new) and np.array_equal(o[1], gid_old) and np.array_equal(o[2], sid_new) and np.array_equal(o[3], sid_old)
    # no single id: an object listed twice, an object in two groups of a component
    assert _fast.subset_ids(np.array([objs[0], objs[0]], dtype=np.int32), g_new, g_old, s_new, s_old) is None
    if G[0] > 1:
        over = [g.copy() for g in g_new]
        over[0][:2, objs[0]] = True
        assert _fast.subset_ids(objs, over, g_old, s_new, s_old) is None
    with pytest.raises(ValueError):
        _fast.subset_ids(np.array([N], dtype=np.int32), g_new, g_old, s_new, s_old)
    # ids handed over as float32 (same item size as int32) are converted, not reinterpreted
    as_float = _fast.subset_ids(objs.astype(np.float32), g_new, g_old, s_new, s_old)
    for a, b in zip(as_float, got):
        assert np.array_equal(a, b)
    # arguments that are not C-contiguous bool arrays are converted (Fortran order, uint8 views, int64 ids)
    alt = _fast.subset_ids(objs.astype(np.int64), [np.asfortranarray(g) for g in g_new], [np.asfortranarray(g) for g in g_old],
                           np.asfortranarray(s_new), s_old.view(np.uint8))
    for a, b in zip(alt, got):
        assert np.array_equal(a, b)


def test_diff_rows_and_touched_groups():
    rng = np.random.default_rng(3)
    for shape, dtype in (((6, 4, 3), np.float32), ((9, 16), np.bool_), ((1, 5), np.float64), ((0, 3), np.float32)):
        mirror = (rng.random(shape) * 4).astype(dtype)
        new = mirror.copy()
        rows = np.flatnonzero(rng.random(shape[0]) < 0.4)
        for r in rows:
            new[r].flat[0] = not new[r].flat[0] if dtype == np.bool_ else new[r].flat[0] + 1
        got = _fast.diff_rows(new, mirror)
        assert got.dtype == np.int32 and np.array_equal(got, rows)
        assert np.array_equal(mirror, new)                   # the mirror took the differing rows
        assert _fast.diff_rows(new, mirror).size == 0
    v = np.asfortranarray(rng.random((5, 4)).astype(np.float32))       # a non-contiguous `new` is compared row by row all the same
    m = np.zeros((5, 4), dtype=np.float32)
    assert np.array_equal(_fast.diff_rows(v, m), np.arange(5)) and np.array_equal(m, v)
    go = np.array([[0, -1, 3], [5, 5, -1]], dtype=np.int32)
    gn = np.array([[1, -1, 3], [5, 6, -1]], dtype=np.int32)
    assert np.array_equal(_fast.touched_groups(go, gn, 8), np.union1d(go[go >= 0], gn[gn >= 0]))
    with pytest.raises(ValueError):
        _fast.touched_groups(go, gn, 6)                      # 6 is out of range


@ext
def test_scan_is_the_python_token_comparison():
    """binding._scan (the extension's C loop) against binding._scan_py (_token / _same) over every kind of entry the bind
    cache records: versioned parameters (same / bumped / replaced array), frozen arrays, thawed arrays, writeable arrays
    compared by content, lists, nothing cached."""
    assert binding._scan is not binding._scan_py
    rng = np.random.default_rng(5)

    def versioned(version=3):
        p = st.ArrayParameter(rng.random((4, 3))) if hasattr(st, "ArrayParameter") else None
        if p is None:
            class P:                                          # (a parameter as the reference's: .value and .version)
                pass
            p = P()
            p.value = rng.random((4, 3))
        p.version = version
        return p
    frozen = rng.random((3, 2))
    frozen.setflags(write=False)
    writeable = rng.random((3, 2))
    view = rng.random((6, 2))[::2]
    params, cached = [], []

    def add(param, entry):
        params.append(param)
        cached.append(entry)
    p = versioned()
    add(p, binding._remember(binding._token(p)))             # same object, same version
    p2 = versioned()
    e2 = binding._remember(binding._token(p2))
    p2.version += 1
    add(p2, e2)                                              # version bumped
    p3 = versioned()
    e3 = binding._remember(binding._token(p3))
    p3._value = p3.value.copy() if hasattr(p3, "_value") else None
    if p3._value is None:
        p3.value = p3.value.copy()
    add(p3, e3)                                              # copy-on-write: another array object, same version
    add(frozen, binding._remember(binding._token(frozen)))   # frozen, recorded without a copy
    thawed = rng.random((3, 2))
    thawed.setflags(write=False)
    e5 = binding._remember(binding._token(thawed))
    thawed.setflags(write=True)
    add(thawed, e5)                                          # thawed since: unknown content
    add(writeable, binding._remember(binding._token(writeable)))        # content compare: equal
    w2 = rng.random((3, 2))
    e7 = binding._remember(binding._token(w2))
    w2[0, 0] += 1
    add(w2, e7)                                              # content compare: differs
    add(view, binding._remember(binding._token(view)))       # a view: by content
    add([[1.0, 2.0]], binding._remember(binding._token(np.array([[1.0, 2.0]]))))   # not an ndarray: converted, by content
    add(versioned(), None)                                   # nothing cached
    add(frozen, (frozen, None, frozen.copy()))               # (an entry with a copy although frozen)
    tok_c, ch_c = binding._scan(params, cached)
    tok_p, ch_p = binding._scan_py(params, cached)
    assert ch_c == ch_p
    assert ch_p == sum(1 << i for i in (1, 2, 4, 6, 9))
    for a, b in zip(tok_c, tok_p):
        assert (a[0] is b[0] or np.array_equal(a[0], b[0])) and a[1] == b[1]
    with pytest.raises(TypeError):
        binding._scan(params, cached[:-1])


@ext
def test_add_rows_many_refreezes_a_resolved_copy_and_checks_every_node_first():
    """(1) after resolve_sharing() the fresh `_value` copy is writeable; the reference's add_changes
    (sbayes/sampling/state.py:340-350) and both Python forms leave it frozen, so must the native form.  (2) dtype / contiguity /
    shape of EVERY node are checked before the first node is touched: a later node in another form returns None (Python
    route) with nothing applied, never a half-applied update."""
    from sbayes_amd.counts import apply_count_rows
    F, S = 3, 2

    def nodes():
        a = st.FeatureCounts(np.zeros((2, F, S), dtype=np.float32))
        b = st.FeatureCounts(np.zeros((3, F, S), dtype=np.float32))
        return a, b

    off = np.array([0, 2, 5], dtype=np.int64)
    touched = np.array([1, 3], dtype=np.int32)
    rows = np.ones((2, F, S), dtype=np.float32)
    a, b = nodes()
    a2 = a.copy()                                        # a and a2 now share their array
    assert a2.shared and a2.value is a.value
    bounds = _fast._h.add_rows_many([a2, b], off, touched, rows)
    assert bounds == [0, 1, 2]
    assert not a2.shared and a2.value is not a.value
    assert not a2.value.flags.writeable and not b.value.flags.writeable
    assert a.value.sum() == 0 and a2.value[1].sum() == F * S and b.value[1].sum() == F * S
    assert a2.version == 1 and list(a2.group_versions) == [0, 1] and list(b.group_versions) == [0, 1, 0]
    with pytest.raises(ValueError):
        a2.value[0, 0, 0] = 1.0
    # a node of another dtype BEHIND a good one: nothing is applied, the caller takes the Python route
    a, _ = nodes()
    bad = st.FeatureCounts(np.zeros((3, F, S), dtype=np.float64))
    assert _fast._h.add_rows_many([a, bad], off, touched, rows) is None
    assert a.version == 0 and a.value.sum() == 0 and list(a.group_versions) == [0, 0]
    # ... and an index beyond a later node's groups likewise
    a, b = nodes()
    assert _fast._h.add_rows_many([a, b], np.array([0, 2, 10], dtype=np.int64), np.array([1, 6], dtype=np.int32), rows) is None
    assert a.version == 0 and a.value.sum() == 0
    # the public entry serves the float64 node through the Python form with the same result
    a, _ = nodes()
    apply_count_rows({"x": a, "y": bad}, ["x", "y"], off, touched, rows)
    assert a.version == 1 and bad.version == 1 and bad.value[1].sum() == F * S and not bad.value.flags.writeable
