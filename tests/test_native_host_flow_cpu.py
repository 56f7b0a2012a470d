"""The per-step control flow the host layer runs in native code (sbayes_amd/csrc/sbe_pyhost.c: node_*, likelihood_call,
store_per_object, update_counts, diff_rows_among) against the Python forms kept beside it -- the reference
forms: sbayes/sampling/state.py:215-321 (cache nodes), sbayes/model/likelihood.py:47-101 (Likelihood.__call__),
sbayes/model/prior.py:596-609 (SourcePrior's cache update), sbayes/sampling/counts.py:55-95 (update_feature_counts).

Two identical problems are driven through the same random edits, one through the native functions and one through the Python
forms; after every operation the values, their types, the cache nodes' recorded versions and the state the (oracle-backed)
engine double holds must be the same.  The real sampler on both routes is tests/test_reference_sampler_cpu.py."""
import copy
import types

import numpy as np
import pytest

from sbayes_amd import _fast, binding, conditionals, counts as my_counts, likelihood, model as sbm, registry
from sbayes_amd import state as st
from sbayes_amd.synthetic import make_workload
from tests._fake_engine import make_get_engine

ext = pytest.mark.skipif(not _fast.HAVE_EXTENSION, reason="sbayes_amd._sbe_pyhost is not built")


def _problem(monkeypatch, engines, seed=3, cls=None):
    wl = make_workload("cfg1", state_seed=seed)
    get_engine = make_get_engine(engines, cls)
    for mod in (registry, likelihood, conditionals, my_counts, binding):
        monkeypatch.setattr(mod, "get_engine", get_engine, raising=True)
    names = ["clusters"] + [f"conf{i}" for i in range(1, wl.n_components)]
    model, sample = sbm.build(wl.features, wl.states_per_feature, names, list(wl.groups), list(wl.concentration), wl.weights, wl.source)
    my_counts.recalculate_feature_counts(model.data.features.values, sample)
    return wl, model, sample


def _node_state(node):
    return (node.cached_version, {k: np.array(v) for k, v in node.cached_group_versions.items()}, np.array(node._value) if isinstance(node._value, np.ndarray) else None)


def _same_nodes(a, b):
    va, ga, xa = _node_state(a)
    vb, gb, xb = _node_state(b)
    assert va == vb and ga.keys() == gb.keys()
    for k in ga:
        assert np.array_equal(ga[k], gb[k]) and not a.cached_group_versions[k].flags.writeable
    if xa is not None and not all(v == -1 for v in np.ravel(np.array(va, dtype=object))):      # (a node never computed holds np.empty)
        assert np.array_equal(xa, xb)


def _all_nodes(sample):
    c = sample.cache
    return [c.component_likelihoods, c.weights_normalized, c.source_prior, *c.group_likelihoods.values()] if hasattr(c, "source_prior") else \
        [c.component_likelihoods, c.weights_normalized, *c.group_likelihoods.values()]


def _move(rng, sample, wl):
    """A proposal-like edit: one object leaves / joins a cluster and a few source rows change; returns (new sample, objects)."""
    new = sample.copy()
    n = wl.shape[0]
    obj = int(rng.integers(0, n))
    k = int(rng.integers(0, new.clusters.value.shape[0]))
    member_of = np.flatnonzero(new.clusters.value[:, obj])
    if member_of.size:                                           # leaves its cluster ...
        k = int(member_of[0])
    with new.clusters.edit_cluster(k) as c:                       # ... or joins cluster k (never two clusters at once)
        c[obj] = not c[obj]
    objs = np.unique(np.concatenate([[obj], rng.integers(0, n, size=int(rng.integers(0, 3)))])).astype(np.int32)
    src = new.source.value[objs].copy()
    comp = rng.integers(0, src.shape[-1], size=src.shape[:2])
    onehot = comp[..., None] == np.arange(src.shape[-1])
    onehot &= src.any(axis=-1, keepdims=True)                      # NA observations keep no source
    new.source.set_groups(objs, onehot)
    return new, objs


@ext
def test_node_protocol_is_the_python_protocol(monkeypatch):
    engines = {}
    wl, model, sample = _problem(monkeypatch, engines)
    rng = np.random.default_rng(5)
    for it in range(40):
        for node in _all_nodes(sample):
            assert _fast.node_outdated(node) == node.is_outdated()
            for key, inpt in node.inputs.items():
                if isinstance(inpt, st.GroupedParameters):
                    for caching in (True, False):
                        got, want = _fast.node_changed(node, key, caching), node.what_changed(key, caching=caching)
                        assert got.dtype == want.dtype == np.int64 and np.array_equal(got, want)
        # commit: native on one copy of the node, Python on another
        for node in _all_nodes(sample):
            twin = copy.copy(node)
            twin.cached_group_versions = dict(node.cached_group_versions)
            _fast.node_commit(node)
            twin.set_up_to_date()
            _same_nodes(node, twin)
            assert not _fast.node_outdated(node)
        new, objs = _move(rng, sample, wl)
        my_counts.update_feature_counts(sample, new, model.data.features.values, objs)
        if it % 3 == 0:
            with new.weights.edit() as w:
                w[...] = rng.dirichlet(np.ones(w.shape[1]), size=w.shape[0]).astype(np.float32)
        sample = new
    with pytest.raises(ValueError, match="GroupedParameters"):
        _fast.node_changed(sample.cache.weights_normalized, "weights")


@ext
def test_diff_rows_among_is_the_full_compare_on_the_listed_rows():
    rng = np.random.default_rng(11)
    for _ in range(50):
        n, row = int(rng.integers(1, 60)), (int(rng.integers(1, 7)), int(rng.integers(1, 4)))
        new = rng.random((n,) + row) < 0.5
        mirror = new.copy()
        changed = np.unique(rng.integers(0, n, size=int(rng.integers(0, 6))))
        for r in changed:
            mirror[r] = ~mirror[r]
        extra = rng.integers(0, n, size=int(rng.integers(0, 5)))
        a = rng.permutation(np.concatenate([changed, extra])).astype(np.int32)
        b = rng.permutation(changed).astype(np.int32) if rng.random() < 0.5 else None
        m1, m2, m3 = mirror.copy(), mirror.copy(), mirror.copy()
        got = _fast.diff_rows_among(new, m1, a, b)
        want = _fast.diff_rows(new, m2)
        assert got.dtype == np.int32 and np.array_equal(got, want) and np.array_equal(m1, new)
        # the NumPy form (no extension / candidates of another dtype) gives the same
        got2 = _fast.diff_rows_among(new, m3, a.astype(np.int64), None if b is None else b.astype(np.int64))
        assert np.array_equal(got2, want) and np.array_equal(m3, new)
    with pytest.raises(ValueError):
        _fast.diff_rows_among(np.zeros((3, 2), dtype=bool), np.zeros((3, 2), dtype=bool), np.array([3], dtype=np.int32))


def _python_forms(monkeypatch):
    """Route Likelihood.__call__, SourcePrior's store and update_feature_counts through their Python forms."""
    no_ext = types.SimpleNamespace(_h=None, node_outdated=lambda c: c.is_outdated(), node_update_value=lambda c, v: c.update_value(v),
                                   node_changed=lambda c, k, caching=True: c.what_changed(k, caching=caching), node_commit=lambda c: c.set_up_to_date(),
                                   HAVE_EXTENSION=False)
    monkeypatch.setattr(likelihood, "_fast", no_ext)
    monkeypatch.setattr(conditionals, "_fast", no_ext)
    monkeypatch.setattr(my_counts, "_NATIVE_UPDATE", False)


@ext
def test_native_likelihood_call_and_update_counts_against_the_python_forms(monkeypatch):
    """The same seeded sequence of proposals, accepted or rejected at random, on two identical problems: one on the native route,
    one on the Python forms.  Same log-likelihoods (value and type), same count tables / versions / group versions, same cache
    nodes, and the engine double of each ends up holding the same slot state (its results come from that state: a row the
    lineage wrongly skipped would change them)."""
    results = []
    for python_forms in (False, True):
        with monkeypatch.context() as mp:
            engines = {}
            wl, model, sample = _problem(mp, engines)
            if python_forms:
                _python_forms(mp)
            feats = model.data.features.values
            rng = np.random.default_rng(17)
            trace = [model.likelihood(sample)]
            for it in range(60):
                new, objs = _move(rng, sample, wl)
                subset = objs if it % 2 else np.isin(np.arange(wl.shape[0]), objs)          # index list / bool mask
                my_counts.update_feature_counts(sample, new, feats, subset)
                ll = model.likelihood(new)
                trace.append(ll)
                if it % 5 == 4:
                    trace.append(model.likelihood(sample))                                   # back to the current sample, then on
                if rng.random() < 0.4:
                    sample = new
            eng = next(iter(engines.values()))
            slot = eng._slot(0)
            results.append((trace, sample, {k: np.array(v) for k, v in slot.items() if isinstance(v, np.ndarray)}))
    (t0, s0, e0), (t1, s1, e1) = results
    assert len(t0) == len(t1)
    for a, b in zip(t0, t1):
        assert type(a) is type(b) and a == b
    for k in s0.feature_counts:
        a, b = s0.feature_counts[k], s1.feature_counts[k]
        assert np.array_equal(a.value, b.value) and a.version == b.version and np.array_equal(a.group_versions, b.group_versions)
        assert not a.value.flags.writeable
    for a, b in zip(_all_nodes(s0), _all_nodes(s1)):
        _same_nodes(a, b)
    assert e0.keys() == e1.keys()
    for k in e0:
        assert np.array_equal(e0[k], e1[k]), k


@ext
def test_native_update_counts_hands_unserved_forms_to_python(monkeypatch):
    engines = {}
    wl, model, sample = _problem(monkeypatch, engines)
    feats = model.data.features.values
    new, objs = _move(np.random.default_rng(2), sample, wl)
    h = _fast._h
    assert h.update_counts(sample, new, feats, slice(0, 5), True) is NotImplemented                    # a slice
    assert h.update_counts(sample, new, feats, [int(o) for o in objs], True) is NotImplemented          # a list
    assert h.update_counts(sample, new, feats, np.array([1, 1], dtype=np.int32), True) is NotImplemented    # a repeated object
    assert h.update_counts(sample, new, feats, objs.astype(np.int64), True) is NotImplemented           # int64 indices
    before = {k: (v.version, v.value.copy()) for k, v in new.feature_counts.items()}
    for k, v in new.feature_counts.items():                                                            # nothing was touched
        assert v.version == before[k][0] and np.array_equal(v.value, before[k][1])
    # the public entry serves all of them (Python form) and agrees with the native form on the forms that one serves
    twin = copy.deepcopy(new)
    my_counts.update_feature_counts(sample, new, feats, objs)
    my_counts.update_feature_counts(sample, twin, feats, [int(o) for o in objs])
    for k in new.feature_counts:
        assert np.array_equal(new.feature_counts[k].value, twin.feature_counts[k].value)
        assert new.feature_counts[k].version == twin.feature_counts[k].version


@ext
def test_source_lineage_sends_what_a_full_compare_would(monkeypatch):
    """Accept / reject sequences with a bind of the source after every proposal: the slot's source on the engine double equals the
    bound sample's, whichever lineage case (parent, sibling, back to the parent, unknown) the bind went through; and with the
    notes forgotten before every bind (full compares) the engine receives the same rows."""
    logs = []
    for forget in (False, True):
        with monkeypatch.context() as mp:
            engines = {}
            wl, model, sample = _problem(mp, engines)
            feats = model.data.features.values
            rng = np.random.default_rng(23)
            eng_log = []
            binding._bind_slot(engines[next(iter(engines))], model, sample, 0, with_source=True)
            eng = next(iter(engines.values()))
            real, real_delta = eng.set_source_rows, eng.set_slot_delta

            def logged(slot, objects, rows, _real=real, _log=eng_log):
                _log.append((np.array(objects), np.array(rows)))
                return _real(slot, objects, rows)

            def logged_delta(slot, **kw):                     # (source rows that go up with the bind's other row uploads)
                if kw.get("source_objects") is not None:
                    eng_log.append((np.array(kw["source_objects"]), np.array(kw["source_rows"])))
                return real_delta(slot, **kw)
            mp.setattr(eng, "set_source_rows", logged, raising=False)
            mp.setattr(eng, "set_slot_delta", logged_delta, raising=False)
            for it in range(50):
                new, objs = _move(rng, sample, wl)
                my_counts.update_feature_counts(sample, new, feats, objs)
                for s in ((new, sample, new) if it % 4 == 0 else (new,)):            # (ClusterJump binds new -> old -> new)
                    if forget:
                        binding.forget_source_lineage()
                    binding._bind_slot(eng, model, s, 0, with_source=True)
                    assert np.array_equal(eng._slot(0)["source"], s.source.value)
                if rng.random() < 0.35:
                    sample = new
            logs.append(eng_log)
    a, b = logs
    assert len(a) == len(b) and len(a) > 0
    for (oa, ra), (ob, rb) in zip(a, b):
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb)


def test_lean_samples_drop_the_stale_component_block_and_recompute_it_whole(monkeypatch):
    """likelihood.LazyBlock (patch.install(operators=True) sets LEAN_SAMPLES): a sample's cache.component_likelihoods block that is
    not current -- never computed, or computed for an earlier state -- is replaced by a shape-only stand-in when the likelihood
    of the sample is asked, so Sample.copy() (CacheNode.assign_from, state.py:317-321) stops copying 8 N F C bytes per proposal;
    likelihood_per_component (conditionals.py:152-223) materialises it and computes EVERY group (the node was cleared), giving
    the array an untouched twin gives."""
    import pickle
    results = []
    for lean in (False, True):
        with monkeypatch.context() as mp:
            engines = {}
            wl, model, sample = _problem(mp, engines)
            mp.setattr(likelihood, "LEAN_SAMPLES", lean)
            mp.setattr(likelihood, "_LEAN_MIN_BYTES", 1)
            feats = model.data.features.values
            rng = np.random.default_rng(4)
            first = np.array(conditionals.likelihood_per_component(model, sample))            # computed once (an initialiser does)
            for it in range(6):
                new, objs = _move(rng, sample, wl)
                my_counts.update_feature_counts(sample, new, feats, objs)
                model.likelihood(new)
                node = new.cache.component_likelihoods
                if lean:
                    assert type(node._value) is likelihood.LazyBlock and node._value.shape == first.shape and node.shape == first.shape
                    assert new.copy().cache.component_likelihoods._value is node._value          # copies share the stand-in
                    back = pickle.loads(pickle.dumps(node._value))
                    assert type(back) is likelihood.LazyBlock and back.shape == first.shape and back.dtype == first.dtype
                    with pytest.raises(RuntimeError, match="lean samples"):
                        node._value[0]
                    with pytest.raises(RuntimeError, match="lean samples"):
                        np.asarray(node._value)
                else:
                    assert type(node._value) is np.ndarray
                sample = new
            out = conditionals.likelihood_per_component(model, sample)
            assert type(sample.cache.component_likelihoods._value) is np.ndarray and not sample.cache.component_likelihoods.is_outdated()
            results.append((first, np.array(out)))
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    # a block that IS current stays (a weights-only change does not touch its inputs)
    with monkeypatch.context() as mp:
        engines = {}
        wl, model, sample = _problem(mp, engines)
        mp.setattr(likelihood, "LEAN_SAMPLES", True)
        mp.setattr(likelihood, "_LEAN_MIN_BYTES", 1)
        conditionals.likelihood_per_component(model, sample)
        model.likelihood(sample)
        assert type(sample.cache.component_likelihoods._value) is np.ndarray
