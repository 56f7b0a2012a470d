"""Host logic of the drop-in layer on CPU: the cached pipeline of sbayes_amd.{likelihood,
conditionals,counts} driven by the recorded reference MCMC traces, with the device replaced by a
test double (tests/_fake_engine.py, oracle-backed).  Checks what the host code owns: which groups
it recomputes, that stale rows survive, that caches are keyed on versions, copy-on-write, pickling."""
import pickle

import numpy as np
import pytest

from sbayes_amd import conditionals, likelihood, registry
from sbayes_amd import model as sbm
from sbayes_amd.counts import recalculate_feature_counts, update_feature_counts
from tests._fake_engine import make_engine_for_observations, FakeEngine, make_get_engine
from tests._fixtures import load_npz, load_synthetic_trace, load_trace, sha


@pytest.fixture()
def fake(monkeypatch):
    engines = {}

    get_engine = make_get_engine(engines)

    for mod in (registry, likelihood, conditionals):
        monkeypatch.setattr(mod, "get_engine", get_engine, raising=True)
    import sbayes_amd.counts as counts_mod
    monkeypatch.setattr(counts_mod, "get_engine", get_engine, raising=True)
    monkeypatch.setattr(registry, "_ENGINES", {})
    monkeypatch.setattr(registry, "engine_for_features",
                        lambda f: next((e for e in engines.values() if e.n_features == f), None)
                        or FakeEngine(np.zeros((1, f, 1), dtype=bool)))
    monkeypatch.setattr(registry, "engine_for_observations", make_engine_for_observations(engines))
    return engines


def build(fx):
    names = fx.meta.get("component_names") or ["clusters", "universal"] + [f"conf{i}" for i in range(1, fx.n_comp - 1)]
    return sbm.build(fx.features, fx.states_per_feature, names, fx.groups, fx.conc, fx.weights, fx.source)


@pytest.mark.parametrize("name", ["test_files", "south_america", "cfg1"])
def test_cached_pipeline_replays_reference_trace(name, fake):
    fx, tr = load_synthetic_trace(name) if name == "cfg1" else (load_npz(name), load_trace(name))
    model, sample = build(fx)
    feats = model.data.features.values
    na = model.data.features.na_values
    recalculate_feature_counts(feats, sample)
    conditionals.likelihood_per_component(model, sample, caching=True)
    model.likelihood(sample, caching=True)
    eng = next(iter(fake.values()))
    n_partial = n_skipped = 0
    for i in range(150 if name == "south_america" else tr.n_steps):
        cand = sample.copy()
        new_clusters, new_source, new_weights = tr.clusters(i), tr.source(i), tr.weights[i]
        moved = np.flatnonzero((new_clusters != sample.clusters.value).any(axis=0) |
                               (new_source != sample.source.value).any(axis=(1, 2)))
        for k in np.flatnonzero((new_clusters != sample.clusters.value).any(axis=1)):
            with cand.clusters.edit_cluster(int(k)) as row:
                row[:] = new_clusters[k]
        if (new_source != sample.source.value).any():
            with cand.source.edit() as src:
                src[moved] = new_source[moved]
        if not np.array_equal(new_weights, sample.weights.value):
            cand.weights.set_value(new_weights.copy())
        if moved.size:
            update_feature_counts(sample, cand, feats, moved)
        changed = cand.cache.component_likelihoods.what_changed(["clusters", "clusters_counts"], caching=True)
        eng.calls.clear()
        lh = conditionals.likelihood_per_component(model, cand, caching=True)
        assert sha(lh) == str(tr.lh_sha[i]), f"step {i}"
        gathered = [c[1] for c in eng.calls if c[0] == "component_lh"]
        if len(changed):
            assert tuple(int(c) for c in changed) in gathered           # only the changed clusters are regathered
            n_partial += len(changed) < cand.n_clusters
        else:
            n_skipped += 1
        w = likelihood.update_weights(cand, caching=True, features=feats)
        with np.errstate(divide="ignore"):
            assert np.log(np.sum(w * lh, axis=-1))[~na].sum() == tr.mixture_ll[i]
        ll = model.likelihood(cand, caching=True)
        assert ll == pytest.approx(tr.last_lh[i], rel=1e-12)
        # the candidate shares buffers with its parent until edited (copy-on-write)
        if not moved.size:
            assert cand.clusters.value is sample.clusters.value
        sample = cand
    if name == "south_america":
        assert n_partial > 0 and n_skipped > 0      # strict-subset updates and no-op steps both occurred


def test_uncached_equals_cached_and_cache_hit_does_no_device_work(fake):
    fx = load_npz("cfg1")
    model, sample = build(fx)
    feats = model.data.features.values
    recalculate_feature_counts(feats, sample)
    eng = next(iter(fake.values()))
    a = conditionals.likelihood_per_component(model, sample, caching=False).copy()
    assert np.array_equal(a, fx.z["lh_per_component"])
    eng.calls.clear()
    b = conditionals.likelihood_per_component(model, sample, caching=True)
    assert not eng.calls and np.array_equal(a, b)                      # pure cache hit
    v1 = model.likelihood(sample, caching=True)
    eng.calls.clear()
    assert model.likelihood(sample, caching=True) == v1 and not eng.calls
    assert model.likelihood(sample, caching=False) == pytest.approx(fx.meta["collapsed_ll"], rel=1e-12)


def test_bind_cache_sees_in_place_edits(fake):
    """unshared parameters are edited IN PLACE (same ndarray, version bumped: sbayes/sampling/state.py:43-61,
    340-350); a bind cache keyed on array identity would keep the slot's old data.  Bind, edit weights / counts /
    clusters / source in place, bind again: every edit reaches the engine, and an untouched sample re-sends nothing."""
    fx = load_npz("cfg1")
    model, sample = build(fx)
    feats = model.data.features.values
    recalculate_feature_counts(feats, sample)
    eng = conditionals._engine(model)
    conditionals._bind_slot(eng, model, sample, 0, with_source=True)
    eng.calls.clear()
    conditionals._bind_slot(eng, model, sample, 0, with_source=True)
    assert not [c for c in eng.calls if c[0].startswith("set_")]                      # nothing re-sent
    arrays = (sample.weights.value, sample.feature_counts["clusters"].value, sample.clusters.value, sample.source.value)
    with sample.weights.edit() as w:
        w[0] = w[0][::-1].copy()
    diff = np.zeros_like(sample.feature_counts["clusters"].value)
    diff[1, 2, 0] = 1.0
    sample.feature_counts["clusters"].add_changes(diff)
    free = int(np.flatnonzero(~sample.clusters.value.any(axis=0))[0])
    sample.clusters.add_object(1, free)
    with sample.source.edit() as src:
        src[free, 0, :] = False
    assert all(a is b for a, b in zip(arrays, (sample.weights.value, sample.feature_counts["clusters"].value,
                                               sample.clusters.value, sample.source.value)))    # edited in place
    eng.calls.clear()
    stale = conditionals._bind_slot(eng, model, sample, 0, with_source=True)
    kinds = [c[0] for c in eng.calls]
    # round 3: only the DELTA goes up -- the one object's source row, the one group's count row
    assert kinds.count("set_weights") == 1 and "set_source" not in kinds and "set_counts" not in kinds
    # (third session of round 4: the changed group matrix, the count row and the source row of ONE bind are one call)
    assert ("set_slot_delta", True, True, True) in eng.calls and 0 in stale
    assert not {"set_source_rows", "set_counts_rows", "set_groups"} & set(kinds)
    st = eng._slot(0)
    assert np.array_equal(st["weights"], sample.weights.value) and np.array_equal(st["source"], sample.source.value)
    assert np.array_equal(st["counts"][0], sample.feature_counts["clusters"].value)
    assert np.array_equal(st["groups"][0], sample.clusters.value)
    # a copy shares the arrays and the versions: nothing is re-sent for it either
    eng.calls.clear()
    conditionals._bind_slot(eng, model, sample.copy(), 0, with_source=True)
    assert not [c for c in eng.calls if c[0].startswith("set_")]


def test_collapsed_values_of_all_components_come_from_one_call_and_follow_the_counts(fake):
    """Likelihood.__call__ asks component after component (likelihood.py:58-63); the drop-in layer evaluates every
    component with the first request (ONE collapsed_loglik_all) and keeps the values on the slot's bind entry until a
    count row changes -- then the changed rows of ALL components go up in ONE set_counts_rows and the values are new."""
    fx = load_npz("south_america")                                    # three components
    model, sample = build(fx)
    feats = model.data.features.values
    recalculate_feature_counts(feats, sample)
    eng = next(iter(fake.values()))
    eng.calls.clear()
    v0 = model.likelihood(sample, caching=False)
    assert v0 == pytest.approx(fx.meta["collapsed_ll"], rel=1e-12)
    assert [c[0] for c in eng.calls if c[0].startswith("collapsed")] == ["collapsed_loglik_all"]
    # (caching=False recounts on the device first -- recount_bound drops the kept values, as it must; one call again)
    eng.calls.clear()
    assert model.likelihood(sample, caching=False) == v0
    assert [c[0] for c in eng.calls if c[0].startswith("collapsed")] == ["collapsed_loglik_all"]
    # one count row of the clusters and one of a confounder change: one row upload for both, one re-evaluation
    names = sample.component_names
    for name in (names[0], names[-1]):
        diff = np.zeros_like(sample.feature_counts[name].value)
        diff[0, 1, 0] = 1.0
        sample.feature_counts[name].add_changes(diff)
    eng.calls.clear()
    v1 = model.likelihood(sample, caching=True)
    kinds = [c[0] for c in eng.calls]
    assert kinds.count("set_counts_rows") == 1 and ("set_counts_rows", 2) in eng.calls and "set_counts" not in kinds
    assert kinds.count("collapsed_loglik_all") == 1 and v1 != v0
    from oracle import sbayes_oracle as orc
    want = sum(orc.collapsed_group_logliks(sample.feature_counts[n].value, c).sum() for n, c in zip(names, fx.conc))
    assert v1 == pytest.approx(want, rel=1e-12)


def test_normalize_weights_never_creates_engines_per_row_count(monkeypatch):
    """normalize_weights is also called with has_components[available] (operators.py:1086), whose row count
    changes from step to step; the engine is looked up by F only."""
    made = []

    class Stub:
        def __init__(self, features, n_groups, n_slots=1, device=0):
            self.n_objects, self.n_features = features.shape[:2]
            made.append(self)

        def normalize_weights(self, weights, has_components):
            return np.zeros((len(has_components),) + np.shape(weights), dtype=np.float32)

        def close(self):
            pass

    monkeypatch.setattr(registry, "Engine", Stub)
    monkeypatch.setattr(registry, "_ENGINES", {})
    monkeypatch.setattr(registry, "_KNOWN", {})
    monkeypatch.setattr(registry, "default_device", lambda: 0)
    monkeypatch.setattr(likelihood, "_ROW_MEMO", type(likelihood._ROW_MEMO)())     # (the stub's zeros must not be kept)
    w = np.full((7, 2), 0.5, dtype=np.float32)
    for n in (3, 50, 11, 3, 49):
        out = likelihood.normalize_weights(w, np.ones((n, 2), dtype=bool))
        assert out.shape == (n, 7, 2)
    assert len(made) == 1 and made[0].n_features == 7


def test_normalize_weights_rows_are_asked_of_the_device_once_per_pattern(monkeypatch):
    """Few rows (update_weights(sample)[object_subset], operators.py:819-842): each has_components pattern's row is a device
    result kept per weights content; the same rows again, or other objects with known patterns, cost no engine call; new
    weights do.  The values are the engine's (here: the oracle's), row for row, and the returned array is the caller's."""
    from oracle import sbayes_oracle as orc
    calls = []

    class Stub:
        def __init__(self, features, n_groups, n_slots=1, device=0):
            self.n_objects, self.n_features = features.shape[:2]

        def normalize_weights(self, weights, has_components):
            calls.append(np.array(has_components))
            return orc.normalize_weights(np.asarray(weights, dtype=np.float32), np.asarray(has_components, dtype=bool))

        def close(self):
            pass

    monkeypatch.setattr(registry, "Engine", Stub)
    monkeypatch.setattr(registry, "_ENGINES", {})
    monkeypatch.setattr(registry, "_KNOWN", {})
    monkeypatch.setattr(registry, "default_device", lambda: 0)
    monkeypatch.setattr(likelihood, "_ROW_MEMO", type(likelihood._ROW_MEMO)())
    rng = np.random.default_rng(5)
    w = rng.dirichlet(np.ones(3), size=9).astype(np.float32)
    hc = np.array([[1, 1, 1], [0, 1, 1], [1, 1, 1], [0, 1, 0]], dtype=bool)
    got = likelihood.normalize_weights(w, hc)
    assert np.array_equal(got, orc.normalize_weights(w, hc)) and len(calls) == 1
    assert len(calls[0]) == 7 and {tuple(r) for r in calls[0]} == {tuple((b >> c) & 1 for c in range(3)) for b in range(1, 8)}
    got[0] = -1.0                                            # the caller's own array: the kept rows are untouched
    again = likelihood.normalize_weights(w, np.array([[1, 0, 1], [1, 1, 1]], dtype=bool))
    assert np.array_equal(again, orc.normalize_weights(w, np.array([[1, 0, 1], [1, 1, 1]], dtype=bool))) and len(calls) == 1
    w2 = w.copy(); w2[0] = [0.2, 0.3, 0.5]
    assert np.array_equal(likelihood.normalize_weights(w2, hc), orc.normalize_weights(w2, hc)) and len(calls) == 2
    assert np.array_equal(likelihood.normalize_weights(w, hc[:1]), orc.normalize_weights(w, hc[:1])) and len(calls) == 2
    # more components than the ride-along covers: exactly the missing patterns are asked for
    w6 = rng.dirichlet(np.ones(6), size=9).astype(np.float32)
    hc6 = rng.random((5, 6)) < 0.6
    hc6[:, 0] = True
    assert np.array_equal(likelihood.normalize_weights(w6, hc6), orc.normalize_weights(w6, hc6))
    assert len(calls) == 3 and len(calls[2]) == len({r.tobytes() for r in hc6})
    # many rows: the direct call
    big = np.ones((likelihood._ROW_MEMO_ROWS + 1, 3), dtype=bool)
    assert likelihood.normalize_weights(w, big).shape == (len(big), 9, 3) and len(calls) == 4 and len(calls[3]) == len(big)


def test_get_engine_after_a_featureless_stand_in(monkeypatch):
    """normalize_weights() before any engine exists creates the stand-in under the key ("features", F);
    get_engine's overlap scan must step over it instead of reading it as an array key."""
    class Stub:
        def __init__(self, features, n_groups, n_slots=1, device=0):
            self.n_objects, self.n_features, self.n_states = features.shape
            self.n_groups = list(n_groups)

        def normalize_weights(self, weights, has_components):
            return np.zeros((len(has_components),) + np.shape(weights), dtype=np.float32)

        def close(self):
            pass

    monkeypatch.setattr(registry, "Engine", Stub)
    monkeypatch.setattr(registry, "_ENGINES", {})
    monkeypatch.setattr(registry, "_KNOWN", {})
    monkeypatch.setattr(registry, "default_device", lambda: 0)
    likelihood.normalize_weights(np.full((5, 2), 0.5, dtype=np.float32), np.ones((3, 2), dtype=bool))
    assert ("features", 5) in registry._ENGINES
    block = np.zeros((12, 5, 3), dtype=bool)
    eng = registry.get_engine(block, [2, 1])                   # raised IndexError before the fix
    assert eng.n_groups == [2, 1] and registry.get_engine(block) is eng


def test_registry_warns_about_a_second_engine_for_another_view_of_the_same_block(monkeypatch):
    """VERDICT r1 nit 9: engines are keyed on the data pointer; a caller that re-slices the feature block per call
    would silently get a second resident copy.  Now it is told."""
    class Stub:
        def __init__(self, features, n_groups, n_slots=1, device=0):
            self.n_objects, self.n_features, self.n_states = features.shape
            self.n_groups = list(n_groups)

        def close(self):
            pass

    monkeypatch.setattr(registry, "Engine", Stub)
    monkeypatch.setattr(registry, "_ENGINES", {})
    monkeypatch.setattr(registry, "default_device", lambda: 0)
    block = np.zeros((12, 5, 3), dtype=bool)
    other = np.zeros((12, 5, 3), dtype=bool)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        e1 = registry.get_engine(block)
        assert registry.get_engine(block) is e1                # same array: same engine, no warning
        registry.get_engine(other)                             # an unrelated block: no warning
    with pytest.warns(RuntimeWarning, match="second engine"):
        e2 = registry.get_engine(block[2:])                    # another view of the first block
    assert e2 is not e1


def test_likelihood_pickles_without_device_state(fake):
    fx = load_npz("cfg1")
    model, sample = build(fx)
    recalculate_feature_counts(model.data.features.values, sample)
    _ = model.likelihood.na_features
    clone = pickle.loads(pickle.dumps(model.likelihood))
    assert clone._na_features is None                                   # lazily re-derived in the new process
    assert np.array_equal(clone.features, model.likelihood.features)
    assert clone.source_index == model.likelihood.source_index


def test_argument_validation_happens_before_the_abi():
    """Shape / dtype errors are raised by the Python shim (SURVEY.md 8(b) 'Errors')."""
    from sbayes_amd.engine import Engine
    with pytest.raises(ValueError, match="n_objects, n_features, n_states"):
        Engine(np.zeros((3, 4), dtype=bool), [1])
    with pytest.raises(TypeError, match="bool"):
        Engine(np.zeros((3, 4, 2), dtype=np.float64), [1])


def test_sample_with_lazy_normalized_weights_pickles_and_copies(fake):
    """Round 3: update_weights() caches a lazily materialised NormalizedWeights that remembers its sample (weakly).  Samples
    are pickled by the reference (MC3 pipes, StateDumper) and copied on every proposal: the cached object must survive
    both -- as its two small inputs, without the sample reference -- and still give the reference's array."""
    fx = load_npz("cfg1")
    model, sample = build(fx)
    w = likelihood.update_weights(sample, features=model.data.features.values)
    assert isinstance(w, likelihood.NormalizedWeights) and w.sample_if_current() is sample
    want = np.asarray(w).copy()
    assert np.array_equal(want, fx.z["weights_normalized"])
    clone = pickle.loads(pickle.dumps(sample))
    cw = clone.cache.weights_normalized.value
    assert isinstance(cw, likelihood.NormalizedWeights) and cw.sample_if_current() is None
    assert np.array_equal(np.asarray(cw), want)
    cand = sample.copy()
    assert cand.cache.weights_normalized.value is w                                  # immutable: copies share it
    rows = w[[1, 3, 4]]
    assert np.array_equal(rows, want[[1, 3, 4]])
    sample.weights.set_value(sample.weights.value[:, ::-1].copy())
    assert w.sample_if_current() is None                                              # no longer that sample's weights


def test_update_feature_counts_lets_the_bound_slot_follow(fake):
    """update_feature_counts (counts.py:55-95) on a sample whose counts the engine slot holds: the device takes the difference
    in the call that computes it (counts_delta(follow_slot=0)) and the next bind sends no count rows; the jump operator's
    note, or a slot bound to other counts, leaves the slot alone and the rows go up with the next bind.  Either way the
    double's slot ends with the candidate's counts."""
    import sbayes_amd.counts as counts_mod
    fx = load_npz("cfg1")
    model, sample = build(fx)
    feats = model.data.features.values
    recalculate_feature_counts(feats, sample)
    model.likelihood(sample, caching=True)                    # binds slot 0 to `sample`
    eng = next(iter(fake.values()))

    def candidate(obj):
        cand = sample.copy()
        k_old = int(np.argmax(sample.clusters.value[:, obj])) if sample.clusters.value[:, obj].any() else -1
        k_new = (k_old + 1) % sample.clusters.value.shape[0]
        if k_old >= 0:
            with cand.clusters.edit_cluster(k_old) as row:
                row[obj] = False
        with cand.clusters.edit_cluster(k_new) as row:
            row[obj] = True
        return cand

    def slot_counts():
        return [np.array(eng._slot(0)["counts"][c]) for c in range(fx.n_comp)]

    from sbayes_amd.binding import _bind_slot
    names = list(sample.feature_counts)
    src_cluster = sample.source.value[..., 0].any(axis=1) & sample.clusters.value.any(axis=0)
    o1, o2, o3 = np.flatnonzero(src_cluster)[:3]              # members of a cluster with observations explained by it
    # 1. the slot holds the sample's counts: it follows, the bind that comes next has nothing to send
    cand = candidate(o1)
    eng.calls.clear()
    update_feature_counts(sample, cand, feats, np.array([o1]))
    assert [c[0] for c in eng.calls] == ["counts_delta"]
    assert any(not np.array_equal(cand.feature_counts[n].value, sample.feature_counts[n].value) for n in names)
    _bind_slot(eng, model, cand, 0)
    kinds = [c[0] for c in eng.calls]
    assert "set_counts_rows" not in kinds and "set_counts" not in kinds, kinds
    for c, name in enumerate(names):
        assert np.array_equal(slot_counts()[c], cand.feature_counts[name].value)
    # 2. the jump operator's note: the slot is left alone, the rows go up with the next bind
    _bind_slot(eng, model, sample, 0)                         # back to `sample` (a rejected step)
    cand = candidate(o2)
    counts_mod.note_jump_state(sample)
    eng.calls.clear()
    before = slot_counts()
    update_feature_counts(sample, cand, feats, np.array([o2]))
    assert all(np.array_equal(a, b) for a, b in zip(before, slot_counts()))
    model.likelihood(cand, caching=True)                      # (binds; forgets the note)
    assert any(c[0] == "set_counts_rows" or (c[0] == "set_slot_delta" and c[2]) for c in eng.calls)
    assert counts_mod._no_follow_from is None
    for c, name in enumerate(names):
        assert np.array_equal(slot_counts()[c], cand.feature_counts[name].value)
    # 3. a slot that holds OTHER counts (still the candidate's) is not touched by an update of `sample`'s copy
    cand2 = candidate(o3)
    eng.calls.clear()
    before = slot_counts()
    update_feature_counts(sample, cand2, feats, np.array([o3]))
    assert all(np.array_equal(a, b) for a, b in zip(before, slot_counts()))
    _bind_slot(eng, model, cand2, 0)
    assert any(c[0] == "set_counts_rows" or (c[0] == "set_slot_delta" and c[2]) for c in eng.calls)
    for c, name in enumerate(names):
        assert np.array_equal(slot_counts()[c], cand2.feature_counts[name].value)


def test_bind_failure_forgets_the_slot():
    """the bind's row diff copies the differing rows into the host mirror BEFORE they are sent; a setter that raises
    in between must not leave mirror and entry claiming rows the device never received -- the slot's entry is dropped, the
    next bind sends the rows again (both the native and the Python form of the bind)."""
    from sbayes_amd import binding

    fx = load_npz("cfg1")
    model, sample = build(fx)
    feats = model.data.features.values
    recalculate_feature_counts.__wrapped__(feats, sample) if hasattr(recalculate_feature_counts, "__wrapped__") else None
    n_groups = [g.shape[0] for g in fx.groups]
    for bind in (binding._bind_slot, binding._bind_slot_py):
        eng = FakeEngine(feats, n_groups)
        for c, name in enumerate(sample.feature_counts):           # (counts from the fixture: the sample's own tables)
            sample.feature_counts[name].set_value(np.asarray(fx.z[f"counts_{c}"], dtype=np.float32).copy())
        bind(eng, model, sample, 0, with_source=True)
        assert 0 in eng._bound and 0 in eng._mirror
        # a counts change, and the next setter fails
        node = sample.feature_counts["clusters"]
        diff = np.zeros(node.value.shape, dtype=np.float32)
        diff[0, 0, 0] = 1.0
        node.add_changes(diff=diff)
        want = node.value.copy()

        def boom(*a, **k):
            raise ValueError("injected")
        saved = {name: getattr(eng, name) for name in ("set_counts_rows", "set_slot_delta") if hasattr(eng, name)}
        for name in saved:
            setattr(eng, name, boom)
        with pytest.raises(ValueError, match="injected"):
            bind(eng, model, sample, 0, with_source=True)
        assert 0 not in eng._bound and 0 not in eng._mirror            # forgotten, not half-updated
        for name, fn in saved.items():
            setattr(eng, name, fn)
        bind(eng, model, sample, 0, with_source=True)                 # sends everything the slot needs
        assert np.array_equal(eng.get_counts(0, 0), want)
        node.add_changes(diff=-diff)
