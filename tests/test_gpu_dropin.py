"""T1/T2 for the drop-in layer: sbayes_amd.{likelihood,conditionals,counts} used the way the
reference sampler uses sbayes.model.likelihood / sbayes.sampling.{conditionals,counts}, on the
mirror Sample (sbayes_amd/state.py), checked against reference golden vectors and a replayed
reference MCMC trace (cached pipeline, partial updates, delta counts)."""
import pickle

import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd import model as sbm
from sbayes_amd.conditionals import (conditional_effect_mean, likelihood_per_component,
                                     likelihood_per_component_exact, mixture_log_likelihood,
                                     observation_likelihoods)
from sbayes_amd.counts import compute_effect_counts, recalculate_feature_counts, update_feature_counts
from sbayes_amd.likelihood import (compute_component_likelihood, compute_component_likelihood_exact,
                                   normalize_weights, update_weights)
from sbayes_amd.registry import release_all
from tests._fixtures import load_npz, load_synthetic_trace, load_trace, sha

pytestmark = pytest.mark.gpu

NPZ = ["cfg1", "south_america", "test_files"]


TRACES = ["test_files", "south_america", "cfg1", "headline"]     # recorded reference MCMC traces


def load_case(name):
    """(fixture, trace): real configs keep their tensors in <name>.npz; the synthetic ones (cfg1: 50 x 30 x 5,
    headline: 1000 x 200 x 10 -- four feature tiles, several object chunks) regenerate them from the seed."""
    if name in ("cfg1", "headline"):
        return load_synthetic_trace(name)
    return load_npz(name), load_trace(name)


def names_of(fx):
    return fx.meta.get("component_names") or ["clusters", "universal"] + [f"conf{i}" for i in range(1, fx.n_comp - 1)]


def build(fx, with_counts=False):
    names = names_of(fx)
    return sbm.build(fx.features, fx.states_per_feature, names, fx.groups, fx.conc, fx.weights, fx.source,
                     counts=fx.counts if with_counts else None)


@pytest.fixture(autouse=True)
def _fresh_engines():
    yield
    release_all()


@pytest.mark.parametrize("name", NPZ)
def test_reference_call_surface(name):
    fx = load_npz(name)
    model, sample = build(fx)
    feats = model.data.features.values
    na = model.data.features.na_values
    assert np.array_equal(model.likelihood.na_features, na)

    # counts.py
    recalculate_feature_counts(feats, sample)
    for c, k in enumerate(sample.component_names):
        assert sample.feature_counts[k].value.dtype == np.float32
        assert np.array_equal(sample.feature_counts[k].value, fx.counts[c])
    assert np.array_equal(compute_effect_counts(feats, fx.groups[0], fx.source[..., 0]), fx.counts[0])

    # conditionals.likelihood_per_component (uncached and cached give the same array object)
    lh = likelihood_per_component(model, sample, caching=False)
    assert lh is sample.cache.component_likelihoods.value
    assert np.array_equal(lh, fx.z["lh_per_component"])
    assert likelihood_per_component(model, sample, caching=True) is lh

    # likelihood.update_weights / normalize_weights
    w = update_weights(sample, caching=False)
    assert w.dtype == np.float32 and np.array_equal(w, fx.z["weights_normalized"])
    assert np.array_equal(normalize_weights(fx.weights, sample.cache.has_components.value, features=feats), w)

    # the 8(d) composition exactly as the reference writes it
    with np.errstate(divide="ignore"):
        ll = np.log(np.sum(update_weights(sample) * likelihood_per_component(model, sample), axis=-1))[~na].sum()
    assert ll == fx.meta["mixture_ll"]
    # ... and the fused kernel
    fused = mixture_log_likelihood(model, sample)
    assert abs(fused - fx.meta["mixture_ll"]) <= 1e-10 * abs(fx.meta["mixture_ll"])
    assert np.array_equal(observation_likelihoods(model, sample), fx.z["obs_lh"])

    # Likelihood.__call__
    got = model.likelihood(sample, caching=False)
    assert abs(got - fx.meta["collapsed_ll"]) <= 1e-6 * abs(fx.meta["collapsed_ll"])
    for c, k in enumerate(sample.component_names):
        np.testing.assert_allclose(sample.cache.group_likelihoods[k].value, fx.group_lh[c], rtol=1e-6)
    assert model.likelihood(sample, caching=True) == got          # fully cached second call

    # exact (leave-one-out) forms
    assert np.array_equal(likelihood_per_component_exact(model, sample), fx.z["lh_exact"])
    with np.errstate(divide="ignore"):
        ll_exact = np.log(observation_likelihoods(model, sample, exact=True))[~na].sum()
    assert ll_exact == fx.meta["mixture_ll_exact"]


@pytest.mark.parametrize("name", NPZ)
def test_compute_component_likelihood_contract(name):
    fx = load_npz(name)
    model, _ = build(fx, with_counts=True)
    feats = model.data.features.values
    buf = fx.z["partial_before"].copy()
    out = compute_component_likelihood(features=feats, probs=fx.probs[0], groups=fx.groups[0],
                                       changed_groups=fx.z["partial_changed"], out=buf[..., 1])
    assert out.base is buf or out is buf[..., 1] or np.shares_memory(out, buf)
    assert np.array_equal(buf, fx.z["partial_after"])
    # exact variant against the oracle's restatement of likelihood.py:136-150
    c = 0
    post = fx.counts[c] + fx.conc[c]
    tables = [orc.normalize(post[None, g] - fx.features[fx.groups[c][g]] * fx.source[fx.groups[c][g], :, c, None])
              for g in range(fx.groups[c].shape[0])]
    want = orc.compute_component_likelihood_exact(fx.features, tables, fx.groups[c], np.arange(len(tables)),
                                                  np.full(fx.features.shape[:2], -3.0))
    got = compute_component_likelihood_exact(feats, tables, fx.groups[c], np.arange(len(tables)),
                                             np.full(fx.features.shape[:2], -3.0))
    assert np.array_equal(got, want)


def test_conditional_effect_mean_dropin():
    fx = load_npz("cfg1")
    model, _ = build(fx, with_counts=True)
    feats = model.data.features.values
    counts = fx.counts[0]
    prior = np.broadcast_to(fx.conc[0], counts.shape)
    unif = np.broadcast_to(fx.states_per_feature.astype(float), counts.shape)
    assert np.array_equal(conditional_effect_mean(prior, counts, features=feats), fx.z["cem_plain"])
    assert np.array_equal(conditional_effect_mean(prior, counts, unif_counts=unif, prior_temperature=1.7,
                                                  temperature=2.5, features=feats), fx.z["cem_temp"])


def test_likelihood_survives_pickle_and_recreates_engine():
    fx = load_npz("cfg1")
    model, sample = build(fx, with_counts=True)
    before = model.likelihood(sample, caching=False)
    clone = pickle.loads(pickle.dumps(model.likelihood))
    release_all()                                   # the "other process" has no engine yet
    sample.cache.clear()
    assert clone(sample, caching=True) == before


@pytest.mark.parametrize("name", TRACES)
def test_trace_replay_cached_pipeline(name):
    """Replay the recorded reference MCMC trace through the cached drop-in pipeline: every step
    builds a candidate by copy(), applies the recorded state delta through the Sample API,
    delta-updates the counts on the device and evaluates with caching=True."""
    fx, tr = load_case(name)
    model, sample = build(fx)
    feats = model.data.features.values
    na = model.data.features.na_values
    recalculate_feature_counts(feats, sample)
    likelihood_per_component(model, sample, caching=True)
    model.likelihood(sample, caching=True)
    n_partial = 0
    for i in range(tr.n_steps):
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
        n_partial += 0 < len(changed) < cand.n_clusters
        lh = likelihood_per_component(model, cand, caching=True)
        assert sha(lh) == str(tr.lh_sha[i]), f"step {i} ({tr.operator[i]})"
        w = update_weights(cand, caching=True)
        with np.errstate(divide="ignore"):
            assert np.log(np.sum(w * lh, axis=-1))[~na].sum() == tr.mixture_ll[i]
        ll = model.likelihood(cand, caching=True)
        assert abs(ll - tr.last_lh[i]) <= 1e-6 * abs(tr.last_lh[i]), f"step {i}"
        glh = np.concatenate([cand.cache.group_likelihoods[k].value for k in cand.component_names])
        np.testing.assert_allclose(glh, tr.group_lh[i], rtol=1e-6, atol=1e-6)
        if i % 25 == 0:
            fused = mixture_log_likelihood(model, cand)
            assert abs(fused - tr.mixture_ll[i]) <= 1e-10 * abs(tr.mixture_ll[i])
            chk = cand.copy()
            recalculate_feature_counts(feats, chk)
            for k in cand.component_names:
                assert np.array_equal(chk.feature_counts[k].value, cand.feature_counts[k].value)
        sample = cand
    if name in ("south_america", "headline"):
        assert n_partial > 20       # the partial-update path (strict subset of groups) was exercised


@pytest.mark.parametrize("name", TRACES)
def test_trace_replay_resident_flow(name):
    """the recorded reference trace replayed through the resident flow --
    state lives in engine slots, each step ships only the delta, counts / tables / collapsed and
    mixture likelihood are recomputed on the device."""
    from sbayes_amd.resident import ResidentChain
    fx, tr = load_case(name)
    model, sample = build(fx)
    chain = ResidentChain(model, sample)
    for c in range(fx.n_comp):
        assert np.array_equal(chain.current.counts(c), fx.counts[c])
    assert abs(chain.current.collapsed_loglik() - fx.meta["collapsed_ll"]) <= 1e-6 * abs(fx.meta["collapsed_ll"])
    prev_clusters, prev_source, prev_weights = fx.groups[0], fx.source, fx.weights
    n_rejected = 0
    for i in range(tr.n_steps):
        clusters, source, weights = tr.clusters(i), tr.source(i), tr.weights[i]
        changed_src = np.flatnonzero((source != prev_source).any(axis=(1, 2)))
        cand = chain.propose(
            clusters=clusters if not np.array_equal(clusters, prev_clusters) else None,
            source_rows=(changed_src, source[changed_src]),
            weights=weights if not np.array_equal(weights, prev_weights) else None)
        ll = cand.collapsed_loglik()
        assert abs(ll - tr.last_lh[i]) <= 1e-6 * abs(tr.last_lh[i]), f"step {i}"
        np.testing.assert_allclose(np.concatenate(cand.collapsed_group_logliks()), tr.group_lh[i], rtol=1e-6, atol=1e-6)
        mix = cand.mixture_loglik()
        assert abs(mix - tr.mixture_ll[i]) <= 1e-10 * abs(tr.mixture_ll[i]), f"step {i}"
        if i % 50 == 7:            # exercise the reject path: the current slot must be untouched
            chain.reject()
            n_rejected += 1
            cur = chain.current
            want = tr.mixture_ll[i - 1] if i else fx.meta["mixture_ll"]
            assert abs(cur.mixture_loglik() - want) <= 1e-10 * abs(want)
            cand = chain.propose(
                clusters=clusters if not np.array_equal(clusters, prev_clusters) else None,
                source_rows=(changed_src, source[changed_src]),
                weights=weights if not np.array_equal(weights, prev_weights) else None)
            assert abs(cand.mixture_loglik() - tr.mixture_ll[i]) <= 1e-10 * abs(tr.mixture_ll[i])
        chain.accept()
        prev_clusters, prev_source, prev_weights = clusters, source, weights
    assert n_rejected > 0
    counts = orc.recalculate_feature_counts(fx.features, [prev_clusters] + fx.groups[1:], prev_source)
    for c in range(fx.n_comp):
        assert np.array_equal(chain.current.counts(c), counts[c])


@pytest.mark.parametrize("name", TRACES)
def test_trace_replay_one_call_steps(name):
    """The recorded reference trace through sbe_step: one engine call (one PCIe round trip, one
    synchronisation) per MCMC step -- delta in; collapsed per-group and mixture log-likelihood out."""
    from sbayes_amd.resident import ResidentChain
    fx, tr = load_case(name)
    model, sample = build(fx)
    chain = ResidentChain(model, sample)
    prev_clusters, prev_source, prev_weights = fx.groups[0], fx.source, fx.weights
    for i in range(tr.n_steps):
        clusters, source, weights = tr.clusters(i), tr.source(i), tr.weights[i]
        changed_src = np.flatnonzero((source != prev_source).any(axis=(1, 2)))
        kwargs = dict(clusters=clusters if not np.array_equal(clusters, prev_clusters) else None,
                      source_rows=(changed_src, source[changed_src]),
                      weights=weights if not np.array_equal(weights, prev_weights) else None)
        ll, group_lh, mix = chain.step(**kwargs)
        assert abs(ll - tr.last_lh[i]) <= 1e-6 * abs(tr.last_lh[i]), f"step {i}"
        np.testing.assert_allclose(group_lh, tr.group_lh[i], rtol=1e-6, atol=1e-6)
        assert abs(mix - tr.mixture_ll[i]) <= 1e-10 * abs(tr.mixture_ll[i]), f"step {i}"
        if i % 40 == 3:
            chain.reject()                          # the current slot is untouched by a rejected step
            ll2, _, mix2 = chain.step(**kwargs)
            assert ll2 == ll and mix2 == mix
        chain.accept()
        prev_clusters, prev_source, prev_weights = clusters, source, weights
    counts = orc.recalculate_feature_counts(fx.features, [prev_clusters] + fx.groups[1:], prev_source)
    for c in range(fx.n_comp):
        assert np.array_equal(chain.current.counts(c), counts[c])


def test_step_forms_agree_bit_for_bit():
    """sbe_step's few-launch form (payload in mapped memory, fused table kernel, epilogue in mapped memory) and its
    call-by-call form give the same numbers bit for bit on random deltas -- cluster moves, source-row changes,
    weight changes, every source row at once, nothing at all -- with accepted and rejected steps; the counts are
    also checked against the oracle's full recount."""
    from sbayes_amd.engine import Engine
    from sbayes_amd.synthetic import make_state, make_workload
    wl = make_workload("cfg1")
    N, F, S = wl.shape
    rng = np.random.default_rng(3)
    engines = []
    for form in (0, 1):
        eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=2)
        eng.set_option(step_form=form)
        for c in range(wl.n_components):
            eng.set_concentration(c, wl.concentration[c])
        eng.load_state(0, wl.groups, wl.weights, source=wl.source)
        for c in range(wl.n_components):
            eng.update_probs(0, c)
        eng.mixture_loglik(0)                       # patterns uploaded
        engines.append(eng)
    try:
        cur_state = (wl.clusters.copy(), wl.weights.copy(), wl.source.copy())
        cur, cand = 0, 1
        for step in range(60):
            clusters, weights, source = (x.copy() for x in cur_state)
            kind = step % 6
            kw = {}
            if kind in (0, 1, 4):                   # move some objects between clusters
                for n in rng.integers(0, N, size=3):
                    clusters[:, n] = False
                    k = int(rng.integers(0, clusters.shape[0] + 1))
                    if k < clusters.shape[0]:
                        clusters[k, n] = True
                kw["clusters"] = clusters
            if kind in (0, 2, 4):                   # resample source rows of a few objects (valid components only)
                _cl, _w, src2 = make_state(wl.features, wl.groups[1:], clusters.shape[0], seed=500 + step)
                objs = np.unique(rng.integers(0, N, size=5 if kind != 4 else 4 * N)).astype(np.int32)
                hc = np.stack([clusters.any(axis=0)] + [g.any(axis=0) for g in wl.groups[1:]], axis=1)
                rows = src2[objs] & hc[objs][:, None, :]
                fix = ~rows.any(-1) & wl.features[objs].any(-1)          # component vanished: use the universal one
                rows[fix, 1] = True
                source[objs] = rows
                kw.update(changed_objects=objs, source_rows=rows)
            if kind in (3, 4):
                weights = rng.dirichlet(np.ones(wl.n_components), size=F).astype(np.float32)
                kw["weights"] = weights
            outs = [eng.step(cur, cand, **kw) for eng in engines]
            (g0, m0, c0), (g1, m1, c1) = outs
            assert np.array_equal(g0, g1), (step, kind)
            assert m0 == m1, (step, kind, m0, m1)
            assert np.array_equal(c0, c1), (step, kind)
            for c in range(wl.n_components):
                assert np.array_equal(engines[0].get_counts(cand, c), engines[1].get_counts(cand, c))
                assert np.array_equal(engines[0].get_probs(cand, c), engines[1].get_probs(cand, c))
            counts = orc.recalculate_feature_counts(wl.features, [clusters] + wl.groups[1:], source)
            for c in range(wl.n_components):
                assert np.array_equal(engines[0].get_counts(cand, c), counts[c]), (step, kind, c)
            if step % 4 != 3:                       # accept; else reject: the next step starts from `cur` again
                cur, cand = cand, cur
                cur_state = (clusters, weights, source)
    finally:
        for eng in engines:
            eng.close()
