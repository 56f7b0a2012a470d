"""SURVEY.md 8(f) rank 1 on the device: cluster-membership posterior (AlterCluster /
AlterClusterWide cores) against the reference operators' recorded outputs and the oracle.
Tolerance 1e-12 relative at temperature 1 (log-space accumulation + exp vs the reference's linear
np.prod; einsum/FMA order of the host NumPy is not part of the contract); 2e-6 for tempered runs
(float32 powf of the device vs glibc)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd import model as sbm
from sbayes_amd.counts import recalculate_feature_counts
from sbayes_amd.operators import cluster_log_marginals, compute_cluster_posterior, compute_raw_cluster_probs
from sbayes_amd.registry import release_all
from sbayes_amd.synthetic import make_state, make_workload
from tests._fixtures import load_npz

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _fresh_engines():
    yield
    release_all()


@pytest.mark.parametrize("name", ["south_america", "test_files"])
def test_cluster_posterior_matches_reference_operators(name):
    fx = load_npz(name)
    z = fx.z
    names = fx.meta["component_names"]
    model, sample = sbm.build(fx.features, fx.states_per_feature, names, fx.groups, fx.conc, fx.weights, fx.source,
                              counts=fx.counts)
    model.prior.prior_cluster_effect.uniform_concentration_array = z["cp_unif"]
    smoothing = float(z["cp_additive_smoothing"])
    for tag, (temp, ptemp, rtol) in {"t1": (1.0, 1.0, 1e-12), "mc3": (1.3, 1.5, 2e-6)}.items():
        for k in range(fx.groups[0].shape[0]):
            key = f"cp_{tag}_k{k}"
            available = z[key + "_available"]
            post = compute_cluster_posterior(model, sample, k, available, temperature=temp, prior_temperature=ptemp,
                                             additive_smoothing=smoothing)
            np.testing.assert_allclose(post, z[key + "_posterior"], rtol=rtol, atol=1e-300)
            raw = compute_raw_cluster_probs(model, sample, k, available, temperature=temp, prior_temperature=ptemp)
            np.testing.assert_allclose(raw, z[key + "_wide_raw"], rtol=rtol, atol=1e-300)
            # the form patch.install(operators=True) uses: the candidate table comes from the caller (here: the
            # reference's ClusterEffectProposals.gibbs formula, operators.py:1254-1282), the marginals from the device
            prior = model.prior.prior_cluster_effect
            u, a = np.asarray(prior.uniform_concentration_array), np.asarray(prior.concentration_array)
            table = orc.normalize(u + (a - u) / ptemp + fx.counts[0][[k]] / temp, axis=-1)
            m = np.exp(cluster_log_marginals(model, sample, table, available, temp, ptemp))
            np.testing.assert_allclose(m[1] / (m[0] + m[1] + np.finfo(np.float32).eps), z[key + "_wide_raw"],
                                       rtol=rtol, atol=1e-300)
            if k == 0 and tag == "t1":
                log_m = model.likelihood.engine.cluster_marginals(0, z[key + "_table"], np.flatnonzero(available))
                np.testing.assert_allclose(np.exp(log_m), z[key + "_marginal_z01"], rtol=1e-12)


def test_cluster_posterior_large_feature_count_does_not_underflow():
    """at the stress shape the reference's np.prod underflows to 0/0 = NaN; the
    log-space device result stays finite and agrees with a log-space evaluation of the oracle."""
    wl = make_workload("stress", shape=(300, 500, 20, 4, (20,), False))
    model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration,
                              wl.weights, wl.source)
    feats = model.data.features.values
    recalculate_feature_counts(feats, sample)
    counts = [sample.feature_counts[k].value for k in sample.component_names]
    available = (~wl.clusters.any(axis=0)) | wl.clusters[1]
    unif = wl.states_per_feature.astype(float)
    post = compute_cluster_posterior(model, sample, 1, available)
    assert np.all(np.isfinite(post)) and np.all((post > 0) & (post < 1))
    # oracle in log space: same weights / likelihood arrays, sum of logs instead of np.prod
    table = orc.conditional_effect_mean(wl.concentration[0], counts[0][[1]], unif_counts=unif,
                                        prior_temperature=1.0, temperature=1.0)
    lh = orc.likelihood_per_component(wl.features, wl.na_values, wl.groups, counts, wl.concentration)
    wz = orc.feature_weights_with_and_without(wl.weights, orc.has_components(wl.groups), available, 1.0)
    cl = np.einsum("...i,...i", wl.features[available], table)
    all_lh = lh[available].copy()
    all_lh[..., 0] = cl
    all_lh[wl.na_values[available], 0] = 1.0
    log_m = np.log(np.einsum("...i,...i", all_lh[None], wz)).sum(axis=-1)
    want = 1.0 / (1.0 + np.exp(log_m[0] - log_m[1]))
    want = (want + 1e-6) / (1 + 2e-6)
    np.testing.assert_allclose(post, want, rtol=1e-10)
    with np.errstate(all="ignore"):
        linear = orc.cluster_posterior(wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights,
                                       1, available, unif)
    assert not np.all(np.isfinite(linear))          # the reference's linear-space form breaks here


@pytest.mark.parametrize("name", ["south_america", "test_files"])
def test_source_resampling_cores(name):
    """calculate_source_posterior and component_likelihood_given_unchanged on
    the device == the reference's recorded outputs (bit-exact at temperature 1; float32 pow
    tolerance when tempered)."""
    from sbayes_amd.operators import calculate_source_posterior, component_likelihood_given_unchanged
    fx = load_npz(name)
    z = fx.z
    model, sample = sbm.build(fx.features, fx.states_per_feature, fx.meta["component_names"], fx.groups, fx.conc,
                              fx.weights, fx.source, counts=fx.counts)
    model.prior.prior_cluster_effect.uniform_concentration_array = z["cp_unif"]
    for conf, unif in zip(sample.confounders, z["sp_conf_unif"]):
        model.prior.prior_confounding_effects[conf].uniform_concentration_array = unif
    subset = z["sp_subset"]
    mask = np.isin(np.arange(fx.features.shape[0]), subset)
    for tag, (temp, ptemp) in {"t1": (1.0, 1.0), "mc3": (1.3, 1.5)}.items():
        post = calculate_source_posterior(model, sample, subset, temp, ptemp)
        assert post.dtype == np.float32
        if tag == "t1":
            assert np.array_equal(post, z[f"sp_{tag}_posterior"])
        else:
            np.testing.assert_allclose(post, z[f"sp_{tag}_posterior"], rtol=2e-6, atol=1e-30)
        for k in range(fx.groups[0].shape[0]):
            lik = component_likelihood_given_unchanged(model, sample, mask, k, temp, ptemp)
            assert lik.dtype == np.float32
            if tag == "t1":
                assert np.array_equal(lik, z[f"sp_{tag}_k{k}_lh_unchanged"]), k
            else:
                np.testing.assert_allclose(lik, z[f"sp_{tag}_k{k}_lh_unchanged"], rtol=3e-6, atol=1e-30)


@pytest.mark.parametrize("name", ["south_america", "test_files"])
def test_source_prior_and_logger_row(name):
    """per-object source prior (float32 logs: 2e-6 relative) and the
    LikelihoodLogger row (bit-exact in float64, hence in its float32 column)."""
    from sbayes_amd.conditionals import observation_likelihoods, source_prior
    fx = load_npz(name)
    z = fx.z
    model, sample = sbm.build(fx.features, fx.states_per_feature, fx.meta["component_names"], fx.groups, fx.conc,
                              fx.weights, fx.source, counts=fx.counts)
    eng = model.likelihood.engine
    total = source_prior(model, sample)
    per_object = eng.source_prior(0)
    np.testing.assert_allclose(per_object, z["spr_per_object"], rtol=2e-6, atol=1e-6)
    assert abs(total - float(z["spr_total"])) <= 2e-6 * abs(float(z["spr_total"]))
    row = observation_likelihoods(model, sample, exact=True).ravel()
    assert np.array_equal(row, orc.logger_row(z["weights_normalized"], z["lh_exact"]))
    assert np.array_equal(row.astype(np.float32), z["logger_row_f32"])


def _gibbs_fixture():
    fx = load_npz("gibbs_source")
    model, sample = sbm.build(fx.features, fx.states_per_feature, fx.meta["component_names"], fx.groups, fx.conc,
                              fx.weights, fx.source, counts=fx.counts)
    return fx, model, sample


@pytest.mark.parametrize("tag", ["all", "subset", "mc3", "prior"])
def test_gibbs_source_proposal_draw_for_draw(tag):
    """SURVEY.md 8(f) rank 3, the whole operator: GibbsSampleSource._propose with the reference's own
    uniforms -> the reference's new source rows, counts and (float32) log_q / log_q_back.  At
    temperature 1 everything is bit-exact; the tempered cases carry the float32 powf tolerance in the
    probabilities (the draws still agree: no uniform of the fixture sits within 1e-6 of a cdf step)."""
    from sbayes_amd.operators import gibbs_sample_source
    fx, model, sample = _gibbs_fixture()
    z = fx.z
    objects = z[f"gs_{tag}_objects"]
    temp, ptemp, from_prior = z[f"gs_{tag}_temps"]
    new, log_q, log_q_back = gibbs_sample_source(model, sample, objects, float(temp), float(ptemp), bool(from_prior),
                                                 z=z[f"gs_{tag}_z"])
    assert np.array_equal(new.source.value, z[f"gs_{tag}_new_source"])
    for c, name in enumerate(sample.component_names):
        assert np.array_equal(new.feature_counts[name].value, z[f"gs_{tag}_counts_{c}"])
    assert log_q.dtype == np.float32 and log_q_back.dtype == np.float32
    if temp == 1.0 and ptemp == 1.0:
        assert log_q == z[f"gs_{tag}_log_q"] and log_q_back == z[f"gs_{tag}_log_q_back"]
    else:
        assert abs(log_q - z[f"gs_{tag}_log_q"]) <= 2e-6 * abs(z[f"gs_{tag}_log_q"])
        assert abs(log_q_back - z[f"gs_{tag}_log_q_back"]) <= 2e-6 * abs(z[f"gs_{tag}_log_q_back"])
    # untouched objects keep their rows; the old sample is not modified
    others = np.setdiff1d(np.arange(sample.n_objects), objects)
    assert np.array_equal(new.source.value[others], fx.source[others])
    assert np.array_equal(sample.source.value, fx.source)


@pytest.mark.parametrize("tag", ["all", "subset", "mc3", "prior"])
def test_resident_gibbs_source_proposal(tag):
    """The same operator on the RESIDENT state (ResidentChain.propose_gibbs_source: nothing of the sample is
    re-uploaded): the reference's new source rows and counts bit for bit with the reference's uniforms, log_q /
    log_q_back to float32 accuracy (the device sums the float64 logs of the float32 probabilities; the reference sums
    float32 logs), candidate likelihoods equal to a from-scratch evaluation of the new state; reject leaves the
    current state untouched, accept makes the candidate current."""
    from sbayes_amd.registry import release_all
    from sbayes_amd.resident import ResidentChain
    fx, model, sample = _gibbs_fixture()
    z = fx.z
    objects = z[f"gs_{tag}_objects"]
    temp, ptemp, from_prior = z[f"gs_{tag}_temps"]
    try:
        chain = ResidentChain(model, sample)
        ll_cur, mix_cur = chain.current.collapsed_loglik(), chain.current.mixture_loglik()
        cand, log_q, log_q_back = chain.propose_gibbs_source(objects, float(temp), float(ptemp), bool(from_prior),
                                                             z=z[f"gs_{tag}_z"])
        eng = chain.eng
        all_objects = np.arange(sample.n_objects)
        assert np.array_equal(eng.get_source_rows(chain.cand, all_objects), z[f"gs_{tag}_new_source"])
        for c in range(eng.n_components):
            assert np.array_equal(cand.counts(c), z[f"gs_{tag}_counts_{c}"])
        assert abs(log_q - z[f"gs_{tag}_log_q"]) <= 3e-6 * abs(z[f"gs_{tag}_log_q"])
        assert abs(log_q_back - z[f"gs_{tag}_log_q_back"]) <= 3e-6 * abs(z[f"gs_{tag}_log_q_back"])
        # candidate likelihoods == the oracle's from-scratch evaluation of the new state
        groups = fx.groups
        new_source = z[f"gs_{tag}_new_source"]
        counts = orc.recalculate_feature_counts(fx.features, groups, new_source)
        na = ~fx.features.any(-1)
        want_mix = orc.mixture_loglik(fx.features, na, groups, counts, fx.conc, fx.weights)
        got_mix = cand.mixture_loglik()
        assert abs(got_mix - want_mix) <= 1e-10 * abs(want_mix)
        ll_cand = cand.collapsed_loglik()
        chain.reject()
        assert chain.current.collapsed_loglik() == ll_cur and chain.current.mixture_loglik() == mix_cur
        cand2, log_q2, log_q_back2 = chain.propose_gibbs_source(objects, float(temp), float(ptemp), bool(from_prior),
                                                                z=z[f"gs_{tag}_z"])
        assert (log_q2, log_q_back2) == (log_q, log_q_back)
        # the one-call form (sbe_gibbs_step) gives the same numbers
        chain.reject()
        log_q3, log_q_back3, ll3, glh3, mix3 = chain.gibbs_step(objects, float(temp), float(ptemp), bool(from_prior),
                                                                z=z[f"gs_{tag}_z"])
        assert abs(log_q3 - log_q) <= 1e-13 * abs(log_q) and abs(log_q_back3 - log_q_back) <= 1e-13 * abs(log_q_back)
        assert mix3 == got_mix and abs(ll3 - ll_cand) <= 1e-12 * abs(ll_cand)
        assert np.array_equal(eng.get_source_rows(chain.cand, all_objects), z[f"gs_{tag}_new_source"])
        for c in range(eng.n_components):
            assert np.array_equal(_SlotCounts(chain, c), z[f"gs_{tag}_counts_{c}"])
        chain.accept()
        assert chain.current.collapsed_loglik() == ll_cand and chain.current.mixture_loglik() == got_mix
    finally:
        release_all()


def _SlotCounts(chain, c):
    return chain.eng.get_counts(chain.cand, c)


def test_gibbs_source_same_global_rng_stream_as_reference():
    """Without explicit uniforms the operator draws np.random.random([n, F, 1]) like sample_categorical
    does: seeding np.random as the fixture did reproduces the recorded proposal."""
    from sbayes_amd.operators import gibbs_sample_source
    fx, model, sample = _gibbs_fixture()
    z = fx.z
    np.random.seed(1001)                                   # the seed the "subset" case was recorded with
    new, log_q, _ = gibbs_sample_source(model, sample, z["gs_subset_objects"])
    assert np.array_equal(new.source.value, z["gs_subset_new_source"])
    assert log_q == z["gs_subset_log_q"]


def test_device_log_q_and_oracle_agree():
    """The device's own fp64 log_q / log_q_back sums against the oracle restatement (fp32 sums there)."""
    fx, model, sample = _gibbs_fixture()
    z = fx.z
    eng = model.likelihood.engine
    from sbayes_amd.conditionals import _bind_slot
    _bind_slot(eng, model, sample, 0, with_source=True)
    for c in range(eng.n_components):
        eng.update_probs(0, c)
    for tag in fx.meta["cases"]:
        objects = z[f"gs_{tag}_objects"]
        temp, ptemp, from_prior = (float(v) for v in z[f"gs_{tag}_temps"])
        want_src, want_q, want_qb, want_counts = orc.gibbs_source_propose(
            fx.features, fx.na_values, fx.groups, fx.counts, fx.conc, fx.weights, fx.source, objects, z[f"gs_{tag}_z"],
            temp, ptemp, bool(from_prior))
        eng.copy_slot(1, 0)
        log_q = eng.sample_source(0, 1, objects, z[f"gs_{tag}_z"], temp, ptemp, bool(from_prior))
        assert np.array_equal(eng.get_source_rows(1, objects), want_src[objects])
        eng.update_counts(1, 0, objects)
        for c in range(eng.n_components):
            assert np.array_equal(eng.get_counts(1, c), want_counts[c])
            eng.update_probs(1, c)
        log_q_back = eng.source_logprob(1, 0, objects, temp, ptemp, bool(from_prior))
        assert abs(log_q - want_q) <= 3e-6 * abs(want_q)
        assert abs(log_q_back - want_qb) <= 3e-6 * abs(want_qb)


PHILOX_KAT = [   # Random123 known-answer vectors for philox4x32-10: (counter, key, output)
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_device_philox_known_answers_and_oracle():
    fx, model, sample = _gibbs_fixture()
    eng = model.likelihood.engine
    ck = np.array([c + k for c, k, _ in PHILOX_KAT], dtype=np.uint32)
    assert np.array_equal(eng.test_philox(ck), np.array([o for _, _, o in PHILOX_KAT], dtype=np.uint32))
    rng = np.random.default_rng(2)
    ck = rng.integers(0, 2 ** 32, size=(5000, 6), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(eng.test_philox(ck), orc.philox4x32_10(ck[:, :4], ck[:, 4:]))


def test_gibbs_source_device_rng_matches_oracle_stream_and_is_reproducible():
    """z = NULL: the engine's Philox stream.  The draws equal the oracle's sample_categorical fed with
    the oracle's restatement of that stream; the draw counter advances by one per call; the same
    (seed, draw) gives the same proposal, another draw a different one."""
    fx, model, sample = _gibbs_fixture()
    z = fx.z
    eng = model.likelihood.engine
    from sbayes_amd.conditionals import _bind_slot
    _bind_slot(eng, model, sample, 0, with_source=True)
    for c in range(eng.n_components):
        eng.update_probs(0, c)
    objects = z["gs_subset_objects"]
    F = fx.features.shape[1]
    eng.set_rng(seed=0x1234_5678_9ABC_DEF0, draw=41)
    rows = []
    for draw in (41, 42):
        assert eng.get_rng() == (0x1234_5678_9ABC_DEF0, draw)
        u = orc.philox_uniforms(0x1234_5678_9ABC_DEF0, draw, objects.size * F).reshape(objects.size, F)
        want_src, want_q, _, _ = orc.gibbs_source_propose(fx.features, fx.na_values, fx.groups, fx.counts, fx.conc,
                                                          fx.weights, fx.source, objects, u)
        eng.copy_slot(1, 0)
        log_q = eng.sample_source(0, 1, objects, None)
        rows.append(eng.get_source_rows(1, objects))
        assert np.array_equal(rows[-1], want_src[objects])
        assert abs(log_q - want_q) <= 3e-6 * abs(want_q)
    assert not np.array_equal(rows[0], rows[1])
    eng.set_rng(seed=0x1234_5678_9ABC_DEF0, draw=41)
    eng.copy_slot(1, 0)
    eng.sample_source(0, 1, objects, None)
    assert np.array_equal(eng.get_source_rows(1, objects), rows[0])


def test_gibbs_source_device_rng_frequencies_follow_the_posterior():
    """Many independent device draws of the same rows: the empirical component frequencies of every
    observation match its posterior row (chi-square over all valid observations, 5 sigma)."""
    from sbayes_amd.operators import calculate_source_posterior, gibbs_sample_source
    fx, model, sample = _gibbs_fixture()
    objects = fx.z["gs_subset_objects"]
    p = calculate_source_posterior(model, sample, objects).astype(np.float64)
    eng = model.likelihood.engine
    eng.set_rng(seed=99)
    n_draws = 400
    freq = np.zeros_like(p)
    for _ in range(n_draws):
        new, _, _ = gibbs_sample_source(model, sample, objects, device_rng=True)
        freq += new.source.value[objects]
    valid = ~fx.na_values[objects]
    assert np.array_equal(freq.sum(-1)[valid], np.full(valid.sum(), n_draws))
    expected = p[valid] * n_draws
    cells = expected > 5
    chi2 = (((freq[valid] - expected) ** 2)[cells] / expected[cells]).sum()
    dof = cells.sum() - valid.sum()
    assert abs(chi2 - dof) < 5 * np.sqrt(2 * dof), (chi2, dof)
    assert np.all(freq[valid][expected == 0] == 0)


def test_bind_cache_sends_only_what_changed_and_never_goes_stale():
    """The operator forms bind the sample into an engine slot on every call; the bind cache skips what the slot
    already holds.  Same results with and without it, through changes of counts, clusters and weights, through
    foreign writes to the slot (any Engine method that changes a slot drops its cache entry), and for writable
    arrays modified in place (compared by content, not identity)."""
    from sbayes_amd.conditionals import _bind_slot, _engine, mixture_log_likelihood
    wl = make_workload("cfg1")
    model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration,
                              wl.weights, wl.source)
    recalculate_feature_counts(model.data.features.values, sample)
    try:
        eng = _engine(model)
        na = ~wl.features.any(-1)

        def want(groups, source, weights):
            counts = orc.recalculate_feature_counts(wl.features, groups, source)
            return orc.mixture_loglik(wl.features, na, groups, counts, wl.concentration, weights)

        ll0 = mixture_log_likelihood(model, sample)
        assert abs(ll0 - want(wl.groups, wl.source, wl.weights)) <= 1e-10 * abs(ll0)
        calls = []
        orig = eng.set_counts
        eng.set_counts = lambda *a, **k: (calls.append("set_counts"), orig(*a, **k))[1]
        assert mixture_log_likelihood(model, sample) == ll0 and calls == []          # nothing re-sent
        # a foreign write to the slot drops the entry: everything is sent again, result unchanged
        eng.set_weights(0, np.full_like(wl.weights, 1.0 / wl.n_components))
        assert mixture_log_likelihood(model, sample) == ll0 and len(calls) == wl.n_components
        # a new state (clusters, source, counts, weights changed) through the same slot
        clusters2, weights2, source2 = make_state(wl.features, wl.groups[1:], wl.clusters.shape[0], seed=77)
        groups2 = [clusters2] + wl.groups[1:]
        model2, sample2 = sbm.build(wl.features, wl.states_per_feature, wl.component_names, groups2, wl.concentration,
                                    weights2, source2)
        recalculate_feature_counts(model2.data.features.values, sample2)
        ll2 = mixture_log_likelihood(model, sample2)
        assert abs(ll2 - want(groups2, source2, weights2)) <= 1e-10 * abs(ll2)
        assert mixture_log_likelihood(model, sample) == ll0                          # and back
        # in-place edits through the parameter API between two binds (unshared parameters edit the SAME ndarray and bump
        # the version: sbayes/sampling/state.py:43-61, 340-350): the cache is keyed on (array, version), never on identity
        fresh_model, fresh = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration,
                                       wl.weights, wl.source)
        recalculate_feature_counts(fresh_model.data.features.values, fresh)
        assert mixture_log_likelihood(model, fresh) == ll0
        w_arr, c_arr = fresh.weights.value, fresh.feature_counts["clusters"].value
        with fresh.weights.edit() as w:
            w[:] = weights2
        diff = np.zeros_like(c_arr)
        diff[0, 0, int(np.flatnonzero(wl.states_per_feature[0])[0])] = 3.0
        fresh.feature_counts["clusters"].add_changes(diff)
        assert fresh.weights.value is w_arr and fresh.feature_counts["clusters"].value is c_arr      # same ndarrays
        counts_now = [fresh.feature_counts[k].value for k in fresh.component_names]
        want_now = orc.mixture_loglik(wl.features, na, wl.groups, counts_now, wl.concentration, weights2)
        got_now = mixture_log_likelihood(model, fresh)
        assert abs(got_now - want_now) <= 1e-10 * abs(want_now) and got_now != ll0
        # unversioned arrays (confounder group matrices, concentration tables) are compared by content
        from sbayes_amd.conditionals import _remember, _same, _token
        w = np.array(wl.weights)
        rec = _remember(_token(w))
        assert _same(_token(w), rec) and _same(_token(w.copy()), rec)
        w[0, 0] += 0.25
        assert not _same(_token(w), rec)
        w.setflags(write=False)
        rec = _remember(_token(w))
        assert _same(_token(w), rec) and not _same(_token(np.zeros_like(w)), rec)
    finally:
        release_all()


# ---- next-heaviest operator expressions (missing #2 / #3) ---------------------------------------------
def _extras_case(name):
    from tests.test_operator_extras_cpu import load_case
    fx, z = load_case(name)
    names = fx.meta["component_names"]
    model, sample = sbm.build(fx.features, fx.states_per_feature, names, fx.groups, fx.conc, fx.weights, fx.source,
                              counts=fx.counts)
    model.prior.prior_cluster_effect.uniform_concentration_array = z["jp_cluster_unif"]
    return fx, z, model, sample


@pytest.mark.parametrize("name", ["south_america", "cfg1", "headline"])
def test_jump_lh_matches_the_reference_operator(name):
    """ClusterJump.get_jump_lh (operators.py:1679-1722) with expected_confounder_features (:1342-1379): the drop-in
    form against the reference operator's own output (float32 products there, sums of fp64 logs here: 2e-5), the
    device's sums of logs against the oracle's (1e-10: same float32 per-feature values, pinned bit for bit to the
    reference's by tests/test_operator_extras_cpu.py)."""
    from sbayes_amd.operators import jump_lh
    from tests.test_operator_extras_cpu import jump_keys
    fx, z, model, sample = _extras_case(name)
    eng = model.likelihood.engine
    unif = z["jp_cluster_unif"]
    for key in jump_keys(z):
        _, tag, s, t = key.split("_")
        temp, ptemp = (1.0, 1.0) if tag == "t1" else (1.3, 1.5)
        i_s, i_t = int(s[1:]), int(t[1:])
        got = jump_lh(model, sample, i_s, i_t, temperature=temp, prior_temperature=ptemp)
        assert got.dtype == np.float32 and got.shape == z[key].shape
        np.testing.assert_allclose(got, z[key], rtol=2e-5, atol=1e-6, err_msg=key)
        # engine level: the two sums of logs
        want = orc.jump_log_lh(fx.features, fx.na_values, fx.groups, fx.counts, fx.conc, unif, fx.weights, i_s, i_t, temp, ptemp)
        kw = dict(temperature=temp, prior_temperature=ptemp, unif_counts=unif)
        tabs = [eng.normalize_tables(fx.counts[0][[k]], fx.conc[0], **kw) for k in (i_s, i_t)]
        pconf = np.concatenate([eng.normalize_tables(fx.counts[c], fx.conc[c], **kw) for c in range(1, fx.n_comp)])
        logs = eng.jump_lh(0, pconf, tabs[0], tabs[1], np.flatnonzero(fx.groups[0][i_s]), ptemp)
        rtol = 1e-10 if tag == "t1" else 2e-6            # (tempered weights go through the device's float32 powf)
        np.testing.assert_allclose(logs, want, rtol=rtol, atol=1e-12, err_msg=key)


@pytest.mark.parametrize("name", ["south_america", "test_files", "cfg1", "headline"])
def test_source_lh_by_feature_matches_the_reference(name):
    """GibbsSampleWeights.source_lh_by_feature (operators.py:677-685): float32 [F] from the device's resident source,
    patterns and weights against the reference's value.  Tolerance: both sides take float32 logs (NumPy's own SIMD
    log there, logf here: ~1 ulp apart) and add them up in float32 in object order; a float32 running sum of N terms
    carries up to N * 2^-24 relative error, and one-ulp differences in the terms move its rounding decisions, so the two
    sums agree to a fraction of that bound, not to float32 epsilon."""
    from sbayes_amd.operators import source_lh_by_feature
    fx, z, model, sample = _extras_case(name)
    got = source_lh_by_feature(model, sample)
    assert got.dtype == np.float32 and got.shape == (fx.features.shape[1],)
    rtol = max(2e-6, 0.5 * fx.features.shape[0] * 2.0 ** -24)
    np.testing.assert_allclose(got, z["swl_lh_by_feature"], rtol=rtol, atol=1e-5)
    # new weights through the same slot (what GibbsSampleWeights._propose does between its two evaluations)
    w2 = np.random.default_rng(4).dirichlet(np.ones(fx.n_comp), size=fx.features.shape[1]).astype(np.float32)
    sample.weights.set_value(w2)
    want = orc.source_lh_by_feature(fx.source, orc.normalize_weights(w2, orc.has_components(fx.groups)), fx.na_values)
    np.testing.assert_allclose(source_lh_by_feature(model, sample), want, rtol=rtol, atol=1e-5)


@pytest.mark.parametrize("tag", ["grow", "shrink", "mc3", "prior"])
def test_cluster_gibbs_sample_source_draw_for_draw(tag):
    """ClusterOperator.gibbs_sample_source (operators.py:796-851) in one engine call (sbe_given_unchanged_gibbs) with the
    reference's own uniforms -> the reference's new source array and counts (always), float32 log_q / log_q_back bit-exact
    at temperature 1 and within the float32 powf tolerance at MC3 temperatures."""
    from tests import _cluster_gibbs_case as case
    try:
        exact, (lq, want_lq), (lqb, want_lqb), sample, objects, fx = case.run_case(tag)
        if exact:
            assert lq == np.float32(want_lq) and lqb == np.float32(want_lqb), (tag, lq, want_lq, lqb, want_lqb)
        else:
            assert abs(lq - want_lq) <= 2e-6 * max(1.0, abs(want_lq)) and abs(lqb - want_lqb) <= 2e-6 * max(1.0, abs(want_lqb))
    finally:
        release_all()

