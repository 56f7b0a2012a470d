"""Drop-in for the likelihood part of sbayes/sampling/conditionals.py
plus the fused north-star evaluation.

  likelihood_per_component(model, sample, caching=True)   conditionals.py:152-223
  likelihood_per_component_exact(model, sample)           conditionals.py:300-367
  conditional_effect_mean(...)                            conditionals.py:105-122
  mixture_log_likelihood(model, sample)                   SURVEY.md 8(d): the composition
        np.log(np.sum(update_weights(s) * likelihood_per_component(model, s), -1))[~na].sum()
        evaluated by ONE fused kernel with everything resident on the device.
"""
from __future__ import annotations

import numpy as np

from . import _fast
from .binding import (_bind_slot, _bind_uniform, _engine, _remember, _same, _tables_current,      # noqa: F401
                      _token)                                                                  # (re-exported)
from .likelihood import LazyBlock, compute_component_likelihood
from .registry import get_engine


def conditional_effect_mean(prior_counts, feature_counts, unif_counts=None, prior_temperature=None,
                            temperature=None, features=None, engine=None):
    """normalize(feature_counts/T + unif + (prior - unif)/T_prior) on the device.  The engine is
    found through `features` (or given); tables are [n_groups, n_features, n_states]."""
    eng = engine if engine is not None else get_engine(features)
    feature_counts = np.asarray(feature_counts)
    prior_counts = np.asarray(prior_counts, dtype=np.float64)
    if prior_counts.shape != feature_counts.shape:
        prior_counts = np.broadcast_to(prior_counts, feature_counts.shape)
    if prior_temperature is not None:
        assert unif_counts is not None
    return eng.normalize_tables(feature_counts, prior_counts, temperature=temperature,
                                prior_temperature=prior_temperature, unif_counts=unif_counts)


def likelihood_per_component(model, sample, caching=True):
    """float64 [n_objects, n_features, n_components] component likelihoods, cached in
    sample.cache.component_likelihoods with the reference's partial-update semantics."""
    features = model.data.features
    confounders = model.data.confounders
    feature_counts = sample.feature_counts
    eng = _engine(model)

    cache = sample.cache.component_likelihoods
    if caching and not _fast.node_outdated(cache):
        return cache.value
    if type(cache._value) is LazyBlock:               # (likelihood.LazyBlock: the block was never computed -- now it is)
        cache._value = cache._value.materialize()

    with cache.edit() as component_likelihood:
        changed_clusters = cache.what_changed(input_key=["clusters", "clusters_counts"], caching=caching)
        if len(changed_clusters) > 0:
            cluster_effect = eng.normalize_tables(
                feature_counts["clusters"].value, np.asarray(model.prior.prior_cluster_effect.concentration_array))
            compute_component_likelihood(
                features=features.values, probs=cluster_effect, groups=sample.clusters.value,
                changed_groups=changed_clusters, out=component_likelihood[..., 0], na_value=1.0)

        for i, conf in enumerate(confounders.keys(), start=1):
            conf_prior = model.prior.prior_confounding_effects[conf]
            hyperprior_has_changed = conf_prior.any_dynamic_priors and cache.ahead_of("universal_counts")
            changed_groups = cache.what_changed(input_key=f"{conf}_counts",
                                                caching=caching and not hyperprior_has_changed)
            if len(changed_groups) == 0:
                continue
            conf_effect = eng.normalize_tables(feature_counts[conf].value,
                                               np.asarray(conf_prior.concentration_array(sample)))
            compute_component_likelihood(
                features=features.values, probs=conf_effect, groups=confounders[conf].group_assignment,
                changed_groups=changed_groups, out=component_likelihood[..., i], na_value=1.0)
        # conditionals.py:216 sets NA observations to 1 here; the device has already written 1 for the NA
        # observations of every row it touched, and untouched rows carry it from the call that wrote them
    return cache.value


# The reference's `likelihood_per_component_subset` (conditionals.py:226-297) is a line-for-line duplicate of
# `likelihood_per_component` with no caller in the reference; it is served by the same function.
likelihood_per_component_subset = likelihood_per_component


def likelihood_per_component_exact(model, sample, slot=0):
    """Leave-one-out component likelihoods (used by the reference's LikelihoodLogger)."""
    eng = _engine(model)
    _bind_slot(eng, model, sample, slot, with_source=True)
    return eng.likelihood_per_component_exact(slot)


def mixture_log_likelihood(model, sample, slot=0) -> float:
    """One uncached eval of the marginal mixture log-likelihood by the fused kernel."""
    eng = _engine(model)
    _bind_slot(eng, model, sample, slot)
    _tables_current(eng, slot)
    return eng.mixture_loglik(slot)


def observation_likelihoods(model, sample, slot=0, exact=False):
    """float64 [n_objects, n_features]: sum_c w * lh per observation (loggers.py:355-357)."""
    eng = _engine(model)
    if exact:
        _bind_slot(eng, model, sample, slot, with_source=True)
        return eng.observation_lh_exact(slot)
    _bind_slot(eng, model, sample, slot)
    _tables_current(eng, slot)
    return eng.observation_lh(slot)


# True while patch.install(operators=True) serves SourcePrior.__call__ from the device: Likelihood.__call__ then takes the
# source prior of the sample along with its collapsed likelihood (one launch, store_source_prior_ahead below)
DEVICE_SOURCE_PRIOR = False


def _store_source_prior(cache, sample, values, caching):
    """SourcePrior.__call__'s cache update (sbayes/model/prior.py:596-609): everything when the weights changed, else the
    objects whose source rows changed; `values` is the float64 [n_objects] vector or a callable returning it, asked only
    when something is listed."""
    if _fast._h is not None:
        # the same steps in one native call (csrc/sbe_pyhost.c: store_per_object); NotImplemented: a node form it does not serve
        if _fast._h.store_per_object(cache, sample.n_objects, values, caching) is None:
            return
    with cache.edit() as per_object:
        if cache.ahead_of("weights"):
            if sample.n_objects > 0:
                per_object[:] = values() if callable(values) else values    # (the reference's changed = arange(n_objects): every object)
        else:
            # the reference asks what_changed(input_key=["source"]): a one-key list, whose result is np.unique of the one-key
            # answer -- already ascending and duplicate-free (flatnonzero / arange, state.py:239-254); asking with the key itself
            # gives the same int64 array without the concatenate + sort (28 us of a 61 us call in place, tools/host_residual.py)
            changed = cache.what_changed("source", caching=caching)
            if len(changed) > 0:
                per_object[changed] = (values() if callable(values) else values)[changed]


def source_prior_wanted(prior, sample):
    """The sample's source-prior cache node if the prior's next evaluation of this sample would have to ask the device
    for it (Model.__call__ = likelihood, then prior: sbayes/model/model.py:47-51), else None."""
    if not DEVICE_SOURCE_PRIOR:
        return None
    sp = getattr(prior, "source_prior", None)
    if sp is None or getattr(sp, "_sbayes_amd_owner", None) is None:
        return None
    cache = getattr(sample.cache, "source_prior", None)
    if cache is None or getattr(sample, "source", None) is None or not _fast.node_outdated(cache):
        return None
    return cache


def store_source_prior_ahead(cache, sample, per_object_values):
    """The per-object values came back with the collapsed likelihood (Engine.collapsed_and_source_prior): the cache
    node is updated now, by the protocol SourcePrior.__call__ would follow a moment later -- which then finds it current."""
    _store_source_prior(cache, sample, per_object_values, True)


def source_prior(model, sample, slot=0, caching=True) -> float:
    """SourcePrior.__call__ (sbayes/model/prior.py:573-611): log prior of the source assignment given the weights, with
    the reference's per-object cache protocol -- everything when the weights changed (`cache.ahead_of("weights")`),
    else the objects whose source rows changed (`what_changed("source")`) -- and the per-object values from the device:
    the sample is bound with its source (changed rows only), N doubles come back."""
    cache = getattr(sample.cache, "source_prior", None)
    if cache is None:                                 # a sample type without that cache node
        eng = _engine(model)
        _bind_slot(eng, model, sample, slot, with_source=True)
        return eng.source_prior(slot).sum()
    if caching and not _fast.node_outdated(cache):
        return cache.value.sum()

    def values():
        eng = _engine(model)
        _bind_slot(eng, model, sample, slot, with_source=True)
        return eng.source_prior(slot)
    _store_source_prior(cache, sample, values, caching)
    return cache.value.sum()
