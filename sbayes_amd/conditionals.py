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

from .likelihood import compute_component_likelihood
from .registry import get_engine


def _engine(model):
    lik = getattr(model, "likelihood", None)
    if lik is not None and hasattr(lik, "engine"):
        return lik.engine
    n_groups = [model.shapes.n_clusters] + [c.n_groups for c in model.data.confounders.values()]
    return get_engine(model.data.features.values, n_groups)


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
    if caching and not cache.is_outdated():
        return cache.value

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


def _token(param):
    """(array, version) of a state parameter.  The reference's (and the mirror's) parameters bump `version` on every
    edit -- set_value, set_items, edit(), edit_group(s), set_groups, FeatureCounts.add_changes
    (sbayes/sampling/state.py:34-61, 97-161, 340-350) -- and edit the SAME ndarray in place whenever the parameter is
    not shared with a copy, so array identity alone says nothing; plain arrays (confounder group matrices,
    concentration tables) have no version and are compared by content."""
    value = getattr(param, "value", param)
    return np.asarray(value), getattr(param, "version", None)


def _same(tok, cached):
    """True if the token `tok` = (array, version) denotes what `cached` = (array, version, private copy) recorded.
    Versioned parameters: same ndarray object AND same version (an in-place edit through the parameter API always
    bumps the version; a copy-on-write edit always creates a new ndarray).  Unversioned arrays: content equality
    against the private copy -- identity is never trusted."""
    if cached is None:
        return False
    arr, version = tok
    ref, ref_version, copy = cached
    if version is not None and ref_version is not None:
        return arr is ref and version == ref_version
    return copy is not None and arr.shape == copy.shape and arr.dtype == copy.dtype and np.array_equal(arr, copy)


def _remember(tok):
    arr, version = tok
    # the ndarray itself is kept alive so that its id cannot be recycled for another array while the entry lives
    return arr, version, (None if version is not None else arr.copy())


def _bind_slot(eng, model, sample, slot, with_source=False):
    """Upload one sample's state into an engine slot: group ids, counts, weights (small), optionally the source
    assignment.  Only what differs from what the slot was last bound to is sent (the operators evaluate the same or
    nearly the same sample many times in a row); returns the components whose probability tables are stale."""
    names = sample.component_names
    C = len(names)
    groups = [_token(sample.clusters)] + [_token(c.group_assignment) for c in sample.confounders.values()]
    conc = [_token(model.prior.prior_cluster_effect.concentration_array)] + [
        _token(model.prior.prior_confounding_effects[k].concentration_array(sample)) for k in names[1:]]
    counts = [_token(sample.feature_counts[name]) for name in names]
    weights = _token(sample.weights)
    source = _token(sample.source) if with_source else None
    cache = getattr(eng, "_bound", None)
    if cache is None:                               # an engine without a bind cache (test doubles): send everything
        for c in range(C):
            eng.set_groups(slot, c, groups[c][0])
            eng.set_concentration(c, conc[c][0])
            eng.set_counts(slot, c, counts[c][0])
        if with_source:
            eng.set_source(slot, source[0])
        eng.set_weights(slot, weights[0])
        return set(range(C))
    old = cache.get(slot) or {"groups": [None] * C, "counts": [None] * C, "weights": None, "source": None, "stale": set(range(C))}
    new = {"groups": list(old["groups"]), "counts": list(old["counts"]), "weights": old["weights"], "source": old["source"],
           "stale": set(old["stale"])}
    conc_changed = [c for c in range(C) if not _same(conc[c], eng._bound_conc.get(c))]
    for c in conc_changed:                          # (drops every slot's entry: all tables depend on it)
        eng.set_concentration(c, conc[c][0])
    if conc_changed:
        new["stale"] = set(range(C))
    for c in range(C):
        if not _same(groups[c], old["groups"][c]):
            eng.set_groups(slot, c, groups[c][0])
            new["groups"][c] = _remember(groups[c])
        if not _same(counts[c], old["counts"][c]):
            eng.set_counts(slot, c, counts[c][0])
            new["counts"][c] = _remember(counts[c])
            new["stale"].add(c)
    if with_source and not _same(source, old["source"]):
        eng.set_source(slot, source[0])
        new["source"] = _remember(source)
    if not _same(weights, old["weights"]):
        eng.set_weights(slot, weights[0])
        new["weights"] = _remember(weights)
    for c in conc_changed:
        eng._bound_conc[c] = _remember(conc[c])
    cache[slot] = new                               # (the setters above dropped the slot's entry)
    return new["stale"]


def _tables_current(eng, slot):
    """Rebuild the probability tables that the last _bind_slot left stale."""
    entry = getattr(eng, "_bound", {}).get(slot)
    stale = set(range(eng.n_components)) if entry is None else entry["stale"]
    for c in sorted(stale):
        eng.update_probs(slot, c)
    if entry is not None:
        entry["stale"] = set()


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


def source_prior(model, sample, slot=0, caching=True) -> float:
    """SourcePrior.__call__ (sbayes/model/prior.py:573-611): log prior of the source assignment given
    the weights, cached per object in sample.cache.source_prior when the sample has that node."""
    eng = _engine(model)
    cache = getattr(sample.cache, "source_prior", None)
    if cache is not None and caching and not cache.is_outdated():
        return cache.value.sum()
    _bind_slot(eng, model, sample, slot, with_source=True)
    per_object = eng.source_prior(slot)
    if cache is not None:
        with cache.edit() as arr:
            arr[:] = per_object
        return cache.value.sum()
    return per_object.sum()
