"""Data-parallel cores of the reference's cluster operators on the engine.

  compute_cluster_posterior   AlterCluster.compute_cluster_posterior      operators.py:1035-1073
  compute_raw_cluster_probs   AlterClusterWide.compute_raw_cluster_probs  operators.py:1420-1472
                              (with the `gibbs` cluster-effect proposal,   operators.py:1254-1282)

The proposal logic, RNG and accept/reject stay the reference's (out of scope); these functions
return exactly what the reference methods return, so an operator can call them in place of its
own method body.  Everything array-sized runs on the device; the O(n_available) scalar tail
(temperature exponent, geo-prior factor, normalisation, smoothing) is host arithmetic as in the
reference.  The per-feature products are accumulated in log space on the device, so unlike the
reference (np.prod) large feature counts do not underflow.
"""
from __future__ import annotations

import numpy as np

from .conditionals import _bind_slot, _engine

EPS = np.finfo(np.float32).eps        # sbayes/util.py:34


def _prepare(model, sample, slot):
    eng = _engine(model)
    _bind_slot(eng, model, sample, slot)
    for c in range(eng.n_components):
        eng.update_probs(slot, c)
    return eng


def _log_marginals(eng, slot, table, available, prior_temperature):
    objects = np.flatnonzero(available) if np.asarray(available).dtype == np.bool_ else np.asarray(available)
    return eng.cluster_marginals(slot, table, objects, prior_temperature)


def compute_cluster_posterior(model, sample, i_cluster, available, temperature=1.0, prior_temperature=1.0,
                              additive_smoothing=1e-6, geo_likelihoods=None, slot=0):
    """Posterior probability of each available object to belong to cluster `i_cluster`."""
    eng = _prepare(model, sample, slot)
    prior = model.prior.prior_cluster_effect
    table = eng.normalize_tables(
        sample.feature_counts["clusters"].value[[i_cluster]], np.asarray(prior.concentration_array),
        temperature=temperature, prior_temperature=prior_temperature,
        unif_counts=np.asarray(prior.uniform_concentration_array))
    log_m = _log_marginals(eng, slot, table, available, prior_temperature) / temperature
    if geo_likelihoods is not None:
        log_m[1] += np.log(geo_likelihoods)
    posterior = 1.0 / (1.0 + np.exp(log_m[0] - log_m[1]))        # m1 / (m0 + m1)
    if additive_smoothing > 0:
        posterior = (posterior + additive_smoothing) / (1 + 2 * additive_smoothing)
    return posterior


def compute_raw_cluster_probs(model, sample, i_cluster, available, temperature=1.0, prior_temperature=1.0,
                              geo_prior_ratio=None, slot=0):
    """AlterClusterWide.compute_raw_cluster_probs with ClusterEffectProposals.gibbs."""
    eng = _prepare(model, sample, slot)
    prior = model.prior.prior_cluster_effect
    table = eng.normalize_tables(
        sample.feature_counts["clusters"].value[[i_cluster]], np.asarray(prior.concentration_array),
        temperature=temperature, prior_temperature=prior_temperature,
        unif_counts=np.asarray(prior.uniform_concentration_array))
    if temperature != 1.0:
        table = table ** (1 / temperature)          # inner1d(features, p) ** (1/T): elementwise on the table
    log_m = _log_marginals(eng, slot, table, available, prior_temperature) / temperature
    m = np.exp(log_m)
    if geo_prior_ratio is not None:
        m[1] *= geo_prior_ratio
    return m[1] / (m[0] + m[1] + EPS)


def calculate_source_posterior(model, sample, object_subset, temperature=1.0, prior_temperature=1.0, slot=0):
    """GibbsSampleSource.calculate_source_posterior (operators.py:554-574): float32
    [n_subset, F, C] posterior of the source assignment of every observation of the subset."""
    eng = _prepare(model, sample, slot)
    if isinstance(object_subset, slice):
        objects = np.arange(sample.n_objects)[object_subset]
    else:
        objects = np.asarray(object_subset)
        if objects.dtype == np.bool_:
            objects = np.flatnonzero(objects)
    return eng.source_posterior(slot, objects, temperature, prior_temperature)


def component_likelihood_given_unchanged(model, sample, object_subset, i_cluster, temperature=1.0,
                                         prior_temperature=1.0):
    """operators.py:863-928: float32 [n_subset, F, C] component likelihoods of the subset's
    observations under effect tables built only from the observations that are NOT resampled.
    `object_subset` is a bool mask [n_objects].  Counts (a9) and tables (a10) come from the device."""
    eng = _engine(model)
    object_subset = np.asarray(object_subset, dtype=bool)
    objects = np.flatnonzero(object_subset)
    source = sample.source.value
    prior = model.prior.prior_cluster_effect
    cluster = sample.clusters.value[i_cluster]
    kept = eng.effect_counts((cluster & ~object_subset)[None, :], source[..., 0])
    tables = [eng.normalize_tables(kept, np.asarray(prior.concentration_array), temperature=temperature,
                                   prior_temperature=prior_temperature,
                                   unif_counts=np.asarray(prior.uniform_concentration_array))]
    group_idx = [np.zeros(objects.size, dtype=np.int32)]            # every subset object sees the cluster table
    for i_conf, conf in enumerate(sample.confounders, start=1):
        conf_prior = model.prior.prior_confounding_effects[conf]
        groups = sample.confounders[conf].group_assignment
        changeable = eng.effect_counts(groups & object_subset[None, :], source[..., i_conf])
        unchangeable = sample.feature_counts[conf].value - changeable
        if conf_prior.any_dynamic_priors:
            prior_counts = conf_prior.concentration_array_given_unchanged(sample, changed_objects=object_subset)
        else:
            prior_counts = conf_prior.concentration_array(sample)
        tables.append(eng.normalize_tables(unchangeable, np.asarray(prior_counts), temperature=temperature,
                                           prior_temperature=prior_temperature,
                                           unif_counts=np.asarray(conf_prior.uniform_concentration_array)))
        sub = groups[:, object_subset]
        group_idx.append(np.where(sub.any(axis=0), sub.argmax(axis=0), -1).astype(np.int32))
    return eng.subset_lh(objects, tables, np.stack(group_idx), temperature)
