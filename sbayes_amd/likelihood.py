"""Drop-in for sbayes/model/likelihood.py on the MI355X engine.

Same public surface as the reference module:
  class Likelihood(data, shapes, prior)       likelihood.py:22-101
  compute_component_likelihood(...)           likelihood.py:104-133
  compute_component_likelihood_exact(...)     likelihood.py:136-150
  update_weights(sample, caching=True)        likelihood.py:153-168
  normalize_weights(weights, has_components)  likelihood.py:171-190
Control flow (what is cached, what `changed_groups` is) is host Python exactly as in the
reference; every array operation runs on the GPU through the C ABI.  There is no CPU
fallback: without the HIP library / a GPU these functions raise.
"""
from __future__ import annotations

import numpy as np

from .counts import recalculate_feature_counts
from . import registry
from .registry import get_engine


class Likelihood:
    """Collapsed Dirichlet-categorical likelihood of the sBayes mixture model."""

    def __init__(self, data, shapes, prior):
        self.features = data.features.values
        self.confounders = data.confounders
        self.shapes = shapes
        self.prior = prior
        self.source_index = {"clusters": 0}
        for i, conf in enumerate(self.confounders, start=1):
            self.source_index[conf] = i
        self._na_features = None
        registry.note_features(self.features, self.n_groups)

    # device handles are per process: never pickled, re-created lazily (mcmc_setup.py:299, model.py:53)
    def __getstate__(self):
        state = dict(self.__dict__)
        state["_na_features"] = None
        return state

    @property
    def n_groups(self):
        return [self.shapes.n_clusters] + [conf.n_groups for conf in self.confounders.values()]

    @property
    def engine(self):
        return get_engine(self.features, self.n_groups)

    @property
    def na_features(self):
        """bool [n_objects, n_features]: observations with no state set (likelihood.py:40)."""
        if self._na_features is None:
            self._na_features = self.engine.na_values()
        return self._na_features

    def __call__(self, sample, caching=True) -> float:
        if not caching:
            recalculate_feature_counts(self.features, sample)
        log_lh = 0.0
        log_lh += self.compute_lh_clusters(sample, caching=caching)
        for conf in self.confounders:
            log_lh += self.compute_lh_confounder(sample, conf, caching=caching)
        return log_lh

    def _group_logliks(self, counts, concentration, groups):
        """float64 per listed group: float32-summed Dirichlet-categorical log-pdf (device)."""
        groups = np.asarray(groups)
        conc = concentration if concentration.ndim == 2 else concentration[groups]
        _, per_group = self.engine.dirichlet_logpdf(counts[groups], conc, per_group=True)
        return per_group

    def compute_lh_clusters(self, sample, caching=True) -> float:
        cache = sample.cache.group_likelihoods["clusters"]
        feature_counts = sample.feature_counts["clusters"].value
        with cache.edit() as lh:
            changed = cache.what_changed("counts", caching=caching)
            if len(changed) > 0:
                lh[changed] = self._group_logliks(
                    feature_counts, np.asarray(self.prior.prior_cluster_effect.concentration_array), changed)
        return cache.value.sum()

    def compute_lh_confounder(self, sample, conf, caching=True) -> float:
        cache = sample.cache.group_likelihoods[conf]
        feature_counts = sample.feature_counts[conf].value
        conf_prior = self.prior.prior_confounding_effects[conf]
        with cache.edit() as lh:
            prior_concentration = conf_prior.concentration_array(sample)
            hyperprior_has_changed = conf_prior.any_dynamic_priors and cache.ahead_of("universal_counts")
            changed = cache.what_changed("counts", caching=caching and not hyperprior_has_changed)
            if len(changed) > 0:
                lh[changed] = self._group_logliks(feature_counts, np.asarray(prior_concentration), changed)
        return cache.value.sum()


def compute_component_likelihood(features, probs, groups, changed_groups, out, na_value=0.0):
    """In-place partial update of `out` [n_objects, n_features] (float64, any strides) with the
    likelihood of every observation under one mixture component; returns `out`.  `na_value` (not in
    the reference signature, default = the reference's result 0) lets likelihood_per_component have
    the device write its NA value 1 directly."""
    return get_engine(features).component_lh(probs, groups, changed_groups, out, na_value)


def compute_component_likelihood_exact(features, probs, groups, changed_groups, out):
    """Leave-one-out form: probs[i] is a per-member table [n_in_group, F, S].  Each member's row
    is one a1 gather with its own table; served by the same device kernel, one group at a time."""
    eng = get_engine(features)
    groups = np.asarray(groups)
    out[~groups.any(axis=0), :] = 0.0
    n_obj = groups.shape[1]
    for i in changed_groups:
        members = np.flatnonzero(groups[i])
        if members.size == 0:
            continue
        tables = np.asarray(probs[i])
        # pseudo-groups: one per member (its own table) + one unchanged group holding everyone
        # else, so that the kernel's "no group -> 0" rule leaves the other rows untouched
        pseudo = np.zeros((members.size + 1, n_obj), dtype=bool)
        pseudo[np.arange(members.size), members] = True
        pseudo[-1] = ~groups[i]
        tables = np.concatenate([tables, np.zeros((1,) + tables.shape[1:], dtype=tables.dtype)])
        eng.component_lh(tables, pseudo, np.arange(members.size), out)
    return out


def normalize_weights(weights, has_components, features=None):
    """float32 [n_rows, n_features, n_components]: weights masked by has_components and renormalised over
    components; has_components may have any number of rows (the reference also calls it with
    has_components[available], operators.py:1086).  `features` (optional) selects the engine explicitly."""
    has_components = np.asarray(has_components)
    if features is not None:
        eng = get_engine(features)
    else:
        eng = registry.engine_for_features(np.shape(weights)[0])
    return eng.normalize_weights(weights, has_components)


def update_weights(sample, caching=True, features=None):
    """Normalised mixture weights of `sample`, cached on (has_components, weights) versions."""
    cache = sample.cache.weights_normalized
    if (not caching) or cache.is_outdated():
        cache.update_value(normalize_weights(sample.weights.value, sample.cache.has_components.value, features))
    return cache.value
