"""Minimal Model / Data / Prior containers with the attribute names the likelihood path reads
(sbayes/model/model.py:24-51, prior.py:325-354, 453-455).  Used where real sBayes is not
importable (GPU box, replay driver, tests); with real sBayes its own objects are used."""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

from .likelihood import Likelihood
from .state import Confounder, Features, ModelShapes, Sample


class ClusterEffectPrior:
    def __init__(self, concentration_array, uniform_concentration_array=None):
        self.concentration_array = np.asarray(concentration_array, dtype=np.float64)
        # symmetric concentration 1.0 on applicable states (prior.py:184-186)
        self.uniform_concentration_array = (
            np.asarray(uniform_concentration_array, dtype=np.float64) if uniform_concentration_array is not None
            else (self.concentration_array > 0).astype(np.float64))


class ConfoundingEffectsPrior:
    any_dynamic_priors = False

    def __init__(self, concentration_array, uniform_concentration_array=None):
        self._concentration_array = np.asarray(concentration_array, dtype=np.float64)
        self.uniform_concentration_array = (
            np.asarray(uniform_concentration_array, dtype=np.float64) if uniform_concentration_array is not None
            else (self._concentration_array[0] > 0).astype(np.float64))

    def concentration_array(self, sample=None):
        return self._concentration_array


class UniversalConfoundingEffectsPrior(ConfoundingEffectsPrior):
    """The reference's DYNAMIC confounding-effects prior (group prior type `universal`, sbayes/model/prior.py:309-315):
    the concentration of every group follows the posterior mean of the `universal` confounder's single group -- it is
    a function of the SAMPLE (`any_dynamic_priors = True`), rewritten in place on every request (prior.py:325-354),
    with the leave-subset-out form of component_likelihood_given_unchanged (prior.py:356-387).  The reference's config
    layer still rejects this type ("not implemented yet", config/config.py:226-232), so the branches it switches on in
    the likelihood path (likelihood.py:92-93, conditionals.py:197-200, operators.py:912-915) are reachable only with
    a hand-built prior: this mirror exists so that they are driven at all (tests/test_dynamic_priors_*.py)."""
    any_dynamic_priors = True

    def __init__(self, n_groups, states_per_feature, precision, universal_prior, features):
        states = np.asarray(states_per_feature, dtype=bool)
        super().__init__(np.broadcast_to(states.astype(np.float64), (n_groups,) + states.shape).copy())
        self.states_per_feature = states
        self.precision = float(precision)
        self.universal_prior = universal_prior          # the `universal` confounder's (static) prior
        self.features = np.asarray(features, dtype=bool)

    def _fill(self, univ_counts):
        with np.errstate(invalid="ignore"):
            mean = univ_counts / univ_counts.sum(axis=-1, keepdims=True)                       # normalize (util.py:990-1007)
        mean = mean.astype(np.float32)
        uniform = (self.states_per_feature / self.states_per_feature.sum(axis=-1, keepdims=True)).astype(np.float32)
        mean = 0.95 * mean + 0.05 * uniform                                                     # prior.py:337-338
        precision = self.precision * self.states_per_feature.sum(axis=-1)[:, np.newaxis]        # prior.py:347
        self._concentration_array[:] = mean * precision                                        # in place, every group
        return self._concentration_array

    def concentration_array(self, sample=None):
        univ = self.universal_prior.concentration_array(sample)[0] + sample.feature_counts["universal"].value[0]
        return self._fill(univ)

    def concentration_array_given_unchanged(self, sample, changed_objects):
        univ = self.universal_prior.concentration_array(sample)[0] + sample.feature_counts["universal"].value[0]
        # (the reference subtracts the subset's observations whose source is component 0 here, prior.py:363-364)
        changeable = np.sum(sample.source.value[changed_objects, :, 0, None] * self.features[changed_objects, :, :], axis=0)
        return self._fill(univ - changeable)


class Prior:
    def __init__(self, cluster_concentration, confounder_concentrations: dict):
        self.prior_cluster_effect = ClusterEffectPrior(cluster_concentration)
        self.prior_confounding_effects = {k: (v if isinstance(v, ConfoundingEffectsPrior) else ConfoundingEffectsPrior(v))
                                          for k, v in confounder_concentrations.items()}


class Data:
    def __init__(self, features, confounders):
        self.features = features
        self.confounders = confounders


class Model:
    def __init__(self, data, n_clusters, prior):
        self.data = data
        self.confounders = data.confounders
        self.n_clusters = n_clusters
        n, f, s = data.features.values.shape
        self.shapes = ModelShapes(
            n_clusters=n_clusters, n_sites=n, n_features=f, n_states=s,
            states_per_feature=data.features.states, n_confounders=len(data.confounders),
            n_groups={k: c.n_groups for k, c in data.confounders.items()})
        self.prior = prior
        self.likelihood = Likelihood(data=data, shapes=self.shapes, prior=prior)


def build(features, states_per_feature, component_names, groups, concentration, weights, source, counts=None):
    """(model, sample) from plain arrays (fixtures / synthetic workloads).  counts=None leaves
    zero counts: call recalculate_feature_counts(model.data.features.values, sample)."""
    feats = Features(features, states=states_per_feature)
    confounders = OrderedDict((name, Confounder(name, g)) for name, g in zip(component_names[1:], groups[1:]))
    prior = Prior(concentration[0], dict(zip(component_names[1:], concentration[1:])))
    model = Model(Data(feats, confounders), n_clusters=groups[0].shape[0], prior=prior)
    n, f, s = feats.values.shape
    if counts is None:
        counts = [np.zeros((g.shape[0], f, s), dtype=np.float32) for g in groups]
    sample = Sample.from_numpy_arrays(
        clusters=groups[0], weights=weights, confounders=confounders, source=source,
        feature_counts=dict(zip(component_names, counts)), model_shapes=model.shapes)
    return model, sample
