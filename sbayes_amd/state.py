"""Host-side boundary types of the likelihood path.

When real sBayes is importable the engine reads the reference's own `Sample` /
`CacheNode` objects (sbayes/sampling/state.py:87-634) by duck typing.  On a machine
without sBayes (the GPU box, the trace-replay driver, the tests) this module provides
the same carriers: versioned parameters whose per-group version stamps drive
`changed_groups`, cache nodes keyed on the versions of their inputs, and a `Sample`
with copy-on-write parameter sharing.  Only the API the likelihood path touches is
implemented; names and semantics follow the reference so the drop-in classes in
`likelihood.py` / `conditionals.py` / `counts.py` work on either.
"""
from __future__ import annotations

from collections import OrderedDict
from contextlib import contextmanager
from copy import copy as _shallow

import numpy as np

FLOAT_TYPE = np.float32     # sbayes/util.py:32


class ArrayParameter:
    """A read-only ndarray with a monotonically increasing `version` (state.py:22-84).
    `copy()` is O(1): both copies share the buffer until one of them is edited."""

    def __init__(self, value, shared=False):
        self._value = value
        self._value.flags.writeable = False
        self.version = 0
        self.shared = shared

    @property
    def value(self):
        return self._value

    @property
    def shape(self):
        return self._value.shape

    def _own(self):
        if self.shared:
            self.resolve_sharing()

    def resolve_sharing(self):
        self._value = self._value.copy()
        self.shared = False

    def _bump(self):
        self._value.flags.writeable = False
        self.version += 1

    def set_value(self, new_value):
        self._value = new_value
        self.shared = False
        self._bump()

    def set_items(self, keys, values):
        self._own()
        self._value.flags.writeable = True
        self._value[keys] = values
        self._bump()

    @contextmanager
    def edit(self):
        self._own()
        self._value.flags.writeable = True
        yield self._value
        self._bump()
        self.tidy()

    def tidy(self):
        pass

    def copy(self):
        self.shared = True
        return _shallow(self)


class GroupedParameters(ArrayParameter):
    """ArrayParameter whose leading (or `group_dim`) axis enumerates groups; remembers for each
    group the version at which it last changed (state.py:87-171)."""

    def __init__(self, value, group_dim=0):
        super().__init__(value)
        self.group_dim = group_dim
        self.group_versions = np.zeros(self.n_groups)

    @property
    def n_groups(self):
        return self.shape[self.group_dim]

    def get_group(self, i):
        index = [slice(None)] * self._value.ndim
        index[self.group_dim] = i
        return self._value[tuple(index)]

    def resolve_sharing(self):
        self.group_versions = self.group_versions.copy()
        super().resolve_sharing()

    def _stamp_all(self):
        self.group_versions = np.full_like(self.group_versions, self.version)

    def tidy(self):                         # a whole-array edit() touches every group
        self._stamp_all()

    def set_value(self, new_value):
        super().set_value(new_value)
        self._stamp_all()

    def set_items(self, keys, values):
        super().set_items(keys, values)
        if isinstance(keys, int):
            self.group_versions[keys] = self.version
        elif isinstance(keys, tuple):
            self.group_versions[keys[self.group_dim]] = self.version
        else:
            raise RuntimeError("`set_items` needs an int or tuple key for GroupedParameters")

    @contextmanager
    def edit_group(self, i):
        self._own()
        self._value.flags.writeable = True
        yield self.get_group(i)
        self._bump()
        self.group_versions[i] = self.version

    def set_group(self, i, values):
        with self.edit_group(i) as g:
            g[...] = values

    @contextmanager
    def edit_groups(self, idxs):
        if self.group_dim != 0:
            raise NotImplementedError
        self._own()
        self._value.flags.writeable = True
        yield self._value[idxs]
        self._bump()
        self.group_versions[idxs] = self.version
        self.tidy()

    def set_groups(self, group_idxs, new_values):
        if self.group_dim != 0:
            raise NotImplementedError
        self._own()
        self._value.flags.writeable = True
        self._value[group_idxs] = new_values
        self._bump()
        self.group_versions[group_idxs] = self.version


class Clusters(GroupedParameters):
    """bool [n_clusters, n_objects] (state.py:174-209)."""

    edit_cluster = GroupedParameters.edit_group

    @property
    def sizes(self):
        return np.count_nonzero(self._value, axis=1)

    @property
    def n_clusters(self):
        return self.shape[0]

    @property
    def n_objects(self):
        return self.shape[1]

    def any_cluster(self):
        return np.any(self._value, axis=0)

    def add_object(self, i_cluster, i_object):
        with self.edit_cluster(i_cluster) as c:
            c[i_object] = True

    def remove_object(self, i_cluster, i_object):
        with self.edit_cluster(i_cluster) as c:
            c[i_object] = False


class FeatureCounts(GroupedParameters):
    """float32 [n_groups, n_features, n_states] holding integers (state.py:324-350)."""

    @property
    def n_features(self):
        return self.shape[1]

    @property
    def n_states(self):
        return self.shape[2]

    def add_changes(self, diff):
        self._own()
        self._value.flags.writeable = True
        self._value += diff
        self._bump()
        self.group_versions[np.any(diff != 0, axis=(1, 2))] = self.version

    def add_changes_rows(self, group_idx, rows):
        """add_changes(diff) for a diff that is zero outside the rows `group_idx` (distinct): the same values, version and
        group versions, without the dense [n_groups, F, S] operand (the drop-in update_feature_counts knows the rows)."""
        self._own()
        if len(group_idx):
            self._value.flags.writeable = True
            self._value[group_idx] += rows
        self._bump()
        if len(group_idx):
            self.group_versions[group_idx[np.any(rows != 0, axis=(1, 2))]] = self.version


class Confounder:
    """Static assignment of objects to the groups of one confounder (load_data.py:138-184)."""

    def __init__(self, name, group_assignment, group_names=None, has_universal_prior=False):
        self.name = name
        self.group_assignment = np.asarray(group_assignment, dtype=bool)
        self.group_names = list(group_names) if group_names is not None else [
            f"g{i}" for i in range(self.group_assignment.shape[0])]
        self.has_universal_prior = has_universal_prior

    def any_group(self):
        return np.any(self.group_assignment, axis=0)

    @property
    def n_groups(self):
        return len(self.group_names)


class Features:
    """One-hot feature block + NA mask (load_data.py:85-135)."""

    def __init__(self, values, states=None, names=None):
        self.values = np.asarray(values, dtype=bool)
        self.states = states if states is not None else np.ones(self.values.shape[1:], dtype=bool)
        self.names = names if names is not None else np.array([f"F{i}" for i in range(self.values.shape[1])])
        self.na_values = np.sum(self.values, axis=-1) == 0

    n_objects = property(lambda self: self.values.shape[0])
    n_features = property(lambda self: self.values.shape[1])
    n_states = property(lambda self: self.values.shape[2])


class ModelShapes:
    """sbayes/model/model_shapes.py:8-31."""

    def __init__(self, n_clusters, n_sites, n_features, n_states, states_per_feature, n_confounders, n_groups):
        self.n_clusters, self.n_sites, self.n_features, self.n_states = n_clusters, n_sites, n_features, n_states
        self.states_per_feature = states_per_feature
        self.n_confounders = n_confounders
        self.n_groups = n_groups

    @property
    def n_components(self):
        return self.n_confounders + 1

    def __getitem__(self, key):
        return getattr(self, key)


_OUTDATED = -1


class CacheNode:
    """A cached value plus the input versions it was computed from (state.py:218-321)."""

    def __init__(self, value):
        self._value = value
        self.inputs = OrderedDict()
        self.input_idx = OrderedDict()
        self.cached_version = ()
        self.cached_group_versions = {}

    # -- wiring ---------------------------------------------------------------------------------
    def add_input(self, key, inpt):
        self.input_idx[key] = len(self.inputs)
        self.inputs[key] = inpt
        self.cached_version = self.outdated_version()
        if isinstance(inpt, GroupedParameters):
            self.clear_group_version(key)

    def outdated_version(self):
        return (_OUTDATED,) * len(self.inputs)

    def clear_group_version(self, key):
        stale = np.full(self.inputs[key].group_versions.shape, float(_OUTDATED))
        stale.flags.writeable = False
        self.cached_group_versions[key] = stale

    # -- queries --------------------------------------------------------------------------------
    @property
    def version(self):
        return tuple(inpt.version for inpt in self.inputs.values())

    @property
    def value(self):
        return self._value

    @property
    def shape(self):
        return self._value.shape

    def is_outdated(self):
        return self.cached_version != self.version

    def ahead_of(self, input_key):
        return self.cached_version[self.input_idx[input_key]] != self.inputs[input_key].version

    def cached_version_by_input(self, input_key):
        return self.cached_version[self.input_idx[input_key]]

    def what_changed(self, input_key, caching=True):
        """Indices of the groups of `input_key` whose version differs from the cached stamp
        (all groups when caching is off); a list of keys gives the sorted union (state.py:239-255)."""
        if isinstance(input_key, list):
            return np.unique(np.concatenate([self.what_changed(k, caching=caching) for k in input_key]))
        inpt = self.inputs[input_key]
        if not isinstance(inpt, GroupedParameters):
            raise ValueError("Can only track what changed for GroupedParameters")
        if not caching:
            return np.arange(inpt.n_groups)
        return np.flatnonzero(self.cached_group_versions[input_key] != inpt.group_versions)

    # -- updates --------------------------------------------------------------------------------
    def set_up_to_date(self):
        self.cached_version = self.version
        for key, inpt in self.inputs.items():
            if isinstance(inpt, GroupedParameters):
                stamp = inpt.group_versions.copy()
                stamp.flags.writeable = False
                self.cached_group_versions[key] = stamp

    def update_value(self, new_value):
        self._value = new_value
        self.set_up_to_date()

    @contextmanager
    def edit(self):
        yield self._value
        self.set_up_to_date()

    def clear(self):
        self.cached_version = self.outdated_version()
        for key in self.cached_group_versions:
            self.clear_group_version(key)

    def assign_from(self, other):
        self._value = _shallow(other._value)
        self.cached_version = other.cached_version
        self.cached_group_versions = dict(other.cached_group_versions)


class HasComponents(CacheNode):
    """bool [n_objects, n_components]; column 0 follows the clusters (state.py:353-376)."""

    def __init__(self, clusters, confounders):
        cols = [clusters.any_cluster()] + [conf.any_group() for conf in confounders.values()]
        super().__init__(value=np.array(cols, dtype=bool).T)
        self.clusters = clusters
        self.inputs["clusters"] = clusters

    @property
    def value(self):
        if self.is_outdated():
            self._value[:, 0] = self.inputs["clusters"].any_cluster()
            self.cached_version = self.version
        return self._value


class ModelCache:
    """The cache nodes of the likelihood path and their dependency wiring (state.py:379-489).
    Prior-side nodes are outside the path and not mirrored."""

    def __init__(self, sample):
        n, f, c = sample.n_objects, sample.n_features, sample.n_components
        self.component_likelihoods = CacheNode(np.empty((n, f, c)))
        self.group_likelihoods = {name: CacheNode(np.empty(sample.n_groups(name))) for name in sample.component_names}
        self.weights_normalized = CacheNode(np.empty((n, f, c)))
        self.has_components = HasComponents(sample.clusters, sample.confounders)

        self.component_likelihoods.add_input("clusters", sample.clusters)
        self.weights_normalized.add_input("has_components", self.has_components)
        self.weights_normalized.add_input("weights", sample.weights)
        self.component_likelihoods.add_input("source", sample.source)
        for comp, counts in sample.feature_counts.items():
            self.group_likelihoods[comp].add_input("counts", counts)
            self.component_likelihoods.add_input(f"{comp}_counts", counts)
            if comp != "clusters" and sample.confounders[comp].has_universal_prior:
                self.group_likelihoods[comp].add_input("universal_counts", sample.feature_counts["universal"])

    def clear(self):
        self.component_likelihoods.clear()
        self.weights_normalized.clear()
        self.has_components.clear()
        for node in self.group_likelihoods.values():
            node.clear()

    def copy(self, new_sample):
        new = ModelCache(new_sample)
        new.component_likelihoods.assign_from(self.component_likelihoods)
        new.weights_normalized.assign_from(self.weights_normalized)
        new.has_components.assign_from(self.has_components)
        for comp, node in new.group_likelihoods.items():
            node.assign_from(self.group_likelihoods[comp])
        return new


class Sample:
    """One MCMC state: clusters, weights, source, feature counts + caches (state.py:492-634)."""

    def __init__(self, clusters, weights, confounders, source, feature_counts, model_shapes,
                 chain=0, _other_cache=None, _i_step=0):
        self._clusters, self._weights, self._source = clusters, weights, source
        self._feature_counts = feature_counts
        self.confounders = confounders
        self.model_shapes = model_shapes
        self.chain = chain
        self.i_step = _i_step
        self.cache = ModelCache(self) if _other_cache is None else _other_cache.copy(new_sample=self)
        self.last_lh = None
        self.last_prior = None
        self.observation_lhs = None

    @classmethod
    def from_numpy_arrays(cls, clusters, weights, confounders, source, feature_counts, model_shapes, chain=0):
        return cls(
            clusters=Clusters(np.array(clusters, dtype=bool)),
            weights=ArrayParameter(np.asarray(weights).astype(FLOAT_TYPE)),
            confounders=confounders,
            source=GroupedParameters(np.array(source, dtype=bool), group_dim=0),
            feature_counts={k: FeatureCounts(np.array(v, dtype=FLOAT_TYPE)) for k, v in feature_counts.items()},
            model_shapes=model_shapes, chain=chain)

    def copy(self):
        return Sample(
            clusters=self._clusters.copy(), weights=self._weights.copy(), source=self._source.copy(),
            feature_counts={k: v.copy() for k, v in self._feature_counts.items()},
            confounders=self.confounders, model_shapes=self.model_shapes, chain=self.chain,
            _other_cache=self.cache, _i_step=self.i_step)

    def everything_changed(self):
        self.cache.clear()

    clusters = property(lambda self: self._clusters)
    weights = property(lambda self: self._weights)
    source = property(lambda self: self._source)
    feature_counts = property(lambda self: self._feature_counts)
    n_clusters = property(lambda self: self._clusters.shape[0])
    n_objects = property(lambda self: self._clusters.shape[1])
    n_features = property(lambda self: self._weights.shape[0])
    n_components = property(lambda self: self._weights.shape[1])
    n_states = property(lambda self: self.model_shapes.n_states)

    @property
    def component_names(self):
        return ["clusters", *self.confounders.keys()]

    def n_groups(self, conf):
        return self.n_clusters if conf == "clusters" else self.model_shapes.n_groups[conf]

    def groups_and_clusters(self):
        out = {name: conf.group_assignment for name, conf in self.confounders.items()}
        out["clusters"] = self.clusters.value
        return out


# the node protocol above runs in native code inside the host layer's functions for nodes of exactly this class (_fast.py)
from . import _fast as _fast_mod  # noqa: E402

_fast_mod.register_node_classes(CacheNode, GroupedParameters)
# ... and these classes' read-only properties (Sample.clusters / .weights / .source / .feature_counts, the parameters' .value) are
# read from the instance by the native bind: each returns the attribute of the same name with a leading underscore
_fast_mod.register_trusted(samples=(Sample,), params=(ArrayParameter, GroupedParameters, Clusters, FeatureCounts), counts=(FeatureCounts,))
