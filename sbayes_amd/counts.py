"""Feature counts on the device -- drop-in for sbayes/sampling/counts.py.

Same names, arguments and return conventions as the reference functions:
  compute_effect_counts      counts.py:10-32
  recalculate_feature_counts counts.py:35-52
  update_feature_counts      counts.py:55-95
The counting itself is the HIP histogram kernel behind sbe_effect_counts."""
from __future__ import annotations

import numpy as np

from .registry import get_engine


def compute_effect_counts(features, group_assignment, source_is_component, object_subset=slice(None)):
    """float32 [n_groups, n_features, n_states]: state counts of the observations whose source is
    this component, per group, restricted to `object_subset` (slice(None), index list or bool mask)."""
    eng = get_engine(features)
    if isinstance(object_subset, slice):
        subset = None if object_subset == slice(None) else np.arange(features.shape[0])[object_subset]
    else:
        subset = np.asarray(object_subset)
        if subset.dtype == np.bool_:
            subset = np.flatnonzero(subset)
    return eng.effect_counts(group_assignment, source_is_component, subset)


def recalculate_feature_counts(features, sample):
    """Recount every mixture component from `sample.source` and store the result in
    `sample.feature_counts` (set_value bumps all group versions, like the reference)."""
    source = sample.source.value
    sample.feature_counts["clusters"].set_value(
        compute_effect_counts(features, sample.clusters.value, source[..., 0]))
    for i, conf in enumerate(sample.confounders.keys(), start=1):
        groups = sample.confounders[conf].group_assignment
        sample.feature_counts[conf].set_value(compute_effect_counts(features, groups, source[..., i]))
    return sample.feature_counts


def update_feature_counts(sample_old, sample_new, features, object_subset):
    """Delta update of `sample_new.feature_counts` for the objects whose assignment changed."""
    counts = sample_new.feature_counts
    names = ["clusters", *sample_new.confounders.keys()]
    for i, name in enumerate(names):
        if name == "clusters":
            g_old, g_new = sample_old.clusters.value, sample_new.clusters.value
        else:
            g_old = sample_old.confounders[name].group_assignment
            g_new = sample_new.confounders[name].group_assignment
        old = compute_effect_counts(features, g_old, sample_old.source.value[..., i], object_subset)
        new = compute_effect_counts(features, g_new, sample_new.source.value[..., i], object_subset)
        counts[name].add_changes(diff=new - old)
    return counts
