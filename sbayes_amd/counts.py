"""Feature counts on the device -- drop-in for sbayes/sampling/counts.py.

Same names, arguments and return conventions as the reference functions:
  compute_effect_counts      counts.py:10-32
  recalculate_feature_counts counts.py:35-52
  update_feature_counts      counts.py:55-95
The counting itself is the HIP histogram kernel behind sbe_effect_counts."""
from __future__ import annotations

import numpy as np

from .engine import GroupOverlapError
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
    """Recount every mixture component from `sample.source` and store the result in `sample.feature_counts`
    (set_value bumps all group versions, like the reference).  The counting runs on the sample's RESIDENT state: only
    the source rows / group ids that differ from what the engine slot holds go up, the [G, F, S] tables come back."""
    from .binding import recount_bound
    names = ["clusters", *sample.confounders.keys()]
    n_groups = [int(np.shape(sample.clusters.value)[0])] + [int(np.shape(sample.confounders[k].group_assignment)[0])
                                                           for k in names[1:]]
    eng = get_engine(features, n_groups)
    try:
        tables = recount_bound(eng, sample)
    except GroupOverlapError:
        # an object in several groups of one component has no resident form (one group id per object and component);
        # the reference counts it once per group (counts.py:28-30), and so does the stateless device histogram
        groups = [sample.clusters.value] + [sample.confounders[k].group_assignment for k in names[1:]]
        source = sample.source.value
        tables = [eng.effect_counts(groups[c], source[..., c]) for c in range(len(names))]
    for name, table in zip(names, tables):
        sample.feature_counts[name].set_value(table)
    return sample.feature_counts


def _subset_indices(object_subset, n_objects):
    if isinstance(object_subset, slice):
        return np.arange(n_objects)[object_subset]
    subset = np.asarray(object_subset)
    return np.flatnonzero(subset) if subset.dtype == np.bool_ else subset.astype(np.int64, copy=False).reshape(-1)


def _group_ids(groups, objs, offset):
    """Global group index of each listed object in one component (-1: in no group); None when a listed object is in
    several groups (no single id: the caller counts through the stateless per-group form)."""
    sub = np.asarray(groups)[:, objs]
    k = np.count_nonzero(sub, axis=0)                 # groups each listed object is in
    if k.size and k.max() > 1:
        return None
    return np.where(k > 0, sub.argmax(axis=0) + offset, -1).astype(np.int32)


def _source_ids(source_rows):
    """bool [n, F, C] -> component id per observation (255: none)."""
    return np.where(source_rows.any(axis=-1), source_rows.argmax(axis=-1), 255).astype(np.uint8)


def update_feature_counts(sample_old, sample_new, features, object_subset):
    """Delta update of `sample_new.feature_counts` for the objects whose assignment changed (counts.py:55-95).
    ONE device call for all components: the subset's group ids and source rows of both samples go up (n * (8 C + 2 F)
    bytes), the count rows of the groups those objects are in come back; the reference's `add_changes(diff)` follows
    with the same `diff` it would have computed (zero rows for every other group)."""
    counts = sample_new.feature_counts
    names = ["clusters", *sample_new.confounders.keys()]
    groups_old = [sample_old.clusters.value] + [sample_old.confounders[k].group_assignment for k in names[1:]]
    groups_new = [sample_new.clusters.value] + [sample_new.confounders[k].group_assignment for k in names[1:]]
    n_groups = [int(np.shape(g)[0]) for g in groups_new]
    eng = get_engine(features, n_groups)
    objs = _subset_indices(object_subset, np.shape(features)[0])
    off = np.concatenate([[0], np.cumsum(n_groups)]).astype(int)
    unique = len(np.unique(objs)) == len(objs)     # (the reference's fancy index would count a repeated object twice)
    gid_old = [_group_ids(groups_old[c], objs, off[c]) for c in range(len(names))] if unique else None
    gid_new = [_group_ids(groups_new[c], objs, off[c]) for c in range(len(names))] if unique else None
    if not unique or any(g is None for g in gid_old) or any(g is None for g in gid_new):
        # repeated objects, or a listed object in several groups of one component (counted once per group,
        # counts.py:28-30): the reference's own two-count difference, each count by the stateless device histogram
        for i, name in enumerate(names):
            old = compute_effect_counts(features, groups_old[i], sample_old.source.value[..., i], object_subset)
            new = compute_effect_counts(features, groups_new[i], sample_new.source.value[..., i], object_subset)
            counts[name].add_changes(diff=new - old)
        return counts
    gid_old, gid_new = np.stack(gid_old), np.stack(gid_new)
    touched, rows = eng.counts_delta(objs, gid_old, gid_new, _source_ids(sample_old.source.value[objs]),
                                     _source_ids(sample_new.source.value[objs]))
    for c, name in enumerate(names):
        diff = np.zeros(counts[name].value.shape, dtype=np.float32)
        mine = (touched >= off[c]) & (touched < off[c + 1])
        diff[touched[mine] - off[c]] = rows[mine]
        counts[name].add_changes(diff=diff)
    return counts
