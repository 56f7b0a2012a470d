"""Feature counts on the device -- drop-in for sbayes/sampling/counts.py.

Same names, arguments and return conventions as the reference functions:
  compute_effect_counts      counts.py:10-32
  recalculate_feature_counts counts.py:35-52
  update_feature_counts      counts.py:55-95
The counting itself is the HIP histogram kernel behind sbe_effect_counts."""
from __future__ import annotations

import weakref

import numpy as np

from . import _fast, _lib
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


_addr = _fast.addr

FOLLOW_COUNTS = True        # update_feature_counts: the bound slot takes the difference on the device (binding.counts_follow_plan)

# A performance hint, never a correctness matter: ClusterJump evaluates new -> OLD -> new after its update_feature_counts
# (forward resampling, then the backward proposal on the old state, operators.py:1724-1790), so a slot that followed the
# difference would be sent the old rows again and then the new ones.  operators.jump_lh notes the sample it was asked about
# (the step's old state); update_feature_counts from that sample leaves the slot alone; Likelihood.__call__ (the end of
# every step) forgets the note.
_no_follow_from = None


def note_jump_state(sample):
    global _no_follow_from
    try:
        _no_follow_from = weakref.ref(sample)
    except TypeError:
        _no_follow_from = None


def forget_jump_state():
    global _no_follow_from
    _no_follow_from = None


def _group_ids(groups, objs, offset, out):
    """out[:] = global group index of each listed object in one component (-1: in no group), by the library's host
    helper (one pass, no temporaries); False when a listed object is in several groups (no single id: the caller
    counts through the stateless per-group form).  `objs` is int32, C-contiguous."""
    g = groups if type(groups) is np.ndarray else np.asarray(groups)
    if g.dtype != np.bool_ or not g.flags.c_contiguous:
        g = np.ascontiguousarray(g, dtype=bool)
    rc = _lib.load().sbe_host_group_ids(_addr(g), g.shape[0], g.shape[1], _addr(objs), objs.size, int(offset), _addr(out))
    if rc < 0:
        raise ValueError("object index out of range in object_subset")
    return rc == 0


def _source_ids(source, objs):
    """bool [N, F, C] one-hot over components, listed objects -> uint8 [n, F] component id per observation (255: none)."""
    s = source if type(source) is np.ndarray else np.asarray(source)
    if s.dtype != np.bool_ or not s.flags.c_contiguous:
        s = np.ascontiguousarray(s, dtype=bool)
    out = np.empty((objs.size, s.shape[1]), dtype=np.uint8)
    if _lib.load().sbe_host_source_ids(_addr(s), s.shape[0], s.shape[1], s.shape[2], _addr(objs), objs.size, _addr(out)) != 0:
        raise ValueError("object index out of range in object_subset")
    return out


def update_feature_counts(sample_old, sample_new, features, object_subset):
    """Delta update of `sample_new.feature_counts` for the objects whose assignment changed (counts.py:55-95).
    ONE device call for all components: the subset's group ids and source rows of both samples go up (n * (8 C + 2 F)
    bytes), the count rows of the groups those objects are in come back; the reference's `add_changes(diff)` follows
    with the same `diff` it would have computed (zero rows for every other group) -- in its row form
    (FeatureCounts.add_changes_rows: patch.install / sbayes_amd.state) where the sample's class has one.

    The function body below is the reference form; with the extension built the same steps run as one native call
    (csrc/sbe_pyhost.c: update_counts -- the same ids, the same engine call, the same add_changes and bind-cache follow-up),
    which hands back NotImplemented for the argument forms it does not serve (slices, lists, repeated objects, overlapping groups)."""
    if _NATIVE_UPDATE:
        hold = _no_follow_from is not None and _no_follow_from() is sample_old
        res = _fast._h.update_counts(sample_old, sample_new, features, object_subset, FOLLOW_COUNTS and not hold)
        if res is not NotImplemented:
            return res
    counts = sample_new.feature_counts
    conf_names = list(sample_new.confounders)
    names = ["clusters", *conf_names]
    groups_old = [sample_old.clusters.value] + [sample_old.confounders[k].group_assignment for k in conf_names]
    groups_new = [sample_new.clusters.value] + [sample_new.confounders[k].group_assignment for k in conf_names]
    n_groups = [g.shape[0] for g in groups_new]
    eng = get_engine(features, n_groups)
    objs = _subset_indices(object_subset, features.shape[0])
    if objs.dtype != np.int32 or not objs.flags.c_contiguous:
        objs = np.ascontiguousarray(objs, dtype=np.int32)
    off = eng.group_offsets
    src_old, src_new = sample_old.source.value, sample_new.source.value
    ids = _fast.subset_ids(objs, groups_new, groups_old, src_new, src_old)
    if ids is None:
        # repeated objects, or a listed object in several groups of one component (counted once per group,
        # counts.py:28-30): the reference's own two-count difference, each count by the stateless device histogram
        for i, name in enumerate(names):
            old = compute_effect_counts(features, groups_old[i], sample_old.source.value[..., i], object_subset)
            new = compute_effect_counts(features, groups_new[i], sample_new.source.value[..., i], object_subset)
            counts[name].add_changes(diff=new - old)
        return counts
    # a slot that holds the counts this difference is added to follows on the device, in the same call (binding.counts_follow_plan)
    from .binding import _token, counts_follow_plan, counts_followed, note_source_lineage
    src_parent = _token(sample_old.source)
    hold = _no_follow_from is not None and _no_follow_from() is sample_old
    plan = counts_follow_plan(eng, sample_new, names) if FOLLOW_COUNTS and not hold else None
    if plan is None:
        touched, rows = eng.counts_delta(objs, *ids)
        # where the two samples' sources differ, for the binds to come (binding.py: source lineage): this call's own contract
        note_source_lineage(src_parent, _token(sample_new.source), objs)
        return apply_count_rows(counts, names, off, touched, rows)
    # (the probability rows are rebuilt along when no table of the slot is stale: the touched components are not known yet)
    rebuild = not plan[2]
    # the subset's new source rows are in the call anyway (src_new): a slot that has a source takes them too
    mirror_src = plan[1]["source"]
    with_source = mirror_src is not None and mirror_src.shape == np.shape(src_new)
    touched, rows = eng.counts_delta(objs, *ids, follow_slot=0, update_probs=rebuild, update_source=with_source)
    bounds = apply_count_rows(counts, names, off, touched, rows, return_bounds=True)
    if not with_source:
        note_source_lineage(src_parent, _token(sample_new.source), objs)
    counts_followed(eng, plan, sample_new, names, touched, bounds, rebuild, objs if with_source else None, source_parent=src_parent)
    return counts


def apply_count_rows(counts, names, off, touched, rows, return_bounds=False):
    """The reference's `add_changes(diff)` per component (counts.py:77, :93) for a difference given as the rows of the global
    group indices `touched` (ascending; every other row is zero): in its row form (FeatureCounts.add_changes_rows: patch.install
    / sbayes_amd.state) where the sample's class has one, else through a dense diff."""
    if _fast._h is not None and type(touched) is np.ndarray and touched.dtype == np.int32 and type(rows) is np.ndarray and rows.dtype == np.float32:
        # every component in ONE native call (sbe_pyhost.c: add_rows_many -- the same resolve_sharing / += / version /
        # group_versions steps as add_changes_rows, node by node); None: a node of another form, the loop below serves it
        bounds = _fast._h.add_rows_many([counts[name] for name in names], off, touched, rows)
        if bounds is not None:
            return bounds if return_bounds else counts
    bounds = np.searchsorted(touched, off).tolist()  # `touched` is sorted: the rows of component c are bounds[c]:bounds[c+1]
    for c, name in enumerate(names):
        node = counts[name]
        lo, hi = bounds[c], bounds[c + 1]
        add_rows = getattr(node, "add_changes_rows", None)
        if add_rows is not None:
            add_rows(touched[lo:hi] - off[c], rows[lo:hi])
            continue
        diff = np.zeros(node.value.shape, dtype=np.float32)
        if hi > lo:
            diff[touched[lo:hi] - off[c]] = rows[lo:hi]
        node.add_changes(diff=diff)
    return bounds if return_bounds else counts


def _get_engine_now(features, n_groups):
    return get_engine(features, n_groups)          # (looked up at call time: tests swap the module's get_engine)


_NATIVE_UPDATE = False
if _fast._h is not None and hasattr(_fast._h, "update_counts"):
    from . import binding as _binding
    if _binding._bind_slot is not _binding._bind_slot_py:            # (the native bind is set up: its helpers are shared)
        _fast._h.update_counts_setup(np.empty, np.dtype(np.int32), np.dtype(np.uint8), _get_engine_now, _binding.note_source_lineage,
                                     _binding._source_followed, apply_count_rows)
        _NATIVE_UPDATE = True
