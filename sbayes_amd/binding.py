"""Bind cache: which sample state an engine slot holds, and the DELTA uploads that bring it to another sample.

The reference's operators evaluate the same or nearly the same sample many times per MCMC step (current sample,
candidate = copy + a few edits, current again after a reject); the device forms of the drop-in layer
(sbayes_amd/operators.py, likelihood.py, conditionals.py) bind the sample they are given into an engine slot on every
call.  `_bind_slot` sends only what differs from what the slot holds ("What crosses PCIe per step"):

  group ids        N * 2 bytes per component whose cluster / group matrix changed
  counts           the [F, S] float32 rows of the groups whose counts differ (Engine.set_counts_rows)
  source           the [F] rows of the objects whose source assignment differs (Engine.set_source_rows)
  weights          F * C * 4 bytes
  concentrations   only when their content changes (static priors: once)

An entry is trusted only through the reference's own change tracking -- same ndarray object AND same parameter version
(sbayes/sampling/state.py:34-61, 97-161, 340-350) -- or through content: when the token differs, counts and source are
compared with the private host mirror of what the slot holds, row by row, and the differing rows go up.  Version
numbers alone are never compared across arrays (independent chains carry similar version counters)."""
from __future__ import annotations

import numpy as np

from .registry import get_engine


def _engine(model):
    lik = getattr(model, "likelihood", None)
    if lik is not None and hasattr(lik, "engine"):
        return lik.engine
    n_groups = [model.shapes.n_clusters] + [c.n_groups for c in model.data.confounders.values()]
    return get_engine(model.data.features.values, n_groups)


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


def _changed_rows(new, mirror):
    """Indices along axis 0 where `new` differs from `mirror` (same shape and dtype), compared as raw bytes."""
    if new.shape[0] == 0 or new.size == 0:
        return np.zeros(0, dtype=np.int64)
    a = np.ascontiguousarray(new).reshape(new.shape[0], -1)
    b = mirror.reshape(mirror.shape[0], -1)
    if a.dtype.itemsize == 1 and a.shape[1] % 8 == 0:           # bool rows: eight observations per compare
        a, b = a.view(np.uint64), b.view(np.uint64)
    return np.flatnonzero((a != b).any(axis=1))


def _send_counts(eng, slot, c, new, mirror, pending):
    """Bring the slot's counts of component c to `new` [G, F, S]; returns (mirror, anything changed).  Changed rows are
    appended to `pending` = ([global group indices], [rows]) -- _bind_slot sends the rows of ALL components with one
    set_counts_rows call; a component the slot has never held goes up whole."""
    new = np.asarray(new, dtype=np.float32)
    if mirror is None or mirror.shape != new.shape or not hasattr(eng, "set_counts_rows"):
        eng.set_counts(slot, c, new)
        return new.copy(), True
    rows = _changed_rows(new, mirror)
    if rows.size == 0:
        return mirror, False
    pending[0].append(int(eng.group_offsets[c]) + rows)
    pending[1].append(new[rows])
    mirror[rows] = new[rows]
    return mirror, True


def _send_source(eng, slot, new, mirror):
    new = np.asarray(new, dtype=bool)
    if mirror is None or mirror.shape != new.shape:
        eng.set_source(slot, new)
        return new.copy()
    rows = _changed_rows(new, mirror)
    if rows.size == 0:
        return mirror
    if 2 * rows.size > new.shape[0]:
        eng.set_source(slot, new)
        return new.copy()
    eng.set_source_rows(slot, rows, new[rows])
    mirror[rows] = new[rows]
    return mirror


def _bind_slot(eng, model, sample, slot, with_source=False):
    """Make engine slot `slot` hold `sample`: group ids, counts, weights, optionally the source assignment, and the
    priors' concentration tables.  Only what differs from what the slot holds is sent (module docstring); returns the
    components whose probability tables are stale.  `model=None` binds the state without the priors' tables and
    without the counts (recount_bound: the counts are about to be recomputed on the device)."""
    names = sample.component_names
    C = len(names)
    groups = [_token(sample.clusters)] + [_token(c.group_assignment) for c in sample.confounders.values()]
    if model is not None:
        conc = [_token(model.prior.prior_cluster_effect.concentration_array)] + [
            _token(model.prior.prior_confounding_effects[k].concentration_array(sample)) for k in names[1:]]
        counts = [_token(sample.feature_counts[name]) for name in names]
    else:
        conc = counts = None
    weights = _token(sample.weights)
    source = _token(sample.source) if with_source else None
    cache = getattr(eng, "_bound", None)
    if cache is None:                               # an engine without a bind cache (test doubles): send everything
        for c in range(C):
            eng.set_groups(slot, c, groups[c][0])
            if model is not None:
                eng.set_concentration(c, conc[c][0])
                eng.set_counts(slot, c, counts[c][0])
        if with_source:
            eng.set_source(slot, source[0])
        eng.set_weights(slot, weights[0])
        return set(range(C))
    old = cache.get(slot) or {"groups": [None] * C, "counts": [None] * C, "weights": None, "source": None, "stale": set(range(C))}
    # the host mirrors of what the slot holds survive the entry being dropped by the engine's own setters as long as
    # those setters are the ones called from here (eng._mirror is cleared by every OTHER slot-changing call: _touch)
    mirrors = eng._mirror.setdefault(slot, {"counts": [None] * C, "source": None}) if hasattr(eng, "_mirror") else \
        {"counts": [None] * C, "source": None}
    new = {"groups": list(old["groups"]), "counts": list(old["counts"]), "weights": old["weights"], "source": old["source"],
           "stale": set(old["stale"]), "lh_all": old.get("lh_all")}
    pending = ([], [])                              # count rows of all components: one set_counts_rows call
    conc_changed = [] if conc is None else [c for c in range(C) if not _same(conc[c], eng._bound_conc.get(c))]
    for c in conc_changed:                          # (drops every slot's entry: all tables depend on it)
        eng.set_concentration(c, conc[c][0])
    if conc_changed:
        new["stale"] = set(range(C))
        new["lh_all"] = None
    for c in range(C):
        if not _same(groups[c], old["groups"][c]):
            eng.set_groups(slot, c, groups[c][0])
            new["groups"][c] = _remember(groups[c])
        if counts is not None and not _same(counts[c], old["counts"][c]):
            mirrors["counts"][c], changed = _send_counts(eng, slot, c, counts[c][0], mirrors["counts"][c], pending)
            new["counts"][c] = _remember(counts[c])
            if changed:
                new["stale"].add(c)
                new["lh_all"] = None                # (Likelihood._group_logliks' memo of the collapsed log-likelihoods)
    if pending[0]:
        eng.set_counts_rows(slot, np.concatenate(pending[0]), np.concatenate(pending[1]))
    if with_source and not _same(source, old["source"]):
        mirrors["source"] = _send_source(eng, slot, source[0], mirrors["source"])
        new["source"] = _remember(source)
    if not _same(weights, old["weights"]):
        eng.set_weights(slot, weights[0])
        new["weights"] = _remember(weights)
    for c in conc_changed:
        eng._bound_conc[c] = _remember(conc[c])
    cache[slot] = new                               # (the setters above dropped the slot's entry)
    if hasattr(eng, "_mirror"):
        eng._mirror[slot] = mirrors                 # (and its mirrors)
    return new["stale"]


def recount_bound(eng, sample, slot=0):
    """recalculate_feature_counts (counts.py:35-52) from RESIDENT data: bind the sample's groups and source (deltas
    only), recount every component on the device, fetch the tables.  The slot's bind entry survives: its count
    mirrors become the fetched tables, so the set_value() that follows on the host costs no upload.
    Returns the float32 [G_c, F, S] tables, component by component."""
    _bind_slot(eng, None, sample, slot, with_source=True)
    entry = eng._bound.get(slot) if hasattr(eng, "_bound") else None
    mirrors = eng._mirror.get(slot) if hasattr(eng, "_mirror") else None
    eng.recount(slot)                               # (drops the slot's entry and mirrors: the counts changed under them)
    tables = [eng.get_counts(slot, c) for c in range(eng.n_components)]
    if entry is not None and mirrors is not None:
        entry["counts"] = [None] * eng.n_components          # whatever parameter holds them next is compared by content
        entry["stale"] = set(range(eng.n_components))
        entry["lh_all"] = None
        mirrors["counts"] = [t.copy() for t in tables]
        eng._bound[slot], eng._mirror[slot] = entry, mirrors
    return tables


def _bind_uniform(eng, model):
    """The cluster prior's uniform concentration (prior.py:184-186) as the engine's resident `unif` operand of the
    tempered tables; sent when its content changes (once per model)."""
    unif = _token(np.asarray(model.prior.prior_cluster_effect.uniform_concentration_array))
    if not _same(unif, getattr(eng, "_bound_unif", None)):
        eng.set_uniform_counts(unif[0])
        eng._bound_unif = _remember(unif)


def _tables_current(eng, slot):
    """Rebuild the probability tables that the last _bind_slot left stale."""
    entry = getattr(eng, "_bound", {}).get(slot)
    stale = set(range(eng.n_components)) if entry is None else entry["stale"]
    if len(stale) == 1:
        eng.update_probs(slot, next(iter(stale)))
    elif stale:
        eng.update_probs(slot, sorted(stale))      # one call (adjacent components: one launch)
    if entry is not None:
        entry["stale"] = set()
