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
numbers alone are never compared across arrays (independent chains carry similar version counters).

CONTRACT for unversioned arrays (confounder group matrices, static concentration tables): an array that is READ-ONLY and owns
its data is recognised by identity alone, with no private copy kept -- the reference freezes its static tables exactly so
(sbayes/model/prior.py:322-323) and `Likelihood._freeze_static_inputs` freezes the group matrices and the cluster prior's
tables.  Whoever thaws such an array (`setflags(write=True)`), writes into it and re-freezes it between two binds changes the
data behind the bind cache's back: the resident copy goes stale SILENTLY.  The sampler never does that to these arrays (its
own thaw / write / freeze pattern, FeatureCounts.add_changes, is on VERSIONED parameters, which are compared by version).
`SBAYES_AMD_VERIFY_FROZEN=1` (debugging aid; the Python form of the bind) keeps a checksum of every frozen array it trusts
and raises when the content behind an unchanged identity differs."""
from __future__ import annotations

import numpy as np

from . import _fast
from .registry import get_engine


def _engine(model):
    lik = getattr(model, "likelihood", None)
    if lik is not None and hasattr(lik, "engine"):
        return lik.engine
    n_groups = [model.shapes.n_clusters] + [c.n_groups for c in model.data.confounders.values()]
    return get_engine(model.data.features.values, n_groups)


_ndarray = np.ndarray


def _token(param):
    """(array, version) of a state parameter.  The reference's (and the mirror's) parameters bump `version` on every
    edit -- set_value, set_items, edit(), edit_group(s), set_groups, FeatureCounts.add_changes
    (sbayes/sampling/state.py:34-61, 97-161, 340-350) -- and edit the SAME ndarray in place whenever the parameter is
    not shared with a copy, so array identity alone says nothing; plain arrays (confounder group matrices,
    concentration tables) have no version and are compared by content, or by identity when they are frozen (_same)."""
    value = getattr(param, "value", param)
    if type(value) is not _ndarray:
        value = np.asarray(value)
    return value, getattr(param, "version", None)


_VERIFY_FROZEN = bool(__import__("os").environ.get("SBAYES_AMD_VERIFY_FROZEN"))
_FROZEN_SUMS = {}           # id(array) -> (array, adler32 of its bytes), SBAYES_AMD_VERIFY_FROZEN only


def _frozen_sum(arr):
    import zlib
    return zlib.adler32(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))


def _same(tok, cached):
    """True if the token `tok` = (array, version) denotes what `cached` = (array, version, private copy) recorded.
    Versioned parameters: same ndarray object AND same version (an in-place edit through the parameter API always
    bumps the version; a copy-on-write edit always creates a new ndarray).  Unversioned arrays: the SAME ndarray object
    (kept alive by the entry, so its id cannot be recycled) that owns its data and is read-only cannot have changed in
    place -- the reference freezes its static concentration tables exactly so (sbayes/model/prior.py:322-323), and the
    drop-in Likelihood freezes the confounders' group matrices and the cluster prior's tables (likelihood.py
    `_freeze_static_inputs`); anything else -- a writeable array (dynamic priors rewrite theirs in place,
    prior.py:325-354), a view -- is compared by content against the private copy."""
    if cached is None:
        return False
    arr, version = tok
    ref, ref_version, copy = cached
    if version is not None and ref_version is not None:
        return arr is ref and version == ref_version
    if arr is ref and copy is None:                 # (_remember kept no copy: the array was frozen and owned its data)
        flags = arr.flags
        if not flags.writeable and flags.owndata:
            if _VERIFY_FROZEN:
                rec = _FROZEN_SUMS.get(id(arr))
                if rec is not None and rec[0] is arr and rec[1] != _frozen_sum(arr):
                    raise RuntimeError("sbayes_amd: a frozen array the bind cache recognises by identity was modified in place "
                                       "(thawed, written, re-frozen) -- the resident copy is stale (binding.py: CONTRACT)")
            return True
        return False                                # thawed since: what it holds now is unknown -> re-send
    return copy is not None and arr.shape == copy.shape and arr.dtype == copy.dtype and np.array_equal(arr, copy)


def _scan_py(params, cached):
    """(tokens, changed): the token of every parameter and a bit mask of those that do not denote what `cached` recorded
    (the extension's scan() is this loop in C: sbayes_amd/csrc/sbe_pyhost.c)."""
    tokens = [_token(p) for p in params]
    changed = 0
    for i, tok in enumerate(tokens):
        if not _same(tok, cached[i]):
            changed |= 1 << i
    return tokens, changed


def _content_equal(arr, copy):
    return arr.shape == copy.shape and arr.dtype == copy.dtype and np.array_equal(arr, copy)


if _fast.HAVE_EXTENSION:
    _fast._h.scan_setup(np.ndarray, np.asarray, _content_equal)
    _scan = _fast._h.scan if not _VERIFY_FROZEN else _scan_py
else:
    _scan = _scan_py


_ROWS_WITH_PROBS = True     # count rows go up with the rebuild of their probability rows (Engine.set_counts_rows(update_probs=True))


def _remember(tok):
    arr, version = tok
    # the ndarray itself is kept alive so that its id cannot be recycled for another array while the entry lives
    if version is not None:
        return arr, version, None
    flags = arr.flags
    if not flags.writeable and flags.owndata:       # frozen: identity is the whole comparison (no copy to keep)
        if _VERIFY_FROZEN:
            _FROZEN_SUMS[id(arr)] = (arr, _frozen_sum(arr))
        return arr, None, None
    return arr, None, arr.copy()


_changed_rows = _fast.diff_rows         # rows of `new` that differ from `mirror`; the mirror takes them on the way


def _send_counts(eng, slot, c, new, mirror, pending):
    """Bring the slot's counts of component c to `new` [G, F, S]; returns (mirror, anything changed).  Changed rows are
    appended to `pending` = ([global group indices], [rows]) -- _bind_slot sends the rows of ALL components with one
    set_counts_rows call; a component the slot has never held goes up whole."""
    if type(new) is not _ndarray or new.dtype != np.float32:
        new = np.asarray(new, dtype=np.float32)
    if mirror is None or mirror.shape != new.shape or not hasattr(eng, "set_counts_rows"):
        eng.set_counts(slot, c, new)
        return new.copy(), True
    rows = _changed_rows(new, mirror)                # (the mirror now holds `new`)
    if rows.size == 0:
        return mirror, False
    pending[0].append(int(eng.group_offsets[c]) + rows)
    pending[1].append(new[rows])
    return mirror, True


# ---- source lineage -----------------------------------------------------------------------------------------------------------
# The source array is the largest thing a bind compares (N * F * C bytes: 400 KB at 1000 x 200 x 2, 35 us per compare, 1.3
# compares per MCMC step -- tools/host_residual.py).  Almost every source the sampler binds is the child of one it bound a moment
# ago: `sample_new = sample.copy()`, a few objects' rows edited, `update_feature_counts(sample, sample_new, features,
# object_subset)` (operators.py:520-560, 796-851, 1724-1790).  That call's own contract says where the two samples differ: the
# counts it maintains are only right if nothing outside `object_subset` changed (counts.py:55-95).  So every such call leaves a
# note -- child token = parent token + these rows -- and a bind whose slot holds the parent, or a sibling (the previous, rejected
# proposal), or the child (back to the current sample after a reject) compares just the noted rows instead of the whole array.
# Tokens are (ndarray object, parameter version), the same identity the whole bind cache rests on; the notes hold the arrays,
# so an id cannot be recycled while a note lives.  No note, or any other relation: the full compare, as before.
_LINEAGE = []               # newest last: (child array, child version, parent array, parent version, int32 rows)
_LINEAGE_KEPT = 6


def _rows32(rows):
    if type(rows) is _ndarray and rows.dtype == np.int32 and rows.ndim == 1 and rows.flags.c_contiguous:
        return rows
    rows = np.asarray(rows)
    if rows.dtype == np.bool_:
        rows = np.flatnonzero(rows)
    if rows.dtype != np.int32 or not rows.flags.c_contiguous or rows.ndim != 1:
        rows = np.ascontiguousarray(rows.reshape(-1), dtype=np.int32)
    return rows


def note_source_lineage(parent_tok, child_tok, rows):
    """content(child) == content(parent) outside `rows` (object indices).  Versioned tokens only."""
    p_arr, p_ver = parent_tok
    c_arr, c_ver = child_tok
    if p_ver is None or c_ver is None or (p_arr is c_arr and p_ver == c_ver) or p_arr.shape != c_arr.shape:
        return None
    rec = (c_arr, c_ver, p_arr, p_ver, _rows32(rows).copy())        # (a private copy: the caller's index array may be reused)
    _LINEAGE.append(rec)
    if len(_LINEAGE) > _LINEAGE_KEPT:
        del _LINEAGE[0]
    return rec


def forget_source_lineage():
    del _LINEAGE[:]


def _lineage_of(arr, version):
    for rec in reversed(_LINEAGE):
        if rec[0] is arr and rec[1] == version:
            return rec
    return None


def _source_candidates(arr, version, rec, known, slot_parent):
    """The rows in which the source (arr, version) can differ from what the slot holds, as a tuple of one or two index arrays, or
    None when nothing is known.  `known`: the entry's token of the slot's source (its content equals the mirror's); `rec`:
    the lineage note of (arr, version); `slot_parent`: (parent array, parent version, rows) of the slot's source."""
    if version is None or known is None or known[1] is None:
        return None
    s_arr, s_ver = known[0], known[1]
    if rec is not None and rec[2] is s_arr and rec[3] == s_ver:
        return (rec[4],)                                              # the slot holds the parent
    if slot_parent is not None:
        if slot_parent[0] is arr and slot_parent[1] == version:
            return (slot_parent[2],)                                  # the slot holds a child of this one (back after a reject)
        if rec is not None and rec[2] is slot_parent[0] and rec[3] == slot_parent[1]:
            return (rec[4], slot_parent[2])                           # siblings: the previous proposal was rejected
    return None


def _send_source(eng, slot, tok, mirrors, known=None):
    """Bring the slot's source to the parameter token `tok` = (array, version); returns (mirror, rows op): a whole-array change
    goes up at once, a few changed rows are handed back as (object indices, rows) for _bind_slot to send -- alone, or with the
    bind's other row uploads in one launch.  `mirrors`: the slot's mirror dictionary (its "source" entry is compared and
    updated; "source_parent" follows the lineage); `known`: the bind entry's token of the source the slot holds, if any."""
    new, version = tok
    mirror = mirrors.get("source")
    rec = _lineage_of(new, version) if version is not None else None
    slot_parent = mirrors.get("source_parent")
    mirrors["source_parent"] = (rec[2], rec[3], rec[4]) if rec is not None else None
    cand = _source_candidates(new, version, rec, known, slot_parent)
    if type(new) is not _ndarray or new.dtype != np.bool_:
        new = np.asarray(new, dtype=bool)
    if mirror is None or mirror.shape != new.shape:
        eng.set_source(slot, new)
        return new.copy(), None
    rows = _changed_rows(new, mirror) if cand is None else _fast.diff_rows_among(new, mirror, *cand)   # (the mirror now holds `new`)
    if rows.size == 0:
        return mirror, None
    if 2 * rows.size > new.shape[0]:
        eng.set_source(slot, new)
        return mirror, None
    return mirror, (rows, new[rows])


def _bind_slot(eng, model, sample, slot, with_source=False):
    """Make engine slot `slot` hold `sample`: group ids, counts, weights, optionally the source assignment, and the
    priors' concentration tables.  Only what differs from what the slot holds is sent (module docstring); returns the
    components whose probability tables are stale.  `model=None` binds the state without the priors' tables and
    without the counts (recount_bound: the counts are about to be recomputed on the device).

    The common case inside an MCMC step -- the slot already holds this sample, or differs from it in one or two
    parameters -- costs a handful of identity / version comparisons: nothing is copied, allocated or rebuilt unless a
    parameter really differs (tools/host_residual.py measures this layer: it is host time of every MCMC step)."""
    confounders = sample.confounders
    conf_names = list(confounders)
    C = 1 + len(conf_names)
    cache = getattr(eng, "_bound", None)
    if cache is None:                               # an engine without a bind cache (test doubles): send everything
        names = ["clusters", *conf_names]
        eng.set_groups(slot, 0, np.asarray(sample.clusters.value))
        for c in range(C):
            if c:
                eng.set_groups(slot, c, np.asarray(confounders[names[c]].group_assignment))
            if model is not None:
                conc_c = model.prior.prior_cluster_effect.concentration_array if c == 0 else \
                    model.prior.prior_confounding_effects[names[c]].concentration_array(sample)
                eng.set_concentration(c, np.asarray(conc_c))
                eng.set_counts(slot, c, np.asarray(sample.feature_counts[names[c]].value))
        if with_source:
            eng.set_source(slot, np.asarray(sample.source.value))
        eng.set_weights(slot, np.asarray(sample.weights.value))
        return set(range(C))
    old = cache.get(slot)
    if old is None:
        old = {"groups": [None] * C, "counts": [None] * C, "weights": None, "source": None, "stale": set(range(C)), "lh_all": None}
    # ---- what differs?  (one pass over the parameters -- _scan: the extension's C loop, or _token / _same below it --
    # nothing is written until something does differ) ----
    old_groups, old_counts = old["groups"], old["counts"]
    params = [sample.clusters]
    for k in conf_names:
        params.append(confounders[k].group_assignment)
    cached = list(old_groups)
    if model is not None:
        prior = model.prior
        conf_priors = prior.prior_confounding_effects
        feature_counts = sample.feature_counts
        bound_conc = eng._bound_conc
        params.append(prior.prior_cluster_effect.concentration_array)
        cached.append(bound_conc.get(0))
        for c, k in enumerate(conf_names, start=1):
            params.append(conf_priors[k].concentration_array(sample))
            cached.append(bound_conc.get(c))
        params.append(feature_counts["clusters"])
        for k in conf_names:
            params.append(feature_counts[k])
        cached += old_counts
    params.append(sample.weights)
    cached.append(old["weights"])
    if with_source:
        params.append(sample.source)
        cached.append(old["source"])
    tokens, changed = _scan(params, cached)
    if not changed:
        if slot not in cache:                       # (a first bind of an empty state: keep the entry)
            cache[slot] = old
        return old["stale"]
    comps = range(C)
    groups = tokens[:C]
    groups_changed = [c for c in comps if (changed >> c) & 1]
    at = C
    conc = counts = None
    conc_changed = counts_changed = ()
    if model is not None:
        conc, counts = tokens[C:2 * C], tokens[2 * C:3 * C]
        conc_changed = [c for c in comps if (changed >> (C + c)) & 1]
        counts_changed = [c for c in comps if (changed >> (2 * C + c)) & 1]
        at = 3 * C
    weights = tokens[at]
    weights_changed = (changed >> at) & 1
    source = None
    source_changed = False
    if with_source:
        source = tokens[at + 1]
        source_changed = (changed >> (at + 1)) & 1
    # ---- send the differences ----
    # the host mirrors of what the slot holds survive the entry being dropped by the engine's own setters as long as
    # those setters are the ones called from here (eng._mirror is cleared by every OTHER slot-changing call: _touch)
    has_mirror = hasattr(eng, "_mirror")
    mirrors = eng._mirror.get(slot) if has_mirror else None
    if mirrors is None:
        mirrors = {"counts": [None] * C, "source": None}
    new = {"groups": list(old_groups), "counts": list(old_counts), "weights": old["weights"], "source": old["source"],
           "stale": set(old["stale"]), "lh_all": old.get("lh_all")}
    pending = ([], [])                              # count rows of all components: one set_counts_rows call
    try:
        return _bind_send(eng, slot, C, changed_sets=(conc_changed, groups_changed, counts_changed, weights_changed, source_changed),
                          toks=(conc, groups, counts, weights, source), new=new, mirrors=mirrors, pending=pending, cache=cache,
                          has_mirror=has_mirror)
    except BaseException:
        # _changed_rows copies the differing rows INTO the host mirror before anything is sent: if a setter raises in
        # between (a shape error, KeyboardInterrupt, a later component failing), mirror and entry would claim rows the
        # device never received and the next bind would send nothing -- forget the slot instead
        _forget_slot(eng, slot)
        raise


def _forget_slot(eng, slot):
    touch = getattr(eng, "_touch", None)
    if touch is not None:
        touch(slot)
    else:
        getattr(eng, "_bound", {}).pop(slot, None)
        getattr(eng, "_mirror", {}).pop(slot, None)


def _bind_send(eng, slot, C, changed_sets, toks, new, mirrors, pending, cache, has_mirror):
    """The send phase of _bind_slot (its own function so that a failure anywhere in it drops the slot's entry)."""
    conc_changed, groups_changed, counts_changed, weights_changed, source_changed = changed_sets
    conc, groups, counts, weights, source = toks
    for c in conc_changed:                          # (drops every slot's entry: all tables depend on it)
        eng.set_concentration(c, conc[c][0])
    if conc_changed:
        new["stale"] = set(range(C))
        new["lh_all"] = None
    # one changed group matrix, the changed count rows and the changed source rows of a bind go up in ONE launch when the
    # engine has the combined call (Engine.set_slot_delta) and at least two of them are there; else call by call, in this order
    groups_op = None
    for c in groups_changed:
        if len(groups_changed) == 1 and hasattr(eng, "set_slot_delta"):
            groups_op = (c, groups[c][0])
        else:
            eng.set_groups(slot, c, groups[c][0])
        new["groups"][c] = _remember(groups[c])
    was_stale = set(new["stale"])
    by_rows = []                                    # components whose change goes up as rows
    for c in counts_changed:
        n_pending = len(pending[0])
        mirrors["counts"][c], changed = _send_counts(eng, slot, c, counts[c][0], mirrors["counts"][c], pending)
        new["counts"][c] = _remember(counts[c])
        if changed:
            if len(pending[0]) > n_pending:
                by_rows.append(c)
            else:
                new["stale"].add(c)                 # (the component went up whole: every table row is stale)
            new["lh_all"] = None                    # (Likelihood._group_logliks' memo of the collapsed log-likelihoods)
    rows_op = None
    if pending[0]:
        # the probability rows of the patched groups are rebuilt by the same launch when the tables of those components
        # were current (the usual case inside a step: the slot held the sample this one was copied from) -- nothing is
        # left stale, _tables_current has no table kernel to run; otherwise the components join the stale set
        fused = _ROWS_WITH_PROBS and all(c not in was_stale for c in by_rows)
        idx = pending[0][0] if len(pending[0]) == 1 else np.concatenate(pending[0])
        rows = pending[1][0] if len(pending[1]) == 1 else np.concatenate(pending[1])
        rows_op = (idx, rows, fused)
        if not fused:
            new["stale"].update(by_rows)
    source_op = None
    if source_changed:
        mirrors["source"], source_op = _send_source(eng, slot, source, mirrors, new["source"])
        new["source"] = _remember(source)
    if (groups_op is not None) + (rows_op is not None) + (source_op is not None) >= 2:
        eng.set_slot_delta(slot,
                           groups_component=groups_op[0] if groups_op else 0, groups=groups_op[1] if groups_op else None,
                           count_idx=rows_op[0] if rows_op else None, count_rows=rows_op[1] if rows_op else None,
                           update_probs=bool(rows_op[2]) if rows_op else False,
                           source_objects=source_op[0] if source_op else None, source_rows=source_op[1] if source_op else None)
    else:
        if groups_op is not None:
            eng.set_groups(slot, groups_op[0], groups_op[1])
        if rows_op is not None:
            eng.set_counts_rows(slot, rows_op[0], rows_op[1], update_probs=True) if rows_op[2] else eng.set_counts_rows(slot, rows_op[0], rows_op[1])
        if source_op is not None:
            eng.set_source_rows(slot, source_op[0], source_op[1])
    if weights_changed:
        eng.set_weights(slot, weights[0])
        new["weights"] = _remember(weights)
    for c in conc_changed:
        eng._bound_conc[c] = _remember(conc[c])
    cache[slot] = new                               # (the setters above dropped the slot's entry)
    if has_mirror:
        eng._mirror[slot] = mirrors                 # (and its mirrors)
    return new["stale"]


_bind_slot_py = _bind_slot                      # the reference form: what the native function transcribes, and its fallback
if _fast.HAVE_EXTENSION and not _VERIFY_FROZEN:
    # binding._bind_slot in native code (csrc/sbe_pyhost.c: bind_slot): the same comparisons, engine calls and cache entries --
    # five binds per MCMC step, ~20 us each as Python (tools/host_residual.py); engines without a bind cache (test doubles)
    # are handed back to the Python form by the function itself
    _fast._h.bind_setup(_send_counts, _send_source, _remember, np.concatenate, _ROWS_WITH_PROBS, _bind_slot_py)
    _bind_slot = _fast._h.bind_slot


def counts_follow_plan(eng, sample, names, slot=0):
    """update_feature_counts (counts.py:55-95) is about to add a difference to `sample.feature_counts` on the host.  When
    engine slot `slot` holds exactly those counts -- the bind entry's token of every component is the parameter's current
    (array, version): the usual state inside an MCMC step, the slot was bound to the sample this one was copied from -- the
    device can take the difference in the call that computes it (Engine.counts_delta(follow_slot=...)) and the rows need
    not be sent back by the next bind.  Returns (entry, mirrors, stale components) or None."""
    bound = getattr(eng, "_bound", None)
    entry = bound.get(slot) if bound is not None else None
    mirrors = eng._mirror.get(slot) if entry is not None and hasattr(eng, "_mirror") else None
    if entry is None or mirrors is None:
        return None
    counts = sample.feature_counts
    for c, name in enumerate(names):
        cached = entry["counts"][c]
        if cached is None or mirrors["counts"][c] is None or not _same(_token(counts[name]), cached):
            return None
    return entry, mirrors, entry["stale"]


def _source_followed(entry, mirrors, known, parent_tok, sample, rows):
    """The slot's source rows `rows` were set to the sample's.  If the slot held `parent_tok` -- the source the sample's was derived
    from by editing exactly those rows -- it now holds the sample's source: the entry says so (the next bind with a source has
    nothing to compare), and the lineage is noted.  Otherwise the entry stays without a source token (content compare)."""
    mirrors["source_parent"] = None
    if parent_tok is None:
        return
    child = _token(sample.source)
    rec = note_source_lineage(parent_tok, child, rows)
    if known is None or known[1] is None or known[0] is not parent_tok[0] or known[1] != parent_tok[1]:
        return
    if rec is None:
        if child[1] is not None and child[0] is parent_tok[0] and child[1] == parent_tok[1]:
            entry["source"] = known                 # (the same parameter state: the rows that went up were its own)
        return
    entry["source"] = (child[0], child[1], None)
    mirrors["source_parent"] = (rec[2], rec[3], rec[4])


def counts_followed(eng, plan, sample, names, touched, bounds, probs_rebuilt, source_rows=None, slot=0, source_parent=None):
    """After that call and the host's own add_changes: the slot's entry (dropped by the engine method) comes back with the
    touched components' tokens and mirror rows brought to the sample's new counts; components whose probability rows were
    not rebuilt by the call join the stale set.  `source_rows` (object indices): the slot's source rows of those objects
    were set to the sample's in the same call -- the mirror takes them.  `source_parent`: the token (array, version) of the
    source parameter the sample's source was derived from by editing those rows; when the slot held exactly that, it now holds
    the sample's source and the entry records its token (_source_followed); otherwise the entry forgets which source
    parameter the slot held, so the next bind with a source compares content with the mirror (and sends whatever differs)."""
    entry, mirrors, _ = plan
    known = entry.get("source") if source_rows is not None else None
    if _fast._h is not None:                        # the same steps in one native call (csrc/sbe_pyhost.c: counts_followed)
        try:
            _fast._h.counts_followed(eng, entry, mirrors, [sample.feature_counts[name] for name in names], eng.group_offsets, touched, bounds,
                                     bool(probs_rebuilt), source_rows, sample.source.value if source_rows is not None else None, slot)
            if source_rows is not None:
                _source_followed(entry, mirrors, known, source_parent, sample, source_rows)
        except BaseException:                       # (half-updated mirrors must not come back into the cache)
            _forget_slot(eng, slot)
            raise
        return
    if source_rows is not None:
        src = sample.source.value
        mirrors["source"][source_rows] = src[source_rows] if src.dtype == np.bool_ else np.asarray(src[source_rows], dtype=bool)
        entry["source"] = None
        _source_followed(entry, mirrors, known, source_parent, sample, source_rows)
    counts = sample.feature_counts
    off = eng.group_offsets
    for c, name in enumerate(names):
        lo, hi = bounds[c], bounds[c + 1]
        if hi > lo:
            node = counts[name]
            local = touched[lo:hi] - off[c]
            mirrors["counts"][c][local] = node.value[local]
            entry["counts"][c] = _remember(_token(node))
            if not probs_rebuilt:
                entry["stale"].add(c)
    entry["lh_all"] = None
    eng._bound[slot] = entry
    eng._mirror[slot] = mirrors


def recount_bound(eng, sample, slot=0):
    """recalculate_feature_counts (counts.py:35-52) from RESIDENT data: bind the sample's groups and source (deltas
    only), recount every component on the device, fetch the tables.  The slot's bind entry survives: its count
    mirrors become the fetched tables, so the set_value() that follows on the host costs no upload.
    Returns the float32 [G_c, F, S] tables, component by component."""
    _bind_slot(eng, None, sample, slot, with_source=True)
    entry = eng._bound.get(slot) if hasattr(eng, "_bound") else None
    mirrors = eng._mirror.get(slot) if hasattr(eng, "_mirror") else None
    eng.recount(slot)                               # (drops the slot's entry and mirrors: the counts changed under them)
    tables = list(eng.get_counts_all(slot)) if hasattr(eng, "get_counts_all") else [eng.get_counts(slot, c) for c in range(eng.n_components)]
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
    arr = model.prior.prior_cluster_effect.uniform_concentration_array
    cached = getattr(eng, "_bound_unif", None)
    # the usual call: the very array the engine was given last time, recorded frozen (no private copy) and still frozen --
    # _same()'s identity rule, without building the token (three of these per MCMC step)
    if cached is not None and cached[0] is arr and cached[2] is None and cached[1] is None:
        flags = arr.flags
        if not flags.writeable and flags.owndata:
            if not _VERIFY_FROZEN:
                return
    unif = _token(np.asarray(arr))
    if not _same(unif, cached):
        eng.set_uniform_counts(unif[0])
        eng._bound_unif = _remember(unif)


def _tables_current(eng, slot):
    """Rebuild the probability tables that the last _bind_slot left stale."""
    entry = getattr(eng, "_bound", {}).get(slot)
    stale = set(range(eng.n_components)) if entry is None else entry["stale"]
    if not stale:
        return
    if len(stale) == 1:
        eng.update_probs(slot, next(iter(stale)))
    elif stale:
        eng.update_probs(slot, sorted(stale))      # one call (adjacent components: one launch)
    if entry is not None:
        entry["stale"] = set()
