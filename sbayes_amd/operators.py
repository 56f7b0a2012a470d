"""Data-parallel cores of the reference's operators on the engine.

  compute_cluster_posterior   AlterCluster.compute_cluster_posterior      operators.py:1035-1073
  compute_raw_cluster_probs   AlterClusterWide.compute_raw_cluster_probs  operators.py:1420-1472
                              (with the `gibbs` cluster-effect proposal,   operators.py:1254-1282)
  calculate_source_posterior  GibbsSampleSource.calculate_source_posterior operators.py:554-574
  gibbs_sample_source         GibbsSampleSource._propose                   operators.py:495-552
  component_likelihood_given_unchanged                                     operators.py:863-928
  jump_lh                     ClusterJump.get_jump_lh                      operators.py:1679-1722
                              (+ ClusterEffectProposals.expected_confounder_features, :1342-1379)
  source_lh_by_feature        GibbsSampleWeights.source_lh_by_feature      operators.py:677-685

The proposal logic, RNG and accept/reject stay the reference's (out of scope); these functions
return exactly what the reference methods return, so an operator can call them in place of its
own method body.  Everything array-sized runs on the device; the O(n_available) scalar tail
(temperature exponent, geo-prior factor, normalisation, smoothing) is host arithmetic as in the
reference.  The per-feature products are accumulated in log space on the device, so unlike the
reference (np.prod) large feature counts do not underflow.
"""
from __future__ import annotations

import numpy as np

from . import _fast
from . import counts as counts_mod
from .binding import _bind_slot, _bind_uniform, _engine, _tables_current, _token, counts_follow_plan, counts_followed, note_source_lineage
from .counts import _source_ids, apply_count_rows, note_jump_state, update_feature_counts

EPS = np.finfo(np.float32).eps        # sbayes/util.py:34


def _any_dynamic_priors(model, sample):
    """any(model.prior.prior_confounding_effects[conf].any_dynamic_priors for conf in sample.confounders) -- the flag is fixed when a
    prior is built (sbayes/model/prior.py:273-314), so the answer is kept on the model's likelihood object per prior table."""
    lik = getattr(model, "likelihood", None)
    priors = model.prior.prior_confounding_effects
    memo = lik.__dict__.get("_dynamic_priors_memo") if lik is not None and hasattr(lik, "__dict__") else None
    if memo is not None and memo[0] is priors and memo[1] == len(sample.confounders):
        return memo[2]
    flag = any(priors[conf].any_dynamic_priors for conf in sample.confounders)
    if lik is not None and hasattr(lik, "__dict__"):
        lik.__dict__["_dynamic_priors_memo"] = (priors, len(sample.confounders), flag)
    return flag


def _prepare(model, sample, slot):
    eng = _engine(model)
    _bind_slot(eng, model, sample, slot)
    _tables_current(eng, slot)
    return eng


def _log_marginals(eng, slot, table, available, prior_temperature):
    objects = np.flatnonzero(available) if np.asarray(available).dtype == np.bool_ else np.asarray(available)
    return eng.cluster_marginals(slot, table, objects, prior_temperature)


def compute_cluster_posterior(model, sample, i_cluster, available, temperature=1.0, prior_temperature=1.0,
                              additive_smoothing=1e-6, geo_likelihoods=None, slot=0):
    """Posterior probability of each available object to belong to cluster `i_cluster`."""
    eng = _prepare(model, sample, slot)
    _bind_uniform(eng, model)
    # the candidate table conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior, T) (operators.py:1046-1052)
    # is built on the device from the slot's resident counts: object ids up, 2 x n doubles back
    objects = np.flatnonzero(available) if np.asarray(available).dtype == np.bool_ else np.asarray(available)
    log_m = eng.cluster_posterior_marginals(slot, i_cluster, objects, temperature, prior_temperature) / temperature
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
    if temperature == 1.0:
        # the `gibbs` proposal's table is conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior, T) (operators.py:1254-1282):
        # at T = 1 exactly the candidate table sbe_cluster_posterior_marginals builds from the slot's resident counts -- one call,
        # the same kernel and the same bits as the explicit-table form below (tests/test_gpu_delta_forms.py)
        _bind_uniform(eng, model)
        objects = np.flatnonzero(available) if np.asarray(available).dtype == np.bool_ else np.asarray(available)
        log_m = eng.cluster_posterior_marginals(slot, i_cluster, objects, 1.0, prior_temperature)
    else:
        table = eng.normalize_tables(
            sample.feature_counts["clusters"].value[[i_cluster]], np.asarray(prior.concentration_array),
            temperature=temperature, prior_temperature=prior_temperature,
            unif_counts=np.asarray(prior.uniform_concentration_array))
        table = table ** (1 / temperature)          # inner1d(features, p) ** (1/T): elementwise on the table
        log_m = _log_marginals(eng, slot, table, available, prior_temperature) / temperature
    m = np.exp(log_m)
    if geo_prior_ratio is not None:
        m[1] *= geo_prior_ratio
    return m[1] / (m[0] + m[1] + EPS)


def cluster_log_marginals(model, sample, table, available, temperature=1.0, prior_temperature=1.0, slot=0):
    """float64 [2, n_available]: log of AlterClusterWide's `marginal_lh_z01` (operators.py:1444-1451) for a candidate
    cluster table `table` [1, F, S] float32 supplied by the caller (any of the reference's ClusterEffectProposals):
    log prod_f (sum_c lh ** (1/T) ... ) ** (1/T), evaluated as sums of logs on the device."""
    eng = _prepare(model, sample, slot)
    table = np.asarray(table, dtype=np.float32)
    if temperature != 1.0:
        table = table ** (1 / temperature)          # inner1d(features, p) ** (1/T): elementwise on the table
    return _log_marginals(eng, slot, table, available, prior_temperature) / temperature


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


def gibbs_sample_source(model, sample, object_subset=slice(None), temperature=1.0, prior_temperature=1.0,
                        sample_from_prior=False, z=None, slots=(0, 1), device_rng=False):
    """GibbsSampleSource._propose (operators.py:495-552) on the device: posterior, draw, new source rows,
    delta counts, new tables and both transition log-probabilities never leave the GPU; what crosses
    PCIe is the uniforms in and [n, F] selected probabilities + the subset's new rows out.

    The uniforms are drawn exactly where the reference draws them -- np.random.random([n, F, 1]) inside
    sample_categorical (preprocessing.py:248) -- so with the same np.random state the proposal is the
    reference's, draw for draw; pass `z` [n, F] to supply them.  log_q / log_q_back are summed on the
    host from the selected float32 probabilities, in the reference's float32 precision.
    device_rng=True takes the uniforms from the engine's Philox stream instead (Engine.set_rng): nothing
    but the object ids goes up; statistically equivalent, not the same numbers.
    Returns (sample_new, log_q, log_q_back) like the reference."""
    eng = _engine(model)
    cur, new = slots
    _bind_slot(eng, model, sample, cur, with_source=True)
    _tables_current(eng, cur)
    n_objects = sample.n_objects
    if isinstance(object_subset, slice):
        objects = np.arange(n_objects)[object_subset]
    else:
        objects = np.asarray(object_subset)
        if objects.dtype == np.bool_:
            objects = np.flatnonzero(objects)
    if device_rng:
        z = None
    else:
        if z is None:
            z = np.random.random([objects.size, eng.n_features, 1])
        z = np.asarray(z, dtype=np.float64).reshape(objects.size, eng.n_features)
    if z is not None and objects.size and getattr(eng, "gibbs_propose_supported", lambda: False)() and \
            (objects.size < 2 or np.unique(objects).size == objects.size):
        # the whole proposal in ONE engine call (sbe_gibbs_propose): drawn ids, both selected-probability arrays and the
        # count rows that changed come back together; the sample bookkeeping below is the reference's
        # ... and the slot takes the proposal on the device (it holds `sample`, bound above, tables current): the bind of the
        # sample built below has nothing to send (binding.counts_follow_plan)
        names = ["clusters", *sample.confounders]
        plan = counts_follow_plan(eng, sample, names, cur) if counts_mod.FOLLOW_COUNTS and hasattr(eng, "_bound") else None
        if plan is not None and (plan[2] or plan[1]["source"] is None or plan[1]["source"].shape != np.shape(sample.source.value)):
            plan = None
        if plan is not None:
            ids, sel, sel_back, touched, rows = eng.gibbs_propose(cur, new, objects, z, temperature, prior_temperature, sample_from_prior,
                                                                  follow=True)
        else:
            ids, sel, sel_back, touched, rows = eng.gibbs_propose(cur, new, objects, z, temperature, prior_temperature, sample_from_prior)
        valid = ids != 255                                   # (= ~na_features[objects]: NA observations get no component)
        with np.errstate(divide="ignore"):
            log_q = np.log(sel[valid]).sum()
            log_q_back = np.log(sel_back[valid]).sum()
        sample_new = sample.copy()
        sample_new.source.set_groups(object_subset, ids[..., None] == np.arange(eng.n_components, dtype=np.uint8))
        bounds = apply_count_rows(sample_new.feature_counts, names, eng.group_offsets, touched, rows, return_bounds=True)
        if plan is not None:
            if touched.size:                                 # (the engine's rule: nothing touched, nothing follows)
                counts_followed(eng, plan, sample_new, names, touched, bounds, True, objects, cur, source_parent=_token(sample.source))
            else:
                eng._bound[cur], eng._mirror[cur] = plan[0], plan[1]
        return sample_new, log_q, log_q_back
    eng.copy_slot(new, cur)
    _, sel = eng.sample_source(cur, new, objects, z, temperature, prior_temperature, sample_from_prior,
                               return_selected=True)
    eng.update_counts(new, cur, objects)
    eng.update_probs(new, range(eng.n_components))          # (one call: adjacent components, one launch)
    _, sel_back = eng.source_logprob(new, cur, objects, temperature, prior_temperature, sample_from_prior,
                                     return_selected=True)
    valid = ~eng.na_values()[objects]
    with np.errstate(divide="ignore"):
        log_q = np.log(sel[valid]).sum()
        log_q_back = np.log(sel_back[valid]).sum()

    sample_new = sample.copy()
    sample_new.source.set_groups(object_subset, eng.get_source_rows(new, objects))
    # the count delta exactly as the reference applies it (update_feature_counts -> add_changes per component): one
    # stateless call that returns the rows of the groups the subset's objects are in -- not 2 C whole tables
    update_feature_counts(sample, sample_new, model.data.features.values, object_subset)
    return sample_new, log_q, log_q_back


def _subset_as_sorted_ids(object_subset, n_objects):
    """int32 ids, ascending and distinct, of the objects `np.isin(np.arange(n_objects), object_subset)` marks
    (ClusterOperator.gibbs_sample_source, operators.py:805) -- through a flag array instead of the sort-based set
    operation (42 us per call on the hosts measured); values outside [0, n_objects) match nothing, as there."""
    subset = np.asarray(object_subset)
    if subset.dtype == np.bool_ and subset.shape == (n_objects,):
        return np.flatnonzero(subset).astype(np.int32)
    subset = subset.reshape(-1)
    if subset.dtype.kind not in "iu":
        return np.flatnonzero(np.isin(np.arange(n_objects), subset)).astype(np.int32)
    if subset.size and (int(subset.min()) < 0 or int(subset.max()) >= n_objects):
        subset = subset[(subset >= 0) & (subset < n_objects)]
    mask = np.zeros(n_objects, dtype=np.bool_)
    mask[subset] = True
    return np.flatnonzero(mask).astype(np.int32)


def cluster_gibbs_sample_source(model, sample_new, sample_old, i_cluster, object_subset, temperature=1.0, prior_temperature=1.0,
                                sample_from_prior=False, slot=0, z=None):
    """ClusterOperator.gibbs_sample_source (operators.py:796-851): the source resampling inside every AlterCluster /
    AlterClusterWide / ClusterJump proposal.  `sample_new` has the clusters already changed and the source not yet
    resampled (its counts are still the old state's); everything between the two samples' bookkeeping -- the likelihood
    under the kept observations, both posteriors, the draw, the selected probabilities -- runs on the device in one call
    (sbe_given_unchanged_gibbs): the object list, 2 n C has_components bytes, n F old source ids and the n F uniforms
    (np.random.random((n, F, 1)), drawn exactly where the reference's sample_categorical draws them) go up, n F drawn ids
    and 2 n F float32 come back.  The sample edits (source.edit(), update_feature_counts) and the float32 sums of logs are
    the reference's own, in its order.  Returns (sample_new, log_q, log_q_back).  Static priors only: None otherwise
    (the caller keeps the reference's method)."""
    if _any_dynamic_priors(model, sample_new):
        return None
    eng = _engine(model)
    features = model.data.features.values
    na_features = model.data.features.na_values
    objects = _subset_as_sorted_ids(object_subset, sample_new.n_objects)   # (operators.py:805: np.isin(arange(N), subset))
    _bind_slot(eng, model, sample_new, slot, with_source=True)
    _bind_uniform(eng, model)
    hc_new = sample_new.cache.has_components.value[objects]
    hc_old = sample_old.cache.has_components.value[objects]
    # the subset's group ids in both samples and its old source ids, one pass of the host helper; the proposal's count
    # delta (update_feature_counts, operators.py:827) then comes out of the SAME engine call as the draw -- None: an
    # object in several groups of a component has no single id, the counts go through update_feature_counts below
    conf_names = list(sample_new.confounders)
    groups_old = [sample_old.clusters.value] + [sample_old.confounders[k].group_assignment for k in conf_names]
    groups_new = [sample_new.clusters.value] + [sample_new.confounders[k].group_assignment for k in conf_names]
    source_old = sample_old.source.value
    sub = _fast.subset_ids(objects, groups_new, groups_old, source_old, source_old) if objects.size else None
    src_old = sub[2] if sub is not None else _source_ids(source_old, objects)
    if z is None:                                                          # (tests pass the reference's recorded uniforms)
        z = np.random.random((objects.size, eng.n_features, 1))
    plan = None
    if sub is not None:
        # the slot holds `sample_new` as it is now (bound above: its counts are the ones the delta belongs to): it takes the
        # proposal on the device -- counts, the touched groups' probability rows when no table is stale, the drawn source rows --
        # and the binds that follow the bookkeeping below have nothing to send (binding.counts_follow_plan)
        names = ["clusters", *conf_names]
        plan = counts_follow_plan(eng, sample_new, names, slot) if counts_mod.FOLLOW_COUNTS and hasattr(eng, "_bound") else None
        if plan is not None and (plan[1]["source"] is None or plan[1]["source"].shape != np.shape(source_old)):
            plan = None
        if plan is not None:
            rebuild = not plan[2]
            ids, sel_new, sel_back, touched, rows = eng.given_unchanged_gibbs(
                slot, i_cluster, objects, hc_new, hc_old, src_old, z, temperature, prior_temperature, sample_from_prior,
                gid_old=sub[0], gid_new=sub[1], follow=True, update_probs=rebuild)
        else:
            ids, sel_new, sel_back, touched, rows = eng.given_unchanged_gibbs(slot, i_cluster, objects, hc_new, hc_old, src_old, z, temperature,
                                                                              prior_temperature, sample_from_prior, gid_old=sub[0], gid_new=sub[1])
    else:
        ids, sel_new, sel_back = eng.given_unchanged_gibbs(slot, i_cluster, objects, hc_new, hc_old, src_old, z, temperature,
                                                           prior_temperature, sample_from_prior)
    x = ids[..., None] == np.arange(eng.n_components, dtype=np.uint8)     # one-hot; all False where NA (id 255)
    src_parent = _token(sample_new.source)                                # (what the slot was bound to above: binding.py, source lineage)
    with sample_new.source.edit() as source:
        source[objects] = x                                               # (NA observations stay 0: operators.py:825)
    if sub is not None:
        bounds = apply_count_rows(sample_new.feature_counts, ["clusters", *conf_names], eng.group_offsets, touched, rows, return_bounds=True)
        if plan is not None:
            if touched.size:                                               # (the engine's rule: nothing touched, nothing follows)
                counts_followed(eng, plan, sample_new, names, touched, bounds, rebuild, objects, slot, source_parent=src_parent)
            else:                                                          # (the call dropped the entry: it still describes the slot)
                eng._bound[slot], eng._mirror[slot] = plan[0], plan[1]
        else:
            note_source_lineage(src_parent, _token(sample_new.source), objects)
    else:
        update_feature_counts(sample_old, sample_new, features, objects)
    valid = ~na_features[objects]
    with np.errstate(divide="ignore"):
        log_q = np.log(sel_new[valid]).sum()                              # float32 logs, float32 sum (operators.py:832)
        log_q_back = np.log(sel_back[valid & (src_old != 255)]).sum()     # (operators.py:847)
    return sample_new, log_q, log_q_back


def component_likelihood_given_unchanged(model, sample, object_subset, i_cluster, temperature=1.0,
                                         prior_temperature=1.0, slot=0):
    """operators.py:863-928: float32 [n_subset, F, C] component likelihoods of the subset's
    observations under effect tables built only from the observations that are NOT resampled.
    `object_subset` is a bool mask [n_objects].  Counts (a9) and tables (a10) come from the device."""
    eng = _engine(model)
    object_subset = np.asarray(object_subset, dtype=bool)
    objects = np.flatnonzero(object_subset)
    if not _any_dynamic_priors(model, sample):
        # static priors: everything the reference reads here is resident once `sample` is bound -- its (new) clusters,
        # its not-yet-resampled source, its counts, the priors' tables -- so the kept / unchangeable counts
        # (operators.py:876-901), their tempered tables and the gather run on the device in ONE call; the object list
        # goes up, [n, F, C] float32 comes back
        _bind_slot(eng, model, sample, slot, with_source=True)
        _bind_uniform(eng, model)
        return eng.given_unchanged_lh(slot, i_cluster, objects, temperature, prior_temperature)
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


def jump_lh(model, sample, i_source_cluster, i_target_cluster, temperature=1.0, prior_temperature=1.0, slot=0):
    """ClusterJump.get_jump_lh (operators.py:1679-1722): float32 [n_members] probability-like score
    lh_jump / (lh_jump + lh_stay) of every member of the source cluster.  The tempered effect tables (a10) and the
    per-member sums of logs come from the device -- no [N, F, C] weight array, no [N, F, S] expected-feature array on
    the host; the O(n_members) tail (exponent 1/T, + EPS, ratio) is the reference's float32 arithmetic."""
    eng = _engine(model)
    note_jump_state(sample)            # (counts.py: the jump's update_feature_counts leaves the slot's counts alone)
    _bind_slot(eng, model, sample, slot)
    _bind_uniform(eng, model)          # the reference uses the CLUSTER prior's uniform concentration for every component
    members = np.flatnonzero(sample.clusters.value[i_source_cluster])                               # (operators.py:1352)
    logs = eng.jump_lh_resident(slot, i_source_cluster, i_target_cluster, members, temperature, prior_temperature)
    with np.errstate(under="ignore"):
        lh_stay = np.exp(logs[0]).astype(np.float32)             # np.prod over features in float32 (operators.py:1707-1710)
        lh_jump = np.exp(logs[1]).astype(np.float32)
        lh_stay **= (1 / temperature)
        lh_jump **= (1 / temperature)
    lh_stay += EPS
    lh_jump += EPS
    return lh_jump / (lh_jump + lh_stay)


def source_lh_by_feature(model, sample, slot=0):
    """GibbsSampleWeights.source_lh_by_feature (operators.py:677-685) of `sample`'s source assignment under its
    normalised weights: float32 [n_features], computed on the device from resident data."""
    eng = _engine(model)
    _bind_slot(eng, model, sample, slot, with_source=True)
    return eng.source_lh_by_feature(slot)
