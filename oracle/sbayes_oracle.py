"""CPU oracle for the sBayes likelihood hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a NumPy restatement of the reference algorithm (NicoNeureiter/sBayes,
snapshot 2025-06-20) for the path named in BASELINE.json `north_star`.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
only as the checker / the timed CPU baseline -- never as the thing shipped.  The product
(`sbayes_amd/`) never imports this module and fails loudly when its HIP library is missing.

Parity status: PINNED.  The reference has no executed test that pins these numbers
(`test/test_model.py:96-207` is commented out), so the oracle is pinned against outputs
of the reference itself, generated in the build container by
`tests/golden/make_golden.py` (stub-imported reference, numba decorators = identity, i.e.
the NumPy path BASELINE.json names) and committed as `tests/golden/*.npz`.
`tests/test_oracle_golden.py` checks every function below against those vectors, plus the
hand-derivable known answers of the reference's commented-out test
(`test/test_model.py:157-207`).

Every function cites the reference lines it restates.  Float32 rounding points of the
reference are kept exactly: probability tables and normalised weights are
rounded to float32, everything downstream of them is float64; the collapsed likelihood is
float32 per feature and float32-summed per group.
"""
from __future__ import annotations

import numpy as np
from scipy.special import gammaln as _gammaln

FLOAT_TYPE = np.float32  # sbayes/util.py:32


# --------------------------------------------------------------------------------------
# a4  normalize                                                   sbayes/util.py:990-1007
# --------------------------------------------------------------------------------------
def normalize(x, axis=-1):
    """x / sum(x, axis) computed in x's dtype, THEN rounded to float32 (util.py:1006-1007).
    The reference asserts that all sums are positive (util.py:1006)."""
    x = np.asarray(x)
    total = np.sum(x, axis=axis, keepdims=True)
    if not np.all(total > 0):
        raise AssertionError(float(np.min(x)))
    return (x / total).astype(FLOAT_TYPE)


# --------------------------------------------------------------------------------------
# a1  compute_component_likelihood                     sbayes/model/likelihood.py:104-133
# --------------------------------------------------------------------------------------
def compute_component_likelihood(features, probs, groups, changed_groups, out):
    """For each changed group i: out[members(i), :] = sum_s features[members, :, s] * probs[i, :, s].
    Rows of objects that are in no group are zeroed first (likelihood.py:121-123); rows of
    members of groups that are NOT in `changed_groups` keep whatever `out` held
    (likelihood.py:126-130).  Later groups overwrite earlier ones.  Returns `out`."""
    in_any = groups.sum(axis=0) != 0
    out[~in_any, :] = 0.0
    for i in changed_groups:
        members = groups[i]
        one_hot = features[members]                    # bool [n_g, F, S]
        table = probs[i]                               # [F, S]
        out[members, :] = (one_hot * table[None, :, :]).sum(axis=-1)
    return out


# --------------------------------------------------------------------------------------
# a2  compute_component_likelihood_exact               sbayes/model/likelihood.py:136-150
# --------------------------------------------------------------------------------------
def compute_component_likelihood_exact(features, probs, groups, changed_groups, out):
    """Same as a1 but probs[i] is a per-member table [n_in_group, F, S] (leave-one-out
    tables built by likelihood_per_component_exact)."""
    out[~groups.any(axis=0), :] = 0.0
    for i in changed_groups:
        members = groups[i]
        out[members, :] = np.einsum("nfs,nfs->nf", features[members], probs[i])
    return out


# --------------------------------------------------------------------------------------
# a5  normalize_weights / has_components   likelihood.py:171-190, sampling/state.py:353-376
# --------------------------------------------------------------------------------------
def has_components(groups_by_component):
    """bool [N, C]: object n is affected by component c iff it is in any group of c
    (state.py:358-362: clusters.any_cluster() and conf.any_group())."""
    return np.stack([np.any(g, axis=0) for g in groups_by_component], axis=1)


def normalize_weights(weights, has_comp):
    """Mask the per-feature mixture weights [F, C] with each object's has_components row and
    renormalise over components -> [N, F, C].  Computed once per distinct row pattern and
    broadcast through the inverse index (likelihood.py:183-190).  dtype follows `weights`
    (float32 in the sampler: state.py:546)."""
    pattern, inverse = np.unique(has_comp, axis=0, return_inverse=True)
    w = pattern[:, None, :] * weights[None, :, :]
    w /= w.sum(axis=-1, keepdims=True)
    return w[np.asarray(inverse).reshape(-1)]


def weight_patterns(weights, has_comp):
    """The factorised form of normalize_weights: (w_pat [P, F, C], pattern_id [N]).  Same
    arithmetic as above, without the broadcast (this is what crosses the C-ABI)."""
    pattern, inverse = np.unique(has_comp, axis=0, return_inverse=True)
    w = pattern[:, None, :] * weights[None, :, :]
    w /= w.sum(axis=-1, keepdims=True)
    return w, np.asarray(inverse).reshape(-1)


# --------------------------------------------------------------------------------------
# a9  feature counts                                       sbayes/sampling/counts.py:10-95
# --------------------------------------------------------------------------------------
def compute_effect_counts(features, group_assignment, source_is_component, object_subset=slice(None)):
    """counts[g, f, s] = #{n in g (and in object_subset): source_is_component[n, f] and
    features[n, f, s]} as float32 (counts.py:20, 28-30)."""
    group_assignment = group_assignment[:, object_subset]
    n_groups = group_assignment.shape[0]
    _, n_features, n_states = features.shape
    counts = np.zeros((n_groups, n_features, n_states), dtype=FLOAT_TYPE)
    if not group_assignment.any():
        return counts
    src = source_is_component[object_subset]
    feat = features[object_subset]
    for i_g in range(n_groups):
        members = group_assignment[i_g]
        if members.any():
            counts[i_g] = np.count_nonzero(src[members][:, :, None] & feat[members], axis=0)
    return counts


def recalculate_feature_counts(features, groups_by_component, source):
    """counts.py:35-52: one compute_effect_counts per mixture component; component c uses
    source[..., c]."""
    return [
        compute_effect_counts(features, groups, source[..., c])
        for c, groups in enumerate(groups_by_component)
    ]


def update_feature_counts(counts, features, groups_old, groups_new, source_old, source_new, object_subset):
    """counts.py:55-95: delta update restricted to `object_subset`; returns new count arrays
    and, per component, the bool mask of groups whose counts changed (state.py:349-350)."""
    new_counts, changed = [], []
    for c in range(len(counts)):
        old_c = compute_effect_counts(features, groups_old[c], source_old[..., c], object_subset)
        new_c = compute_effect_counts(features, groups_new[c], source_new[..., c], object_subset)
        diff = new_c - old_c
        new_counts.append(counts[c] + diff)
        changed.append(np.any(diff != 0, axis=(1, 2)))
    return new_counts, changed


# --------------------------------------------------------------------------------------
# a8  dirichlet_categorical_logpdf                             sbayes/util.py:1373-1394
# --------------------------------------------------------------------------------------
def dirichlet_categorical_logpdf(counts, a):
    """Per feature: lgamma(sum a) - lgamma(n + sum a) + sum_{s: a>0} (lgamma(c + a) - lgamma(a)),
    rounded to float32 (util.py:1390-1394).  lgamma(0) = +inf in the dead lanes is masked
    by the `a > 0` select."""
    counts = np.asarray(counts)
    a = np.asarray(a)
    n = counts.sum(axis=-1)
    sum_a = a.sum(axis=-1)
    const = _gammaln(sum_a) - _gammaln(n + sum_a)
    with np.errstate(invalid="ignore"):
        series = np.where(a > 0, _gammaln(counts + a) - _gammaln(a), 0.0)
    return (const + series.sum(axis=-1)).astype(FLOAT_TYPE)


# --------------------------------------------------------------------------------------
# a7  Likelihood.__call__ (collapsed form)              sbayes/model/likelihood.py:47-101
# --------------------------------------------------------------------------------------
def collapsed_group_logliks(counts, concentration):
    """float64 [G]: per group the float32 `.sum()` over features of a8 (likelihood.py:74-77,
    95-99), stored into the float64 per-group cache (state.py:401-404).
    `concentration` is [F, S] (clusters: prior.py:453-455) or [G, F, S] (confounders:
    prior.py:285)."""
    n_groups = counts.shape[0]
    out = np.empty(n_groups, dtype=np.float64)
    for g in range(n_groups):
        a = concentration if concentration.ndim == 2 else concentration[g]
        out[g] = dirichlet_categorical_logpdf(counts[g], a).sum()
    return out


def collapsed_loglik(counts_by_component, concentration_by_component):
    """Likelihood.__call__: sum over components of the float64 sum over groups of the
    per-group values (likelihood.py:58-63, 79, 101)."""
    log_lh = 0.0
    for counts, conc in zip(counts_by_component, concentration_by_component):
        log_lh += collapsed_group_logliks(counts, conc).sum()
    return log_lh


# --------------------------------------------------------------------------------------
# a10 conditional_effect_mean                       sbayes/sampling/conditionals.py:105-122
# --------------------------------------------------------------------------------------
def conditional_effect_mean(prior_counts, feature_counts, unif_counts=None,
                            prior_temperature=None, temperature=None):
    if prior_temperature is not None:
        assert unif_counts is not None
        prior_counts = unif_counts + (prior_counts - unif_counts) / prior_temperature
    if temperature is not None:
        feature_counts = feature_counts / temperature
    return normalize(feature_counts + prior_counts, axis=-1)


# --------------------------------------------------------------------------------------
# a3  likelihood_per_component                     sbayes/sampling/conditionals.py:152-223
# --------------------------------------------------------------------------------------
def component_probs(counts, concentration):
    """normalize(counts + prior) -> float32 [G, F, S] (conditionals.py:175-179, 200-204).
    A 2-D concentration [F, S] broadcasts over groups (cluster effect prior)."""
    return normalize(counts + concentration, axis=-1)


def likelihood_per_component(features, na_values, groups_by_component, counts_by_component,
                             concentration_by_component, changed_by_component=None, out=None):
    """float64 [N, F, C].  Component c's slice is written by a1 through the strided view
    out[..., c] (conditionals.py:182-188, 208-214); finally NA observations are set to 1 in
    every component (conditionals.py:216).  `changed_by_component=None` means caching=False
    (all groups, state.py:251-252); components whose changed list is empty are skipped
    (conditionals.py:173, 196-197)."""
    n_objects, n_features, _ = features.shape
    n_comp = len(groups_by_component)
    if out is None:
        out = np.empty((n_objects, n_features, n_comp), dtype=np.float64)
    for c in range(n_comp):
        groups = groups_by_component[c]
        changed = np.arange(groups.shape[0]) if changed_by_component is None else changed_by_component[c]
        if len(changed) == 0:
            continue
        probs = component_probs(counts_by_component[c], concentration_by_component[c])
        compute_component_likelihood(features, probs, groups, changed, out[..., c])
    out[na_values] = 1.0
    return out


def likelihood_per_component_exact(features, na_values, groups_by_component, counts_by_component,
                                   concentration_by_component, source):
    """conditionals.py:300-367: leave-one-out tables -- each member's own observation
    (where its source is this component) is subtracted from the group's posterior counts
    before normalising."""
    n_objects, n_features, _ = features.shape
    n_comp = len(groups_by_component)
    out = np.empty((n_objects, n_features, n_comp), dtype=np.float64)
    for c in range(n_comp):
        groups = groups_by_component[c]
        post = counts_by_component[c] + concentration_by_component[c]
        tables = []
        for g in range(groups.shape[0]):
            members = groups[g]
            own = features[members] * source[members, :, c, None]
            tables.append(normalize(post[None, g, :, :] - own, axis=-1))
        compute_component_likelihood_exact(features, tables, groups, np.arange(groups.shape[0]), out[..., c])
    out[na_values] = 1.0
    return out


# --------------------------------------------------------------------------------------
# a6 + SURVEY.md 8(d): the mixture log-likelihood scalar ("one eval")
#     loggers.py:355-357 / operators.py:568-574 / operators.py:1060-1061
# --------------------------------------------------------------------------------------
def mixture_observation_lh(weights_normalized, component_lh):
    """float64 [N, F]: sum_c w[n,f,c] * lh[n,f,c] (float32 x float64 -> float64)."""
    return np.sum(weights_normalized * component_lh, axis=-1)


def mixture_loglik(features, na_values, groups_by_component, counts_by_component,
                   concentration_by_component, weights):
    """One uncached eval of LL = sum_{n,f not NA} log sum_c w[n,f,c] p_c[g_c(n), f, x(n,f)].
    Reference composition:
      np.log(np.sum(update_weights(s, caching=False) *
                    likelihood_per_component(model, s, caching=False), axis=-1))[~na].sum()"""
    lh = likelihood_per_component(features, na_values, groups_by_component,
                                  counts_by_component, concentration_by_component)
    w = normalize_weights(weights, has_components(groups_by_component))
    with np.errstate(divide="ignore"):
        return np.log(mixture_observation_lh(w, lh))[~na_values].sum()


# --------------------------------------------------------------------------------------
# cluster-membership posterior
#     sbayes/sampling/operators.py:1035-1095 (AlterCluster.compute_cluster_posterior,
#     compute_feature_weights_with_and_without), :1420-1472 (AlterClusterWide.compute_raw_cluster_probs)
# --------------------------------------------------------------------------------------
def feature_weights_with_and_without(weights, has_comp, available, prior_temperature=1.0):
    """float64 [2, n_available, F, C]: normalised weights of every available object if it were
    outside (z=0) / inside (z=1) a cluster (operators.py:1075-1095)."""
    w_cur = normalize_weights(weights, has_comp)[available]
    w_cur = normalize(w_cur ** (1 / prior_temperature), axis=-1)
    hc = has_comp[available].copy()
    hc[:, 0] = ~hc[:, 0]
    w_flip = normalize_weights(weights ** (1 / prior_temperature), hc)
    out = np.empty((2, *w_cur.shape))
    out[1] = np.where(hc[:, np.newaxis, [0]], w_flip, w_cur)
    out[0] = np.where(hc[:, np.newaxis, [0]], w_cur, w_flip)
    return out


def cluster_marginals(features, na_values, lh_per_component, table, available, weights_z01, lh_exponent=None):
    """float64 [2, n_available]: prod_f sum_c lh[n,f,c] * w_z[n,f,c] with the cluster component's
    likelihood replaced by the candidate table's (operators.py:1054-1061 / 1444-1451)."""
    cluster_lh = np.einsum("...i,...i", features[available], table)
    if lh_exponent is not None:
        cluster_lh = cluster_lh ** lh_exponent
    all_lh = lh_per_component[available].copy()
    all_lh[..., 0] = cluster_lh
    all_lh[na_values[available], 0] = 1.0
    feature_lh = np.einsum("...i,...i", all_lh[np.newaxis, ...], weights_z01)
    return np.prod(feature_lh, axis=-1)


def cluster_posterior(features, na_values, groups_by_component, counts_by_component, concentration_by_component,
                      weights, i_cluster, available, unif_counts, temperature=1.0, prior_temperature=1.0,
                      additive_smoothing=1e-6, geo_likelihoods=None):
    """AlterCluster.compute_cluster_posterior (gibbsish, operators.py:1035-1073)."""
    table = conditional_effect_mean(concentration_by_component[0], counts_by_component[0][[i_cluster]],
                                    unif_counts=unif_counts, prior_temperature=prior_temperature,
                                    temperature=temperature)
    lh = likelihood_per_component(features, na_values, groups_by_component, counts_by_component,
                                  concentration_by_component)
    wz = feature_weights_with_and_without(weights, has_components(groups_by_component), available, prior_temperature)
    m = cluster_marginals(features, na_values, lh, table, available, wz) ** (1 / temperature)
    if geo_likelihoods is not None:
        m[1] *= geo_likelihoods
    post = m[1] / (m[0] + m[1])
    if additive_smoothing > 0:
        post = (post + additive_smoothing) / (1 + 2 * additive_smoothing)
    return post


def wide_raw_cluster_probs(features, na_values, groups_by_component, counts_by_component, concentration_by_component,
                           weights, i_cluster, available, unif_counts, temperature=1.0, prior_temperature=1.0,
                           geo_prior_ratio=None):
    """AlterClusterWide.compute_raw_cluster_probs with the `gibbs` effect proposal
    (operators.py:1254-1282, 1420-1472)."""
    eps = np.finfo(FLOAT_TYPE).eps
    c = unif_counts + (concentration_by_component[0] - unif_counts) / prior_temperature \
        + counts_by_component[0][[i_cluster]] / temperature
    table = normalize(c, axis=-1)
    lh = likelihood_per_component(features, na_values, groups_by_component, counts_by_component,
                                  concentration_by_component)
    wz = feature_weights_with_and_without(weights, has_components(groups_by_component), available, prior_temperature)
    m = cluster_marginals(features, na_values, lh, table, available, wz, lh_exponent=1 / temperature) ** (1 / temperature)
    if geo_prior_ratio is not None:
        m[1] *= geo_prior_ratio
    return m[1] / (m[0] + m[1] + eps)


# --------------------------------------------------------------------------------------
# data-parallel cores of Gibbs source resampling
#     operators.py:554-574 (calculate_source_posterior), :863-928 (component_likelihood_given_unchanged)
# --------------------------------------------------------------------------------------
def source_posterior(lh_per_component, weights_normalized, object_subset, temperature=1.0, prior_temperature=1.0):
    """float32 [n_subset, F, C]: normalize(lh ** (1/T) * w ** (1/T_prior)) over components."""
    sp = (lh_per_component[object_subset] ** (1 / temperature)
          * weights_normalized[object_subset] ** (1 / prior_temperature))
    return normalize(sp, axis=-1)


def component_likelihood_given_unchanged(features, na_values, groups_by_component, counts_by_component,
                                         concentration_by_component, source, object_subset, i_cluster,
                                         unif_cluster, unif_confounders, temperature=1.0, prior_temperature=1.0):
    """float32 [n_subset, F, C]: component likelihoods of the subset's observations under tables
    built only from the observations that are NOT being resampled (object_subset: bool [N])."""
    n_sub = np.count_nonzero(object_subset)
    n_comp = len(groups_by_component)
    lik = np.zeros((n_sub, features.shape[1], n_comp), dtype=FLOAT_TYPE)
    cluster = groups_by_component[0][i_cluster]
    cluster_features = features * source[:, :, 0, None]
    cluster_effect = conditional_effect_mean(
        prior_counts=concentration_by_component[0],
        feature_counts=np.sum(cluster_features[cluster & ~object_subset], axis=0),
        unif_counts=unif_cluster, prior_temperature=prior_temperature, temperature=temperature)
    lik[..., 0] = np.sum(cluster_effect[None, ...] * features[object_subset], axis=-1)
    for c in range(1, n_comp):
        groups = groups_by_component[c]
        feats_c = features * source[:, :, c, None]
        changeable = np.array([np.sum(feats_c[g & object_subset], axis=0) for g in groups])
        effect = conditional_effect_mean(
            prior_counts=concentration_by_component[c], feature_counts=counts_by_component[c] - changeable,
            unif_counts=unif_confounders[c - 1], prior_temperature=prior_temperature, temperature=temperature)
        sub_groups = groups[:, object_subset]
        in_subset = np.any(sub_groups, axis=1)
        feats_sub = features[object_subset]
        for g, p_g in zip(sub_groups[in_subset], effect[in_subset]):
            lik[g, :, c] = np.einsum("ijk,jk->ij", feats_sub[g], p_g)
    lik[na_values[object_subset]] = 1.0
    return lik ** (1 / temperature)


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
    Random123's philox4x32 with 10 rounds).  counter uint32 [..., 4], key uint32 [..., 2] -> uint32 [..., 4].
    The engine's optional device stream for the Gibbs source draw (not part of the reference, which uses
    np.random): restated here so the device draws can be checked number for number."""
    c = [np.asarray(counter, dtype=np.uint64)[..., i].copy() for i in range(4)]
    k = [np.asarray(key, dtype=np.uint64)[..., i].copy() for i in range(2)]
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k[0], p1 & m32, (p0 >> np.uint64(32)) ^ c[3] ^ k[1], p0 & m32]
        k = [(k[0] + np.uint64(0x9E3779B9)) & m32, (k[1] + np.uint64(0xBB67AE85)) & m32]
    return np.stack(c, axis=-1).astype(np.uint32)


def philox_uniforms(seed, draw, n):
    """The engine's uniform i (0 <= i < n) of draw `draw` under `seed`: counter (i_lo, i_hi, draw_lo,
    draw_hi), key (seed_lo, seed_hi); 53 bits from the first two words like MT19937's genrand_res53."""
    i = np.arange(n, dtype=np.uint64)
    ctr = np.stack([i & np.uint64(0xFFFFFFFF), i >> np.uint64(32),
                    np.full(n, int(draw) & 0xFFFFFFFF, dtype=np.uint64), np.full(n, int(draw) >> 32, dtype=np.uint64)], axis=-1)
    key = np.array([int(seed) & 0xFFFFFFFF, int(seed) >> 32], dtype=np.uint64)
    r = philox4x32_10(ctr, np.broadcast_to(key, (n, 2)))
    return ((r[:, 0] >> 5).astype(np.float64) * 67108864.0 + (r[:, 1] >> 6).astype(np.float64)) / 9007199254740992.0


def sample_categorical(p, z):
    """sbayes/preprocessing.py:224-256 with the uniforms `z` (shape p.shape[:-1]) passed in instead of
    drawn from np.random: cumulative sums in p's dtype, divided by the last one, first category whose
    cdf exceeds z (argmax of the bool row: 0 when none does)."""
    cdf = np.cumsum(p, axis=-1)
    cdf /= cdf[..., [-1]]
    return np.argmax(np.asarray(z)[..., None] < cdf, axis=-1)


def gibbs_source_propose(features, na_values, groups_by_component, counts_by_component,
                         concentration_by_component, weights, source, objects, z, temperature=1.0,
                         prior_temperature=1.0, sample_from_prior=False):
    """GibbsSampleSource._propose (sbayes/sampling/operators.py:495-552) for the object indices
    `objects` with the uniforms z [n_sub, F]: returns (new_source, log_q, log_q_back, new_counts).
    p is float32, so are its logs and their sums (the reference returns np.float32 scalars)."""
    objects = np.asarray(objects)
    temperature, prior_temperature = float(temperature), float(prior_temperature)    # Python floats: weak promotion
    n_comp = len(groups_by_component)
    hc = has_components(groups_by_component)
    w = normalize_weights(weights, hc)

    def posterior(counts):
        if sample_from_prior:                                   # operators.py:520-522
            return normalize(w[objects] ** (1 / prior_temperature), axis=-1)
        lh = likelihood_per_component(features, na_values, groups_by_component, counts, concentration_by_component)
        return source_posterior(lh, w, objects, temperature, prior_temperature)

    p = posterior(counts_by_component)
    x = np.eye(n_comp, dtype=bool)[sample_categorical(p, z)]
    x[na_values[objects]] = False                               # operators.py:527
    new_source = source.copy()
    new_source[objects] = x
    mask = np.zeros(features.shape[0], dtype=bool)
    mask[objects] = True
    new_counts, _ = update_feature_counts(counts_by_component, features, groups_by_component, groups_by_component,
                                          source, new_source, mask)
    with np.errstate(divide="ignore"):
        log_q = np.log(p[new_source[objects]]).sum()
        p_back = p if sample_from_prior else posterior(new_counts)
        log_q_back = np.log(p_back[source[objects]]).sum()
    return new_source, log_q, log_q_back, new_counts


# --------------------------------------------------------------------------------------
# SourcePrior.__call__ (sbayes/model/prior.py:573-611) and the
# LikelihoodLogger row (sbayes/sampling/loggers.py:354-359)
# --------------------------------------------------------------------------------------
def source_prior_per_object(weights_normalized, source, na_values):
    """float64 [N] holding float32 values: sum over valid features of log(sum_c w * s)."""
    valid = ~na_values
    obs_weights = np.sum(weights_normalized * source, axis=-1)
    with np.errstate(divide="ignore"):
        obs_log = np.log(obs_weights, where=valid, out=np.zeros_like(obs_weights))
    return np.sum(obs_log, where=valid, axis=-1).astype(np.float64)


def logger_row(weights_normalized, lh_exact):
    """float64 [N*F]: what LikelihoodLogger._write_sample appends (before the float32 column cast)."""
    return np.sum(weights_normalized * lh_exact, axis=2).ravel()


# --------------------------------------------------------------------------------------
# ClusterJump.get_jump_lh (sbayes/sampling/operators.py:1679-1722) with
# ClusterEffectProposals.expected_confounder_features (:1342-1379) and posterior_counts (:1254-1259)
# --------------------------------------------------------------------------------------
def weights_heated(weights, has_comp, prior_temperature=1.0):
    """normalize(update_weights(sample) ** (1 / prior_temperature)) (operators.py:1348-1349, 1684-1685): float32."""
    w = normalize_weights(weights, has_comp)
    return normalize(w ** (1 / float(prior_temperature)), axis=-1)


def expected_confounder_features(features, groups_by_component, counts_by_component, concentration_by_component,
                                 unif_cluster, weights, temperature=1.0, prior_temperature=1.0):
    """float32 [N, F, S] (operators.py:1342-1379): per confounder group the tempered effect table
    normalize(unif + (prior - unif) / T_prior + counts / T) weighted by the heated weight of that component and
    ACCUMULATED in float32 in component / group order.  NB the reference passes the CLUSTER prior's uniform
    concentration for every confounder (:1352)."""
    n_objects, n_features, n_states = features.shape
    expected = np.zeros((n_objects, n_features, n_states), dtype=FLOAT_TYPE)
    wh = weights_heated(weights, has_components(groups_by_component), prior_temperature)
    for c in range(1, len(groups_by_component)):
        prior = concentration_by_component[c]
        post = unif_cluster + (prior - unif_cluster) / float(prior_temperature) + counts_by_component[c] / float(temperature)
        p_conf = normalize(post, axis=-1)
        for i_g, g in enumerate(groups_by_component[c]):
            # weights_heated[g, :, [i_comp], None] (advanced indices g and [i_comp] broadcast to n_g): [n_g, F, 1]
            expected[g] += wh[g][:, :, c][..., np.newaxis] * p_conf[np.newaxis, i_g, ...]
    return expected


def jump_lh_per_feature(features, groups_by_component, counts_by_component, concentration_by_component, unif_cluster,
                        weights, i_source, i_target, temperature=1.0, prior_temperature=1.0):
    """(stay, jump): float32 [n_members, F] per-feature likelihoods of the source cluster's members staying /
    jumping to the target cluster (operators.py:1684-1709, before the product over features)."""
    source_cluster = groups_by_component[0][i_source]
    wh = weights_heated(weights, has_components(groups_by_component), prior_temperature)
    w_clust = wh[source_cluster, :, 0]
    prior = concentration_by_component[0]
    p_src = conditional_effect_mean(prior, counts_by_component[0][[i_source]], unif_counts=unif_cluster,
                                    prior_temperature=float(prior_temperature), temperature=float(temperature))
    p_tgt = conditional_effect_mean(prior, counts_by_component[0][[i_target]], unif_counts=unif_cluster,
                                    prior_temperature=float(prior_temperature), temperature=float(temperature))
    p_conf = expected_confounder_features(features, groups_by_component, counts_by_component, concentration_by_component,
                                          unif_cluster, weights, temperature, prior_temperature)[source_cluster]
    p_total_source = p_conf + w_clust[..., np.newaxis] * p_src
    p_total_target = p_conf + w_clust[..., np.newaxis] * p_tgt
    feats = features[source_cluster]
    return np.sum(feats * p_total_source, axis=-1), np.sum(feats * p_total_target, axis=-1)


def jump_lh(features, na_values, groups_by_component, counts_by_component, concentration_by_component, unif_cluster,
            weights, i_source, i_target, temperature=1.0, prior_temperature=1.0):
    """ClusterJump.get_jump_lh (operators.py:1679-1722): float32 [n_members] = lh_jump / (lh_jump + lh_stay), the
    products over features in float32 exactly like the reference (np.prod underflows to 0 beyond F ~ 75)."""
    stay_pf, jump_pf = jump_lh_per_feature(features, groups_by_component, counts_by_component, concentration_by_component,
                                           unif_cluster, weights, i_source, i_target, temperature, prior_temperature)
    valid = ~na_values[groups_by_component[0][i_source]]
    with np.errstate(under="ignore"):
        lh_stay = np.prod(stay_pf, axis=-1, where=valid)
        lh_jump = np.prod(jump_pf, axis=-1, where=valid)
        lh_stay **= (1 / float(temperature))
        lh_jump **= (1 / float(temperature))
    eps = np.finfo(np.float32).eps                     # sbayes/util.py:34
    lh_stay += eps
    lh_jump += eps
    return lh_jump / (lh_jump + lh_stay)


def jump_log_lh(features, na_values, groups_by_component, counts_by_component, concentration_by_component, unif_cluster,
                weights, i_source, i_target, temperature=1.0, prior_temperature=1.0):
    """float64 [2, n_members]: sums of logs of the float32 per-feature values (what the device form returns)."""
    stay_pf, jump_pf = jump_lh_per_feature(features, groups_by_component, counts_by_component, concentration_by_component,
                                           unif_cluster, weights, i_source, i_target, temperature, prior_temperature)
    valid = ~na_values[groups_by_component[0][i_source]]
    with np.errstate(divide="ignore"):
        return np.stack([np.where(valid, np.log(stay_pf.astype(np.float64)), 0.0).sum(axis=-1),
                         np.where(valid, np.log(jump_pf.astype(np.float64)), 0.0).sum(axis=-1)])


def jump_ratio_from_logs(log_stay_jump, temperature=1.0):
    """The host tail of get_jump_lh (operators.py:1706-1722) from the two sums of logs, in the reference's float32:
    exp of a sum of logs in place of np.prod (equal to float32 rounding; both underflow to 0 together)."""
    with np.errstate(under="ignore"):
        lh_stay = np.exp(log_stay_jump[0]).astype(np.float32)
        lh_jump = np.exp(log_stay_jump[1]).astype(np.float32)
        lh_stay **= (1 / float(temperature))
        lh_jump **= (1 / float(temperature))
    eps = np.finfo(np.float32).eps
    lh_stay += eps
    lh_jump += eps
    return lh_jump / (lh_jump + lh_stay)


# --------------------------------------------------------------------------------------
# GibbsSampleWeights.source_lh_by_feature (sbayes/sampling/operators.py:677-685)
# --------------------------------------------------------------------------------------
def source_lh_by_feature(source, weights_normalized, na_values):
    """float32 [F]: sum over objects of log(sum_c source * w), NA observations count 1."""
    p = np.sum(source * weights_normalized, axis=-1)
    p[na_values] = 1
    with np.errstate(divide="ignore"):
        return np.sum(np.log(p), axis=0)
