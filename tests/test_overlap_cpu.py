"""Overlapping groups, T0: the oracle against the reference's recorded outputs
(tests/golden/overlap.npz, written by make_golden.py overlap_fixture), and the drop-in host layer's fall-back from the
count-deriving resident forms (one group id per object and component: the last group, marked) to the stateless device calls, on the
oracle-backed engine double.

Reference semantics pinned here: a1 lets the LAST WRITTEN group win, in the order `changed_groups` lists them, and
leaves rows of unchanged groups alone (likelihood.py:121-130); a9 counts an object once per group it is in
(counts.py:28-30)."""
import json

import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd.synthetic import make_workload
from tests._fixtures import GOLDEN

A1_CASES = ["all", "rev", "c20", "c02", "c1", "none"]


def load_overlap():
    z = np.load(GOLDEN / "overlap.npz")
    meta = json.loads(str(z["meta"]))
    wl = make_workload("cfg1")
    return z, meta, wl


def test_oracle_a1_with_overlap_matches_the_reference():
    z, _, wl = load_overlap()
    ov = z["groups"]
    assert (ov.sum(axis=0) > 1).sum() >= 10 and ov[:, 4].all() and not ov[:, :4].any()
    for tag in A1_CASES:
        buf = z["a1_before"].copy()
        orc.compute_component_likelihood(wl.features, z["probs"], ov, z[f"a1_changed_{tag}"], buf[..., 1])
        assert np.array_equal(buf, z[f"a1_after_{tag}"]), tag
    # the order of changed_groups decides (object 5 is in groups 0 and 2): the two orders differ exactly there
    a, b = z["a1_after_c20"][..., 1], z["a1_after_c02"][..., 1]
    both = ov[0] & ov[2]
    assert both[5] and not np.array_equal(a[both], b[both]) and np.array_equal(a[~both], b[~both])
    # a strict subset leaves an object of (changed 1, unchanged 3) with group 1's row; objects only in 3 stay stale
    c1 = z["a1_after_c1"][..., 1]
    only3 = ov[3] & ~ov[1]
    assert np.array_equal(c1[only3], z["a1_before"][..., 1][only3])


def test_oracle_counts_with_overlap_match_the_reference():
    z, _, wl = load_overlap()
    ov, src = z["groups"], wl.source[..., 1]
    mask = np.zeros(wl.shape[0], dtype=bool)
    mask[z["subset_idx"]] = True
    assert np.array_equal(orc.compute_effect_counts(wl.features, ov, src), z["counts_full"])
    assert np.array_equal(orc.compute_effect_counts(wl.features, ov, src, z["subset_idx"]), z["counts_subset_idx"])
    assert np.array_equal(orc.compute_effect_counts(wl.features, ov, src, mask), z["counts_subset_mask"])
    # once per group: the total exceeds the number of counted observations
    n_obs = (src & wl.features.any(axis=-1)).sum()
    assert z["counts_full"].sum() > n_obs * 0 and z["counts_full"].sum() == (ov.sum(axis=0)[:, None] * (src & wl.features.any(axis=-1))).sum()


def _sample_level(z, wl):
    groups = [wl.groups[0], wl.groups[1], z["groups"]]
    unif = wl.states_per_feature.astype(np.float64)
    conc = [unif.copy(), np.broadcast_to(unif, (1,) + unif.shape).copy(), z["conc_2"]]
    return groups, conc


def test_oracle_sample_level_with_overlap():
    z, meta, wl = load_overlap()
    groups, conc = _sample_level(z, wl)
    counts = orc.recalculate_feature_counts(wl.features, groups, z["source"])
    for c in range(3):
        assert np.array_equal(counts[c], z[f"sample_counts_{c}"])
    assert np.array_equal(orc.collapsed_group_logliks(counts[2], conc[2]), z["group_lh_2"])
    assert abs(orc.collapsed_loglik(counts, conc) - meta["collapsed_ll"]) <= 1e-6 * abs(meta["collapsed_ll"])
    lh = orc.likelihood_per_component(wl.features, wl.na_values, groups, counts, conc)
    assert np.array_equal(lh, z["lh_per_component"])
    assert orc.mixture_loglik(wl.features, wl.na_values, groups, counts, conc, z["weights"]) == meta["mixture_ll"]
    new_counts, _ = orc.update_feature_counts(counts, wl.features, groups, groups, z["source"], z["delta_source_new"],
                                              z["subset_idx"])
    for c in range(3):
        assert np.array_equal(new_counts[c], z[f"delta_counts_{c}"])


def test_drop_in_layer_falls_back_to_the_stateless_calls(monkeypatch):
    """recalculate_feature_counts / update_feature_counts / Likelihood.__call__ on a sample whose third component has
    overlapping groups: the resident bind is refused (GroupOverlapError), the stateless device calls serve the
    reference's semantics; likelihood_per_component never binds (a1 is stateless)."""
    from sbayes_amd import conditionals, counts as my_counts, likelihood, model as sbm, registry
    from sbayes_amd.engine import GroupOverlapError
    from tests._fake_engine import make_get_engine
    z, meta, wl = load_overlap()
    groups, conc = _sample_level(z, wl)
    engines = {}
    get_engine = make_get_engine(engines)
    for mod in (registry, likelihood, conditionals, my_counts):
        monkeypatch.setattr(mod, "get_engine", get_engine, raising=True)
    from sbayes_amd import binding
    monkeypatch.setattr(binding, "get_engine", get_engine, raising=True)
    model, sample = sbm.build(wl.features, wl.states_per_feature, meta["component_names"], groups, conc, z["weights"],
                              z["source"])
    feats = model.data.features.values
    my_counts.recalculate_feature_counts(feats, sample)
    for c, k in enumerate(meta["component_names"]):
        assert np.array_equal(sample.feature_counts[k].value, z[f"sample_counts_{c}"])
    eng = next(iter(engines.values()))
    # (round 6) the resident bind takes the overlapping matrix -- the last group is the object's id -- and only the calls that
    # would derive counts from the ids refuse the slot
    eng.set_groups(0, 2, groups[2])
    with pytest.raises(GroupOverlapError, match=r"object \d+ is in groups \d+ and \d+ of component 2"):
        eng.recount(0)
    ll = model.likelihood(sample, caching=False)
    assert abs(ll - meta["collapsed_ll"]) <= 1e-6 * abs(meta["collapsed_ll"])
    np.testing.assert_allclose(sample.cache.group_likelihoods["overlapping"].value, z["group_lh_2"], rtol=1e-6)
    lh = conditionals.likelihood_per_component(model, sample, caching=False)
    assert np.array_equal(lh, z["lh_per_component"])
    new = sample.copy()
    with new.source.edit() as src:
        src[z["subset_idx"]] = z["delta_source_new"][z["subset_idx"]]
    my_counts.update_feature_counts(sample, new, feats, z["subset_idx"])
    for c, k in enumerate(meta["component_names"]):
        assert np.array_equal(new.feature_counts[k].value, z[f"delta_counts_{c}"])
    # the fused evaluation on RESIDENT state follows the reference's uncached evaluation (the last group's table: likelihood.py:126-130)
    mix = conditionals.mixture_log_likelihood(model, sample)
    assert abs(mix - meta["mixture_ll"]) <= 1e-10 * abs(meta["mixture_ll"])
