"""Overlapping groups on the device and the argument checks added in round 4.

* stateless calls follow the reference on overlap, bit for bit against its recorded outputs (tests/golden/overlap.npz):
  sbe_component_lh -- the LAST WRITTEN group wins, in changed_groups order, unchanged groups' rows stay
  (likelihood.py:121-130); sbe_effect_counts -- once per group (counts.py:28-30);
* resident state (one group id per object and component; round 6): sbe_set_groups keeps the LAST group of an object in several --
  the group an uncached likelihood evaluation ends up with -- and marks the slot; likelihood evaluations run on it (the fused
  mixture kernel reproduces the reference's value), calls that would derive COUNTS from the ids refuse it with SBE_ERR_DATA
  naming object, groups and component; the cluster matrices of sbe_step / sbe_step_batch stay strict (the batch names the chain);
* the drop-in functions' count paths fall back to the stateless calls and reproduce the reference's sample-level results;
* a repeated moved object (batch: SBE_ERR_ARG; single: last entry wins like the matrix form),
  sbe_set_counts_rows on a component whose counts are not resident (SBE_ERR_STATE), the origin named by a deferred
  data check."""
import json

import numpy as np
import pytest

from sbayes_amd import model as sbm
from sbayes_amd.conditionals import likelihood_per_component, mixture_log_likelihood
from sbayes_amd.counts import compute_effect_counts, recalculate_feature_counts, update_feature_counts
from sbayes_amd.engine import Engine, EngineError, GroupOverlapError
from sbayes_amd.likelihood import compute_component_likelihood
from sbayes_amd.registry import release_all
from sbayes_amd.synthetic import make_workload
from tests._fixtures import GOLDEN

pytestmark = pytest.mark.gpu
A1_CASES = ["all", "rev", "c20", "c02", "c1", "none"]


@pytest.fixture(autouse=True)
def _fresh_engines():
    yield
    release_all()


def load_overlap():
    z = np.load(GOLDEN / "overlap.npz")
    return z, json.loads(str(z["meta"])), make_workload("cfg1")


def test_stateless_calls_follow_the_reference_on_overlap():
    z, _, wl = load_overlap()
    ov = z["groups"]
    for tag in A1_CASES:
        buf = z["a1_before"].copy()
        compute_component_likelihood(wl.features, z["probs"], ov, z[f"a1_changed_{tag}"], buf[..., 1])
        assert np.array_equal(buf, z[f"a1_after_{tag}"]), tag
    src = wl.source[..., 1]
    mask = np.zeros(wl.shape[0], dtype=bool)
    mask[z["subset_idx"]] = True
    assert np.array_equal(compute_effect_counts(wl.features, ov, src), z["counts_full"])
    assert np.array_equal(compute_effect_counts(wl.features, ov, src, z["subset_idx"]), z["counts_subset_idx"])
    assert np.array_equal(compute_effect_counts(wl.features, ov, src, mask), z["counts_subset_mask"])


def _sample_level(z, wl):
    groups = [wl.groups[0], wl.groups[1], z["groups"]]
    unif = wl.states_per_feature.astype(np.float64)
    conc = [unif.copy(), np.broadcast_to(unif, (1,) + unif.shape).copy(), z["conc_2"]]
    return groups, conc


def test_resident_state_rejects_overlap_and_the_drop_in_layer_falls_back():
    z, meta, wl = load_overlap()
    groups, conc = _sample_level(z, wl)
    model, sample = sbm.build(wl.features, wl.states_per_feature, meta["component_names"], groups, conc, z["weights"],
                              z["source"])
    feats = model.data.features.values
    eng = model.likelihood.engine
    # (round 6) sbe_set_groups takes the overlapping matrix: the LAST group containing an object is its resident id ...
    eng.set_groups(0, 2, groups[2])
    want_ids = np.full(wl.shape[0], -1, dtype=np.int32)
    for g in range(groups[2].shape[0]):
        want_ids[groups[2][g]] = g
    assert np.array_equal(eng.get_group_ids(0, 2), want_ids) and groups[2].sum(axis=0).max() > 1
    # ... and the calls that would derive COUNTS from one id per object refuse the marked slot, naming object, groups, component
    eng.set_groups(0, 0, groups[0]); eng.set_groups(0, 1, groups[1]); eng.set_source(0, z["source"]); eng.set_weights(0, z["weights"])
    with pytest.raises(GroupOverlapError, match=r"object 4 is in groups 0 and 1 of component 2.*counts\.py:28-30.*sbe_recount") as info:
        eng.recount(0)
    assert info.value.code == 4                                            # SBE_ERR_DATA
    eng.copy_slot(1, 0)                                                    # (the mark travels with the slot)
    with pytest.raises(GroupOverlapError, match=r"one-call step"):
        eng.step(1, 2)
    eng.set_group_ids(1, 2, want_ids)                                      # ids as such cannot overlap: the mark is gone
    eng.recount(1)
    recalculate_feature_counts(feats, sample)
    for c, k in enumerate(meta["component_names"]):
        assert np.array_equal(sample.feature_counts[k].value, z[f"sample_counts_{c}"])
    ll = model.likelihood(sample, caching=False)
    assert abs(ll - meta["collapsed_ll"]) <= 1e-6 * abs(meta["collapsed_ll"])
    np.testing.assert_allclose(sample.cache.group_likelihoods["overlapping"].value, z["group_lh_2"], rtol=1e-6)
    assert np.array_equal(likelihood_per_component(model, sample, caching=False), z["lh_per_component"])
    new = sample.copy()
    with new.source.edit() as src:
        src[z["subset_idx"]] = z["delta_source_new"][z["subset_idx"]]
    update_feature_counts(sample, new, feats, z["subset_idx"])
    for c, k in enumerate(meta["component_names"]):
        assert np.array_equal(new.feature_counts[k].value, z[f"delta_counts_{c}"])
    # the fused evaluation on RESIDENT state = the reference's uncached evaluation of the overlapping sample (1e-10), and the
    # collapsed likelihood now comes from the resident counts the bind sent (no stateless fall-back left in Likelihood.__call__)
    mix = mixture_log_likelihood(model, sample)
    assert abs(mix - meta["mixture_ll"]) <= 1e-10 * abs(meta["mixture_ll"]), (mix, meta["mixture_ll"])
    from sbayes_amd.conditionals import observation_likelihoods
    from oracle import sbayes_oracle as orc
    lh = orc.likelihood_per_component(wl.features, wl.na_values, groups, [z[f"sample_counts_{c}"] for c in range(3)], conc)
    w = orc.normalize_weights(z["weights"], orc.has_components(groups))
    assert np.array_equal(observation_likelihoods(model, sample), orc.mixture_observation_lh(w, lh))


def _resident_engine(wl, n_slots):
    eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=n_slots)
    for c in range(wl.n_components):
        eng.set_concentration(c, wl.concentration[c])
    for slot in range(0, n_slots, 2):
        eng.load_state(slot, wl.groups, wl.weights, source=wl.source)
        for c in range(wl.n_components):
            eng.update_probs(slot, c)
        eng.mixture_loglik(slot)
    return eng


def test_steps_reject_an_overlapping_cluster_matrix():
    wl = make_workload("cfg1")
    with _resident_engine(wl, 4) as eng:
        bad = wl.clusters.copy()
        n = int(np.flatnonzero(bad[0])[0])
        bad[1, n] = True                                                   # object n in clusters 0 and 1
        with pytest.raises(GroupOverlapError, match=rf"object {n} is in groups 0 and 1 of component 0"):
            eng.step(0, 1, clusters=bad)
        eng.set_option(step_form=1)                                        # the call-by-call form checks the same way
        with pytest.raises(GroupOverlapError, match=rf"object {n} is in groups 0 and 1 of component 0"):
            eng.step(0, 1, clusters=bad)
        eng.set_option(step_form=0)
        good = wl.clusters.copy()
        glh, mix, _ = eng.step(0, 1, clusters=good)                        # the engine is usable afterwards
        assert np.isfinite(mix) and np.isfinite(glh).all()
        stacked = np.stack([good, bad])
        with pytest.raises(EngineError, match=rf"chain 1.*object {n} is in groups 0 and 1"):
            eng.step_batch([0, 2], [1, 3], clusters=stacked)


def test_repeated_moved_objects():
    wl = make_workload("cfg1")
    K = wl.clusters.shape[0]
    with _resident_engine(wl, 4) as eng:
        ids = np.where(wl.clusters.any(axis=0), wl.clusters.argmax(axis=0), -1)
        n = int(np.flatnonzero(ids == -1)[0])
        # batch: refused, naming chain and object
        with pytest.raises(EngineError, match=rf"chain 0: object {n} listed twice in moved_objects") as info:
            eng.step_batch_delta([0], [1], [0, 2], [n, n], [0, -1])
        assert info.value.code == 1                                        # SBE_ERR_ARG
        # single: the last entry wins, as in the matrix form (none -> 0 -> 1 == none -> 1)
        eng.step(0, 1)                                                     # records of the slot pair: the patching form is live
        glh_d, mix_d, ch_d = eng.step_delta(0, 1, [n, n], [0, K - 1])
        want = wl.clusters.copy()
        want[:, n] = False
        want[K - 1, n] = True
        glh_m, mix_m, ch_m = eng.step(2, 3, clusters=want)
        assert np.array_equal(glh_d, glh_m) and np.array_equal(ch_d, ch_m)
        assert abs(mix_d - mix_m) <= 1e-12 * abs(mix_m)
        for c in range(wl.n_components):
            assert np.array_equal(eng.get_counts(1, c), eng.get_counts(3, c))


def test_count_rows_need_resident_counts_and_deferred_checks_name_their_origin():
    wl = make_workload("cfg1")
    F, S = wl.shape[1:]
    with Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=1) as eng:
        with pytest.raises(EngineError, match=r"counts of component 1 not set .*sbe_set_counts first") as info:
            eng.set_counts_rows(0, [wl.groups[0].shape[0]], np.zeros((1, F, S), dtype=np.float32))
        assert info.value.code == 3                                        # SBE_ERR_STATE
        for c in range(wl.n_components):
            eng.set_concentration(c, wl.concentration[c])
        eng.load_state(0, wl.groups, wl.weights, source=wl.source)
        eng.set_counts_rows(0, [wl.groups[0].shape[0]], eng.get_counts(0, 1)[:1])
        # a deferred data check surfaces in a later call: its message says where the data came in
        eng.set_option(deferred_checks=True)
        conc = wl.concentration[0].copy()
        conc[0] = 0.0                                                      # feature 0 of every cluster: counts + prior can sum to 0
        eng.set_concentration(0, conc)
        eng.set_counts(0, 0, np.zeros_like(eng.get_counts(0, 0)))
        eng.update_probs(0, 1)
        eng.update_probs(0, 0)                                             # returns at once: nothing is checked yet
        with pytest.raises(EngineError, match=r"normalize: \d+ rows have a non-positive sum.*deferred data check: raised by "
                                              r"sbe_update_probs"):
            eng.mixture_loglik(0)
        eng.set_option(deferred_checks=False)
        with pytest.raises(EngineError, match=r"normalize: \d+ rows have a non-positive sum \(sbayes/util.py:1006 assert\)$"):
            eng.update_probs(0, 0)                                         # immediate mode: at the call, no note
