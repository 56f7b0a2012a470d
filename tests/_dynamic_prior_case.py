"""The `any_dynamic_priors = True` branches of the path driven end to end -- TEST INFRASTRUCTURE shared by
tests/test_dynamic_priors_cpu.py (oracle-backed double) and tests/test_gpu_dynamic_priors.py (the device).

tests/golden/dynamic_prior.npz holds the outputs of the REFERENCE's own ConfoundingEffectsPrior with the `universal`
group prior type (make_golden.py dynamic_prior_fixture): concentration tables that follow the universal counts
(prior.py:325-354), the cached Likelihood.__call__ / likelihood_per_component across a hyperprior change
(likelihood.py:92-93, conditionals.py:197-200) and component_likelihood_given_unchanged with
concentration_array_given_unchanged (operators.py:905-915, prior.py:356-387)."""
import json
from collections import OrderedDict

import numpy as np

from sbayes_amd import model as sbm
from sbayes_amd.state import Confounder, Features, Sample
from sbayes_amd.synthetic import make_workload
from tests._fixtures import GOLDEN


def load():
    z = np.load(GOLDEN / "dynamic_prior.npz")
    meta = json.loads(str(z["meta"]))
    sh = meta["shape"]
    wl = make_workload("dynamic", shape=(sh[0], sh[1], sh[2], sh[3], tuple(sh[4]), sh[5]))
    return z, meta, wl


def build(z, meta, wl):
    """(model, sample) on the mirror types with the dynamic prior on the third component."""
    names = meta["component_names"]
    feats = Features(wl.features, states=wl.states_per_feature)
    confounders = OrderedDict()
    for name, g in zip(names[1:], wl.groups[1:]):
        confounders[name] = Confounder(name, g, has_universal_prior=(name != "universal"))
    unif = wl.states_per_feature.astype(np.float64)
    universal_prior = sbm.ConfoundingEffectsPrior(np.broadcast_to(unif, (1,) + unif.shape).copy())
    dynamic = sbm.UniversalConfoundingEffectsPrior(wl.groups[2].shape[0], wl.states_per_feature, meta["precision"],
                                                   universal_prior, wl.features)
    prior = sbm.Prior(unif.copy(), {"universal": universal_prior, names[2]: dynamic})
    model = sbm.Model(sbm.Data(feats, confounders), n_clusters=wl.clusters.shape[0], prior=prior)
    n, f, s = wl.shape
    counts0 = {k: np.zeros((g.shape[0], f, s), dtype=np.float32) for k, g in zip(names, wl.groups)}
    sample = Sample.from_numpy_arrays(clusters=z["clusters"], weights=z["weights"], confounders=confounders,
                                      source=z["source"], feature_counts=counts0, model_shapes=model.shapes)
    return model, sample, dynamic


def drive(z, meta, wl, exact_tables=True):
    """The fixture's scenario on the drop-in layer; every comparison is against the reference's recorded outputs."""
    from sbayes_amd.conditionals import likelihood_per_component
    from sbayes_amd.counts import recalculate_feature_counts, update_feature_counts
    from sbayes_amd.operators import component_likelihood_given_unchanged
    names = meta["component_names"]
    model, sample, dynamic = build(z, meta, wl)
    feats = model.data.features.values
    assert dynamic.any_dynamic_priors and sample.confounders[names[2]].has_universal_prior
    recalculate_feature_counts(feats, sample)

    def check(tag, smp):
        assert np.array_equal(dynamic.concentration_array(smp), z[f"{tag}_conc_2"])        # the mirror prior = the reference's
        ll = model.likelihood(smp, caching=True)
        assert abs(ll - meta[f"{tag}_collapsed_ll"]) <= 2e-6 * abs(meta[f"{tag}_collapsed_ll"]), (tag, ll)
        for i, k in enumerate(names):
            assert np.array_equal(smp.feature_counts[k].value, z[f"{tag}_counts_{i}"])
            np.testing.assert_allclose(smp.cache.group_likelihoods[k].value, z[f"{tag}_group_lh_{i}"], rtol=2e-6, atol=1e-6)
        lh = likelihood_per_component(model, smp, caching=True)
        assert np.array_equal(lh, z[f"{tag}_lh"]), tag

    check("s0", sample)
    new = sample.copy()
    subset = z["s1_subset"]
    with new.source.edit() as src:
        src[subset] = z["s1_source"][subset]
    update_feature_counts(sample, new, feats, subset)
    # conf1's counts are unchanged, its concentration is not: the cached group values of `sample` must NOT be reused
    assert np.array_equal(new.feature_counts[names[2]].value, sample.feature_counts[names[2]].value)
    cache = new.cache.group_likelihoods[names[2]]
    assert cache.ahead_of("universal_counts") and len(cache.what_changed("counts", caching=True)) == 0
    check("s1", new)
    assert not np.array_equal(z["s1_group_lh_2"], z["s0_group_lh_2"])
    # back on the OLD sample: its tables are the old concentration's again (the prior rewrites its array in place:
    # the bind cache compares content, binding._same)
    assert np.array_equal(likelihood_per_component(model, sample, caching=False), z["s0_lh"])
    mask = np.zeros(wl.shape[0], dtype=bool)
    mask[subset] = True
    for tag, t, tp in (("plain", 1.0, 1.0), ("tempered", 2.5, 1.7)):
        assert np.array_equal(dynamic.concentration_array_given_unchanged(new, mask), z[f"given_unchanged_conc_{tag}"])
        got = component_likelihood_given_unchanged(model, new, mask, 0, temperature=t, prior_temperature=tp)
        want = z[f"given_unchanged_{tag}"]
        assert got.shape == want.shape and got.dtype == want.dtype
        if t == 1.0 and exact_tables:
            assert np.array_equal(got, want), tag
        else:
            np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7, err_msg=tag)
    return model, sample, new
