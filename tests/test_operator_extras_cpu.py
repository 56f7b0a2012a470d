"""T0 for the next-heaviest operator expressions (missing #2 / #3): the oracle's restatements of
ClusterJump.get_jump_lh (+ expected_confounder_features) and GibbsSampleWeights.source_lh_by_feature against what the
reference's own operators returned on the fixtures (tests/golden/make_golden.py:operator_extras_case)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from tests._fixtures import crc, load_npz, load_synthetic_trace

CASES = ["south_america", "cfg1", "headline"]


def load_case(name):
    if name in ("cfg1", "headline"):
        fx, tr = load_synthetic_trace(name)
        return fx, tr.z
    fx = load_npz(name)
    return fx, fx.z


def jump_keys(z):
    return sorted(k for k in z.files if k.startswith("jp_") and k.count("_") == 3 and k != "jp_cluster_unif")


@pytest.mark.parametrize("name", CASES)
def test_oracle_jump_lh_matches_the_reference_operator(name):
    fx, z = load_case(name)
    keys = jump_keys(z)
    assert keys
    unif = z["jp_cluster_unif"]
    for key in keys:
        _, tag, s, t = key.split("_")
        temp, ptemp = (1.0, 1.0) if tag == "t1" else (1.3, 1.5)
        i_s, i_t = int(s[1:]), int(t[1:])
        args = (fx.features, fx.groups, fx.counts, fx.conc, unif, fx.weights, i_s, i_t, temp, ptemp)
        stay_pf, jump_pf = orc.jump_lh_per_feature(*args)
        assert np.array_equal(stay_pf[:16], z[key + "_stay_pf"]) and crc(stay_pf) == int(z[key + "_stay_pf_crc"])
        assert np.array_equal(jump_pf[:16], z[key + "_jump_pf"]) and crc(jump_pf) == int(z[key + "_jump_pf_crc"])
        got = orc.jump_lh(fx.features, fx.na_values, *args[1:])
        assert got.dtype == np.float32 and np.array_equal(got, z[key]), key
        # the sum-of-logs route (what the device returns) reproduces the reference's float32 products to rounding
        logs = orc.jump_log_lh(fx.features, fx.na_values, *args[1:])
        np.testing.assert_allclose(orc.jump_ratio_from_logs(logs, temp), z[key], rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("name", CASES + ["test_files"])
def test_oracle_source_lh_by_feature_matches_the_reference(name):
    fx, z = load_case(name)
    w = orc.normalize_weights(fx.weights, orc.has_components(fx.groups))
    got = orc.source_lh_by_feature(fx.source, w, fx.na_values)
    assert got.dtype == np.float32 and np.array_equal(got, z["swl_lh_by_feature"])


@pytest.mark.parametrize("tag", ["grow", "shrink", "mc3", "prior"])
def test_cluster_gibbs_sample_source_on_the_double_matches_the_reference(tag, monkeypatch):
    """operators.cluster_gibbs_sample_source (host logic) over the oracle-backed double's given_unchanged_gibbs == the
    reference's ClusterOperator.gibbs_sample_source (operators.py:796-851) fed with the same uniforms: new source array,
    counts and float32 log_q / log_q_back BIT FOR BIT, tempered cases included (the double restates the expressions in
    the reference's dtypes)."""
    from sbayes_amd import binding, conditionals, counts, likelihood, registry
    from tests import _cluster_gibbs_case as case
    from tests._fake_engine import make_get_engine
    engines = {}
    get_engine = make_get_engine(engines)
    for mod in (registry, likelihood, conditionals, counts, binding):
        monkeypatch.setattr(mod, "get_engine", get_engine, raising=True)
    exact, (lq, want_lq), (lqb, want_lqb), sample, objects, fx = case.run_case(tag)
    assert lq == np.float32(want_lq) and lqb == np.float32(want_lqb), (tag, lq, want_lq, lqb, want_lqb)
    others = np.setdiff1d(np.arange(sample.n_objects), objects)
    assert np.array_equal(sample.source.value, fx.source)                             # the old sample is not modified
    kinds = [c[0] for c in next(iter(engines.values())).calls]
    assert kinds.count("given_unchanged_gibbs") == 1 and "counts_delta" not in kinds     # (the count delta rides on the same call)

