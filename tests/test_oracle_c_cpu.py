"""The plain-C restatement of one mixture log-likelihood eval (oracle/sbayes_oracle_c.c: bench.py's compiled CPU baseline and
a second, independently written checker) against the NumPy oracle and the reference's recorded values (tests/golden/*.npz)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from oracle import sbayes_oracle_c as orc_c
from sbayes_amd.synthetic import make_workload
from tests._fixtures import load_npz

NPZ = ["cfg1", "south_america", "test_files"]


@pytest.mark.parametrize("name", NPZ)
def test_c_oracle_reproduces_the_reference_recorded_mixture_ll(name):
    fx = load_npz(name)
    counts = orc.recalculate_feature_counts(fx.features, fx.groups, fx.source)
    got = orc_c.mixture_loglik(fx.features, fx.na_values, fx.groups, counts, fx.conc, fx.weights)
    want = fx.meta["mixture_ll"]                                  # recorded from the reference itself (make_golden.py)
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)
    assert abs(got - orc.mixture_loglik(fx.features, fx.na_values, fx.groups, counts, fx.conc, fx.weights)) <= 1e-12 * abs(want)


@pytest.mark.parametrize("name", ["cfg1", "headline"])
def test_c_oracle_equals_the_numpy_oracle_on_the_synthetic_workloads(name):
    wl = make_workload(name)
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    want = orc.mixture_loglik(wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
    got = orc_c.mixture_loglik(wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
    assert abs(got - want) <= 1e-12 * abs(want), (got, want)


def test_c_oracle_lets_the_last_group_win_on_overlap():
    """an object in several groups of a component takes the LAST group's table in an uncached evaluation."""
    import json
    from tests._fixtures import GOLDEN
    z = np.load(GOLDEN / "overlap.npz")
    meta = json.loads(str(z["meta"]))
    wl = make_workload("cfg1")
    groups = [wl.groups[0], wl.groups[1], z["groups"]]
    unif = wl.states_per_feature.astype(np.float64)
    conc = [unif.copy(), np.broadcast_to(unif, (1,) + unif.shape).copy(), z["conc_2"]]
    counts = [z[f"sample_counts_{c}"] for c in range(3)]
    got = orc_c.mixture_loglik(wl.features, wl.na_values, groups, counts, conc, z["weights"])
    assert abs(got - meta["mixture_ll"]) <= 1e-12 * abs(meta["mixture_ll"])
