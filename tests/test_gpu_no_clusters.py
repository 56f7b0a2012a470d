"""n_clusters == 0: the confounders-only baseline the reference supports (its initializer returns an empty cluster
matrix, sbayes/sampling/initializers.py:357; the `clusters` config value is an unconstrained int).  Component 0 then
has no group: empty tables, every object in "no group", has_components[:, 0] all False.  ADVICE r1 (medium)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd import model as sbm
from sbayes_amd.conditionals import likelihood_per_component, mixture_log_likelihood
from sbayes_amd.counts import recalculate_feature_counts
from sbayes_amd.engine import MIXTURE_ONEHOT, MIXTURE_PACKED, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, Engine
from sbayes_amd.likelihood import update_weights
from sbayes_amd.registry import release_all
from sbayes_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu


def _workload_without_clusters(name):
    wl = make_workload(name)
    N, F, S = wl.shape
    groups = [np.zeros((0, N), dtype=bool)] + wl.groups[1:]
    hc = orc.has_components(groups)
    assert not hc[:, 0].any() and hc[:, 1].all()
    rng = np.random.default_rng(5)
    C = len(groups)
    idx = np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)
    source = np.eye(C, dtype=bool)[idx]
    source[wl.na_values] = False
    return wl, groups, source


def _oracle(wl, groups, source):
    counts = orc.recalculate_feature_counts(wl.features, groups, source)
    lh = orc.likelihood_per_component(wl.features, wl.na_values, groups, counts, wl.concentration,
                                      out=np.zeros(wl.features.shape[:2] + (len(groups),)))   # column 0 is never written
    w = orc.normalize_weights(wl.weights, orc.has_components(groups))
    mix = np.log(orc.mixture_observation_lh(w, lh))[~wl.na_values].sum()
    collapsed = orc.collapsed_loglik(counts, wl.concentration)
    return counts, lh, w, mix, collapsed


@pytest.mark.parametrize("name", ["cfg1", "stress"])
def test_engine_with_zero_clusters(name):
    wl, groups, source = _workload_without_clusters(name)
    counts, lh, w, mix, collapsed = _oracle(wl, groups, source)
    n_groups = [g.shape[0] for g in groups]
    assert n_groups[0] == 0
    with Engine(wl.features, n_groups, n_slots=2) as eng:
        for c in range(len(groups)):
            eng.set_concentration(c, wl.concentration[c])
        eng.load_state(0, groups, wl.weights, source=source)
        for c in range(len(groups)):
            eng.update_probs(0, c)
            assert np.array_equal(eng.get_counts(0, c), counts[c])
        assert eng.get_counts(0, 0).shape == (0,) + wl.features.shape[1:]
        assert np.array_equal(eng.weights_normalized(0), w)
        got_lh = eng.likelihood_per_component(0)
        assert np.array_equal(got_lh[..., 1:], lh[..., 1:])
        assert np.array_equal(got_lh[..., 0][~wl.na_values], np.zeros(np.count_nonzero(~wl.na_values)))
        for kernel in (MIXTURE_PACKED, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_ONEHOT):
            eng.set_option(kernel=kernel)
            got = eng.mixture_loglik(0)
            assert abs(got - mix) <= 1e-10 * abs(mix), (kernel, got, mix)
        eng.set_option(kernel=MIXTURE_PACKED)
        got_collapsed = sum(eng.collapsed_loglik(0, c).sum() for c in range(len(groups)))
        assert abs(got_collapsed - collapsed) <= 1e-6 * abs(collapsed)
        # one-call step without a cluster component to move: source rows only
        objs = np.arange(0, wl.shape[0], 7, dtype=np.int32)[:20]
        new_source = source.copy()
        new_source[objs] = np.roll(source[objs], 1, axis=-1) & orc.has_components(groups)[objs][:, None, :]
        fix = ~new_source[objs].any(-1) & ~wl.na_values[objs]
        rows = new_source[objs]
        rows[fix, 1] = True
        new_source[objs] = rows
        want = _oracle(wl, groups, new_source)
        for form in (0, 1):
            eng.set_option(step_form=form)
            glh, mix2, _ = eng.step(0, 1, changed_objects=objs, source_rows=rows)
            assert abs(mix2 - want[3]) <= 1e-10 * abs(want[3]), form
            assert abs(glh.sum() - want[4]) <= 1e-6 * abs(want[4]), form
            for c in range(len(groups)):
                assert np.array_equal(eng.get_counts(1, c), want[0][c])


def test_drop_in_likelihood_with_zero_clusters():
    wl, groups, source = _workload_without_clusters("cfg1")
    counts, lh, w, mix, collapsed = _oracle(wl, groups, source)
    try:
        model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, groups, wl.concentration,
                                  wl.weights, source)
        recalculate_feature_counts(model.data.features.values, sample)
        got = model.likelihood(sample, caching=False)
        assert abs(got - collapsed) <= 1e-6 * abs(collapsed)
        assert model.likelihood(sample, caching=True) == got
        got_lh = likelihood_per_component(model, sample, caching=False)
        assert np.array_equal(got_lh[..., 1:], lh[..., 1:])
        assert np.array_equal(update_weights(sample, caching=False), w)
        fused = mixture_log_likelihood(model, sample)
        assert abs(fused - mix) <= 1e-10 * abs(mix)
    finally:
        release_all()
