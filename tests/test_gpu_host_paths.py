"""Both branches of the engine's small-input / small-result plumbing (round 3, second pass over the drop-in path):

  stage()            input read in place out of the mapped ring (<= 64 KB)  |  ordinary upload to scratch
  upload_segments()  patterns + tuple tables with one scatter launch        |  one upload per array (> 64 KB together)
  out_target()       result written straight into host-mapped memory        |  device scratch + staging copy
  collapsed_groups() one launch, F*S terms of a group in LDS                |  k_dcl + k_group_sum_f32 (> 96 KB)
  normalize_weights  weights read out of the ring (<= 256 rows)             |  out of device memory
  update_probs       one component                                          |  a set of components (sbe_update_probs_mask)

The suite's fixtures mostly sit on the left-hand side; the shapes here are chosen to take the right-hand one as well.
Everything is compared with the oracle-backed double on the same state (bit-exact unless stated)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload
from tests._fake_engine import FakeEngine

pytestmark = pytest.mark.gpu

SHAPES = {
    "flat": (24, 20, 4, 2, (), False),                       # everything small: the in-place / mapped branches
    "wide_tables": (48, 700, 20, 3, (2,), True),             # F*S*8 = 112 KB per group: two-kernel collapsed form; 56 KB rows
    "long": (30000, 6, 3, 4, (5,), False),                   # 30 000 objects: pattern / tuple arrays > 64 KB, big id lists
}


def _pair(name):
    wl = make_workload(name, shape=SHAPES[name])
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    n_groups = [g.shape[0] for g in wl.groups]
    eng, fake = Engine(wl.features, n_groups, n_slots=2), FakeEngine(wl.features, n_groups)
    for e in (eng, fake):
        for c in range(len(wl.groups)):
            e.set_concentration(c, wl.concentration[c])
            e.set_groups(0, c, wl.groups[c])
            e.set_counts(0, c, counts[c])
        e.set_source(0, wl.source)
        e.set_weights(0, wl.weights)
        e.set_uniform_counts(wl.states_per_feature.astype(np.float64))
    return eng, fake, wl, counts


@pytest.mark.parametrize("name", list(SHAPES))
def test_collapsed_loglik_both_forms(name):
    eng, fake, wl, counts = _pair(name)
    try:
        want_all = fake.collapsed_loglik_all(0)
        np.testing.assert_allclose(eng.collapsed_loglik_all(0), want_all, rtol=2e-6, atol=1e-6)
        for c in range(eng.n_components):
            per_group, per_feature = eng.collapsed_loglik(0, c, per_feature=True)
            want_pf = orc.dirichlet_categorical_logpdf(counts[c], wl.concentration[c])
            np.testing.assert_allclose(per_feature, want_pf, rtol=2e-6, atol=1e-6)
            # the group value is the float32 NumPy-order sum of the engine's own per-feature row: exact
            assert np.array_equal(per_group, np.array([row.sum(dtype=np.float32) for row in per_feature], dtype=np.float64))
            assert np.array_equal(eng.collapsed_loglik(0, c), per_group)
    finally:
        eng.close()


@pytest.mark.parametrize("name", list(SHAPES))
def test_row_uploads_and_component_sets(name):
    """set_counts_rows / set_source_rows (in place or through scratch), get_counts / get_source_rows (mapped or copied),
    update_probs with a set of components: resident state and tables equal the double's, bit for bit."""
    eng, fake, wl, counts = _pair(name)
    try:
        rng = np.random.default_rng(3)
        N, F, C = wl.source.shape
        off = eng.group_offsets
        # new count rows for a few groups of several components, one call
        idx = np.unique(rng.integers(0, eng.n_groups_total, size=min(4, eng.n_groups_total)))
        rows = rng.integers(0, 50, size=(idx.size, F, eng.n_states)).astype(np.float32)
        for e in (eng, fake):
            e.set_counts_rows(0, idx, rows)
        # new source rows for a subset (big enough to leave the in-place branch at the long shape)
        objs = np.sort(rng.choice(N, size=min(N, 12000 if name == "long" else 9), replace=False))
        pick = rng.integers(0, C, size=(objs.size, F))
        new_rows = pick[..., None] == np.arange(C)
        new_rows[~wl.features[objs].any(-1)] = False
        for e in (eng, fake):
            e.set_source_rows(0, objs, new_rows)
        assert np.array_equal(eng.get_source_rows(0, objs), new_rows)
        for c in range(C):
            assert np.array_equal(eng.get_counts(0, c), fake._slot(0)["counts"][c])
        # tables: every component singly on one side, as sets on the other
        sets = [[0], list(range(1, C))] if C > 1 else [[0]]
        for s in sets:
            eng.update_probs(0, s)
        for c in range(C):
            want = orc.component_probs(fake._slot(0)["counts"][c], wl.concentration[c])
            assert np.array_equal(eng.get_probs(0, c), want), c
        eng.update_probs(0, range(C))
        for c in range(C):
            assert np.array_equal(eng.get_probs(0, c), orc.component_probs(fake._slot(0)["counts"][c], wl.concentration[c]))
        with pytest.raises(ValueError):
            eng.update_probs(0, [C])
        # and the fused evaluation sees all of it
        ll = eng.mixture_loglik(0)
        want_ll = orc.mixture_loglik(wl.features, ~wl.features.any(-1), wl.groups, fake._slot(0)["counts"], wl.concentration,
                                     wl.weights)
        assert abs(ll - want_ll) <= 1e-10 * abs(want_ll)
    finally:
        eng.close()


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("n_rows", [1, 7, 256, 300])
def test_normalize_weights_row_form(name, n_rows):
    """normalize_weights(weights, has_components[rows]) (likelihood.py:171-190) in row form, both input branches:
    bit-exact against the oracle's np.unique form."""
    eng, fake, wl, _ = _pair(name)
    try:
        rng = np.random.default_rng(n_rows)
        C = eng.n_components
        hc = rng.random((n_rows, C)) < 0.6
        hc[:, min(1, C - 1)] = True                                # (the universal confounder: never an all-False row)
        got = eng.normalize_weights(wl.weights, hc)
        assert got.dtype == np.float32 and got.shape == (n_rows, eng.n_features, C)
        assert np.array_equal(got, orc.normalize_weights(wl.weights, hc))
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["flat", "long"])
def test_resident_operator_forms_at_both_sizes(name):
    """source_prior / source_posterior / source_lh_by_feature / given_unchanged_lh with small (mapped) and large (copied)
    results and subset lists."""
    eng, fake, wl, _ = _pair(name)
    try:
        rng = np.random.default_rng(5)
        N = wl.source.shape[0]
        for c in range(eng.n_components):
            eng.update_probs(0, c)
        np.testing.assert_allclose(eng.source_prior(0), fake.source_prior(0), rtol=2e-6, atol=1e-6)
        # the reference adds N float32 logs in float32, one after the other (up to N/2 ulp of accumulated rounding: 2e-4
        # relative observed at N = 30 000); the device adds them in float64 and rounds once -- so it agrees with the
        # reference's value within the reference's own rounding bound, and with the exactly summed logs at float32 accuracy
        got = eng.source_lh_by_feature(0)
        np.testing.assert_allclose(got, fake.source_lh_by_feature(0), rtol=max(3e-5, N * 2.0 ** -25), atol=1e-5)
        w = orc.normalize_weights(wl.weights, orc.has_components(wl.groups))
        p = np.where(~wl.features.any(-1), np.float32(1), (w * wl.source).sum(-1, dtype=np.float32))
        with np.errstate(divide="ignore"):
            exact = np.log(p, dtype=np.float32).astype(np.float64).sum(axis=0)
        np.testing.assert_allclose(got, exact, rtol=2e-6, atol=1e-5)
        for n in (1, 40, min(N, 9000)):
            objs = np.sort(rng.choice(N, size=min(n, N), replace=False))
            assert np.array_equal(eng.source_posterior(0, objs), fake.source_posterior(0, objs)), n
            got = eng.given_unchanged_lh(0, 1, objs)
            assert np.array_equal(got, fake.given_unchanged_lh(0, 1, objs)), n
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["flat", "long"])
def test_set_groups_after_small_moves_equals_a_fresh_slot(name):
    """sbe_set_groups after a cluster move: the ids, the pattern / tuple tables (followed on the host in O(moved objects)
    when the set of has_components patterns stays what it is, derived from all N otherwise) and the per-pattern weights go
    up in one launch.  A slot walked through a chain of moves -- single objects, a handful, a move that empties a cluster
    (the pattern set changes), a big reshuffle (full derivation) -- answers like a slot that receives the final groups
    at once: same mixture log-likelihood bits, same source prior, same membership marginals."""
    eng, fake, wl, counts = _pair(name)
    try:
        rng = np.random.default_rng(11)
        clusters = wl.groups[0].copy()
        K, N = clusters.shape
        for c in range(eng.n_components):
            eng.update_probs(0, c)
        # slot 1: the same state, set up once per stop
        def fresh(groups0):
            for c in range(eng.n_components):
                eng.set_groups(1, c, groups0 if c == 0 else wl.groups[c])
                eng.set_counts(1, c, counts[c])
                eng.update_probs(1, c)
            eng.set_source(1, wl.source)
            eng.set_weights(1, wl.weights)
        def move(n_obj, empty_cluster=None):
            if empty_cluster is not None:
                clusters[empty_cluster] = False
                return
            for n in rng.choice(N, size=n_obj, replace=False):
                clusters[:, n] = False
                k = int(rng.integers(0, K + 1))
                if k < K:
                    clusters[k, n] = True
        objs = np.sort(rng.choice(N, size=min(N, 50), replace=False))
        for step, (n_obj, empty) in enumerate([(1, None), (1, None), (3, None), (1, None), (0, 1), (2, None), (N // 2, None), (1, None), (1, None)]):
            move(n_obj, empty)
            eng.set_groups(0, 0, clusters)
            fresh(clusters)
            assert eng.mixture_loglik(0) == eng.mixture_loglik(1), step
            assert np.array_equal(eng.source_prior(0), eng.source_prior(1)), step
            assert np.array_equal(eng.cluster_posterior_marginals(0, 0, objs), eng.cluster_posterior_marginals(1, 0, objs)), step
            assert np.array_equal(eng.weights_normalized(0), eng.weights_normalized(1)), step
        want = orc.normalize_weights(wl.weights, orc.has_components([clusters] + list(wl.groups[1:])))
        assert np.array_equal(eng.weights_normalized(0), want)
    finally:
        eng.close()


@pytest.mark.parametrize("name", list(SHAPES))
def test_collapsed_likelihood_and_source_prior_in_one_call(name):
    """Model.__call__ = likelihood + prior (model.py:47-51): sbe_collapsed_and_source_prior returns what the two calls
    return, bit for bit -- one launch (flat, long) or the two calls behind one entry (wide_tables: a group's terms exceed
    the LDS budget) -- and agrees with the double within the two calls' own tolerances."""
    eng, fake, wl, counts = _pair(name)
    try:
        for c, table in enumerate(eng.get_counts_all(0)):            # (every component's counts in one call)
            assert table.dtype == np.float32 and np.array_equal(table, eng.get_counts(0, c)) and np.array_equal(table, counts[c])
        per_group, per_object = eng.collapsed_and_source_prior(0)
        assert np.array_equal(per_group, eng.collapsed_loglik_all(0))
        assert np.array_equal(per_object, eng.source_prior(0))
        want_g, want_o = fake.collapsed_and_source_prior(0)
        np.testing.assert_allclose(per_group, want_g, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(per_object, want_o, rtol=2e-6, atol=1e-6)
        # a moved object and a changed source row reach both halves
        clusters = wl.groups[0].copy()
        clusters[:, 3] = False
        clusters[0, 3] = True
        eng.set_groups(0, 0, clusters)
        per_group2, per_object2 = eng.collapsed_and_source_prior(0)
        assert np.array_equal(per_group2, per_group) and np.array_equal(per_object2, eng.source_prior(0))
        assert not np.array_equal(per_object2, per_object) or np.array_equal(clusters, wl.groups[0])
    finally:
        eng.close()


@pytest.mark.parametrize("name", list(SHAPES))
def test_set_slot_delta_is_the_three_setters(name):
    """sbe_set_slot_delta: set_groups + set_counts_rows(update_probs) + set_source_rows of one bind in ONE launch (flat: all
    three through the mapped ring; wide_tables: count rows beyond the ring's direct size; long: the pattern / tuple arrays
    beyond it) -- the slot ends exactly as after the three calls: ids and tables (mixture log-likelihood, normalised weights),
    counts and probability rows, source rows (source prior); every subset of the three as well."""
    eng, fake, wl, counts = _pair(name)
    try:
        rng = np.random.default_rng(23)
        C = eng.n_components
        N = wl.source.shape[0]
        off = eng.group_offsets
        for c in range(C):
            eng.update_probs(0, c)
        clusters = wl.groups[0].copy()
        moved = rng.choice(N, size=min(N, 3), replace=False)
        clusters[:, moved] = False
        clusters[0, moved] = True
        idx = np.array([off[0], off[C - 1]], dtype=np.int32) if C > 1 else np.array([off[0]], dtype=np.int32)
        rows = np.stack([rng.integers(0, 30, size=counts[0].shape[1:]).astype(np.float32) for _ in idx])
        objs = np.sort(rng.choice(N, size=min(N, 5), replace=False)).astype(np.int32)
        pick = rng.integers(0, C + 1, size=(objs.size, wl.source.shape[1]))
        src_rows = pick[..., None] == np.arange(C)
        for use in ((1, 1, 1), (1, 1, 0), (1, 0, 1), (0, 1, 1), (0, 1, 0)):
            eng.copy_slot(1, 0)
            eng.set_slot_delta(1, groups_component=0, groups=clusters if use[0] else None,
                               count_idx=idx if use[1] else None, count_rows=rows if use[1] else None, update_probs=True,
                               source_objects=objs if use[2] else None, source_rows=src_rows if use[2] else None)
            # (slot 1 is read back, then rebuilt call by call from the same start)
            got = (eng.mixture_loglik(1), eng.weights_normalized(1), [eng.get_counts(1, c) for c in range(C)],
                   [eng.get_probs(1, c) for c in range(C)], eng.get_source_rows(1, objs), eng.source_prior(1))
            eng.copy_slot(1, 0)
            if use[0]:
                eng.set_groups(1, 0, clusters)
            if use[1]:
                eng.set_counts_rows(1, idx, rows, update_probs=True)
            if use[2]:
                eng.set_source_rows(1, objs, src_rows)
            want = (eng.mixture_loglik(1), eng.weights_normalized(1), [eng.get_counts(1, c) for c in range(C)],
                    [eng.get_probs(1, c) for c in range(C)], eng.get_source_rows(1, objs), eng.source_prior(1))
            assert got[0] == want[0] or (np.isnan(got[0]) and np.isnan(want[0])), (name, use)
            assert np.array_equal(got[1], want[1], equal_nan=True), (name, use, "weights")
            for c in range(C):
                assert np.array_equal(got[2][c], want[2][c]) and np.array_equal(got[3][c], want[3][c]), (name, use, c)
            assert np.array_equal(got[4], want[4]) and np.array_equal(got[5], want[5], equal_nan=True), (name, use, "source")
        with pytest.raises(Exception, match="out of range"):
            eng.set_slot_delta(1, groups=clusters, count_idx=[10 ** 6], count_rows=rows[:1], update_probs=False)
        assert np.array_equal(eng.weights_normalized(1), orc.normalize_weights(wl.weights, orc.has_components([clusters] + list(wl.groups[1:]))))
    finally:
        eng.close()
