"""T1: HIP engine (through the C ABI) == CPU oracle == reference golden vectors.
Bit-exact for tables, counts, ids and per-observation values; <= 1e-10 relative for the
summed mixture log-likelihood (north_star tolerance); <= 1e-6 relative for the float32
collapsed likelihood."""
import json

import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd.engine import (LOG_PER_OBS, LOG_PRODUCT, MIXTURE_ONEHOT, MIXTURE_PACKED, MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2,
                               MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA, Engine, EngineError)
from sbayes_amd.synthetic import make_state, make_workload
from tests._fixtures import GOLDEN, crc, load_json, load_npz

pytestmark = pytest.mark.gpu

MIX_RTOL = 1e-10       # north_star: summed log-likelihood within 1e-10 relative
COLLAPSED_RTOL = 1e-6  # float32-limited

NPZ = ["cfg1", "south_america", "test_files"]


def engine_for(fx, n_slots=2):
    eng = Engine(fx.features, [g.shape[0] for g in fx.groups], n_slots=n_slots)
    for c in range(fx.n_comp):
        eng.set_concentration(c, fx.conc[c])
    return eng


def load_fixture_state(eng, fx, slot=0):
    for c in range(fx.n_comp):
        eng.set_groups(slot, c, fx.groups[c])
    eng.set_source(slot, fx.source)
    eng.recount(slot)
    for c in range(fx.n_comp):
        eng.update_probs(slot, c)
    eng.set_weights(slot, fx.weights)


@pytest.mark.parametrize("name", NPZ)
def test_ingest_counts_tables_bit_exact(name):
    fx = load_npz(name)
    with engine_for(fx) as eng:
        assert np.array_equal(eng.na_values(), fx.na_values)
        assert eng.info()["n_na"] == int(fx.na_values.sum())
        load_fixture_state(eng, fx)
        for c in range(fx.n_comp):
            assert np.array_equal(eng.get_counts(0, c), fx.counts[c])          # a9
            assert np.array_equal(eng.get_probs(0, c), fx.probs[c])            # a4 via a3
        assert np.array_equal(eng.weights_normalized(0), fx.z["weights_normalized"])   # a5


@pytest.mark.parametrize("name", NPZ)
def test_dense_outputs_bit_exact(name):
    fx = load_npz(name)
    with engine_for(fx) as eng:
        load_fixture_state(eng, fx)
        assert np.array_equal(eng.likelihood_per_component(0), fx.z["lh_per_component"])      # a3
        assert np.array_equal(eng.observation_lh(0), fx.z["obs_lh"])                           # a6
        assert np.array_equal(eng.likelihood_per_component_exact(0), fx.z["lh_exact"])         # a2


@pytest.mark.parametrize("name", NPZ)
@pytest.mark.parametrize("kernel", [MIXTURE_PACKED, MIXTURE_ONEHOT, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS,
                                    MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED_TUPLE_MFMA])
@pytest.mark.parametrize("log_mode", [LOG_PER_OBS, LOG_PRODUCT])
def test_mixture_loglik(name, kernel, log_mode):
    fx = load_npz(name)
    with engine_for(fx) as eng:
        load_fixture_state(eng, fx)
        eng.set_option(kernel=kernel, log_mode=log_mode)
        ll = eng.mixture_loglik(0)
        if kernel == MIXTURE_PACKED_TUPLE_MFMA:
            # every fixture takes the matrix-pipe form when forced: 16 slots x <= 8 tuples per block, or -- south_america's 17
            # group tuples (3 clusters + none, universal, 6 families + none) -- the wide form of round 6: 4 slots x <= 32 tuples
            assert "k_mixture_tuple_mfma" in eng.last_mixture_kernel()
            # (one state per launch: the few-tuple fixtures take four slots per block with ONE M tile -- fewer rounds x passes x M tiles)
            assert ("4 slots x M tiles 3" if name == "south_america" else "4 slots x M tiles 1") in eng.last_mixture_kernel(), eng.last_mixture_kernel()
        want = fx.meta["mixture_ll"]
        assert abs(ll - want) <= MIX_RTOL * abs(want), (ll, want)
        assert eng.mixture_loglik(0) == ll      # deterministic reduction order


@pytest.mark.parametrize("name", NPZ)
def test_collapsed_loglik(name):
    fx = load_npz(name)
    with engine_for(fx) as eng:
        load_fixture_state(eng, fx)
        total = 0.0
        for c in range(fx.n_comp):
            per_group, per_feature = eng.collapsed_loglik(0, c, per_feature=True)
            assert per_feature.dtype == np.float32
            np.testing.assert_allclose(per_feature, fx.dcl[c], rtol=2e-6, atol=1e-6)
            np.testing.assert_allclose(per_group, fx.group_lh[c], rtol=COLLAPSED_RTOL)
            total += per_group.sum()
        assert abs(total - fx.meta["collapsed_ll"]) <= COLLAPSED_RTOL * abs(fx.meta["collapsed_ll"])
        # all components in one call: the same per-group values, bit for bit
        every = eng.collapsed_loglik_all(0)
        assert np.array_equal(every, np.concatenate([eng.collapsed_loglik(0, c) for c in range(fx.n_comp)]))


@pytest.mark.parametrize("name", NPZ)
def test_component_lh_partial_update(name):
    """a1 literal contract incl. H7: strided out view, stale rows, zeroed no-group rows."""
    fx = load_npz(name)
    with engine_for(fx) as eng:
        buf = fx.z["partial_before"].copy()
        eng.component_lh(fx.probs[0], fx.groups[0], fx.z["partial_changed"], buf[..., 1])
        assert np.array_equal(buf, fx.z["partial_after"])
        # all groups changed, every component, float64 tables too
        for c in range(fx.n_comp):
            for dtype in (np.float32, np.float64):
                out = np.full((fx.features.shape[0], fx.features.shape[1]), -7.0)
                want = orc.compute_component_likelihood(fx.features, fx.probs[c].astype(dtype), fx.groups[c],
                                                        np.arange(fx.groups[c].shape[0]), out.copy())
                got = eng.component_lh(fx.probs[c].astype(dtype), fx.groups[c], np.arange(fx.groups[c].shape[0]), out)
                assert got is out and np.array_equal(out, want)
        # empty changed list: only the no-group rows are zeroed
        out = np.full((fx.features.shape[0], fx.features.shape[1]), 3.0)
        want = orc.compute_component_likelihood(fx.features, fx.probs[0], fx.groups[0], np.array([], dtype=np.int64), out.copy())
        eng.component_lh(fx.probs[0], fx.groups[0], np.array([], dtype=np.int64), out)
        assert np.array_equal(out, want)


@pytest.mark.parametrize("name", NPZ)
def test_delta_counts(name):
    """a9 delta form: update_feature_counts on an object subset (counts.py:55-95)."""
    fx = load_npz(name)
    with engine_for(fx) as eng:
        load_fixture_state(eng, fx, slot=0)
        eng.copy_slot(1, 0)
        subset = fx.z["delta_subset"]
        eng.set_groups(1, 0, fx.z["delta_clusters_new"])
        eng.set_source_rows(1, subset, fx.z["delta_source_new"][subset])
        changed = eng.update_counts(1, 0, subset)
        off = eng.group_offsets
        for c in range(fx.n_comp):
            want = fx.z[f"delta_counts_{c}"]
            assert np.array_equal(eng.get_counts(1, c), want)
            assert np.array_equal(changed[off[c]:off[c + 1]], np.any(want != fx.counts[c], axis=(1, 2)))
            assert np.array_equal(eng.get_counts(0, c), fx.counts[c])      # old slot untouched
        # two-call form gives the same counts
        eng.copy_slot(1, 0)
        eng.accumulate_counts(1, subset, -1)
        eng.set_groups(1, 0, fx.z["delta_clusters_new"])
        eng.set_source_rows(1, subset, fx.z["delta_source_new"][subset])
        eng.accumulate_counts(1, subset, +1)
        for c in range(fx.n_comp):
            assert np.array_equal(eng.get_counts(1, c), fx.z[f"delta_counts_{c}"])


def test_conditional_effect_mean_tempered():
    fx = load_npz("cfg1")
    with engine_for(fx) as eng:
        load_fixture_state(eng, fx)
        eng.update_probs(0, 0)
        assert np.array_equal(eng.get_probs(0, 0), fx.z["cem_plain"])
        eng.update_probs(0, 0, temperature=2.5, prior_temperature=1.7,
                         unif_counts=fx.states_per_feature.astype(float))
        assert np.array_equal(eng.get_probs(0, 0), fx.z["cem_temp"])


@pytest.mark.parametrize("name", ["headline", "stress"])
def test_big_synthetic_against_reference_digests(name):
    """BASELINE.json configs[2] / configs[4] shapes: reference outputs pinned by CRC/scalars."""
    meta = load_json(name)
    wl = make_workload(name)
    with Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=1) as eng:
        for c in range(wl.n_components):
            eng.set_concentration(c, wl.concentration[c])
            eng.set_groups(0, c, wl.groups[c])
        eng.set_source(0, wl.source)
        eng.recount(0)
        for c in range(wl.n_components):
            eng.update_probs(0, c)
        eng.set_weights(0, wl.weights)
        assert [crc(eng.get_counts(0, c)) for c in range(wl.n_components)] == meta["counts_crc"]
        assert [crc(eng.get_probs(0, c)) for c in range(wl.n_components)] == meta["probs_crc"]
        assert crc(eng.weights_normalized(0)) == meta["w_crc"]
        lh = eng.likelihood_per_component(0)
        assert crc(lh) == meta["lh_crc"]
        assert crc(eng.observation_lh(0)) == meta["obs_crc"]
        assert crc(eng.likelihood_per_component_exact(0)) == meta["lh_exact_crc"]
        want = meta["mixture_ll"]
        for kernel in (MIXTURE_PACKED, MIXTURE_ONEHOT, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_ONEHOT_GENERAL) + ((MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA) if name == "headline" else ()):
            for log_mode in (LOG_PER_OBS, LOG_PRODUCT):
                eng.set_option(kernel=kernel, log_mode=log_mode)
                ll = eng.mixture_loglik(0)
                assert abs(ll - want) <= MIX_RTOL * abs(want), (kernel, log_mode, ll, want)
        total = sum(eng.collapsed_loglik(0, c).sum() for c in range(wl.n_components))
        assert abs(total - meta["collapsed_ll"]) <= COLLAPSED_RTOL * abs(meta["collapsed_ll"])
        for c in range(wl.n_components):
            np.testing.assert_allclose(eng.collapsed_loglik(0, c), np.array(meta["group_lh"][c]), rtol=COLLAPSED_RTOL)


def test_batch_of_states_matches_oracle():
    """Batched entry point: B distinct states over one resident feature block."""
    wl = make_workload("cfg1")
    B = 5
    with Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=B) as eng:
        for c in range(wl.n_components):
            eng.set_concentration(c, wl.concentration[c])
        want = []
        for b in range(B):
            clusters, weights, source = make_state(wl.features, wl.groups[1:], wl.clusters.shape[0], seed=100 + b)
            groups = [clusters] + wl.groups[1:]
            eng.load_state(b, groups, weights, source=source)
            for c in range(wl.n_components):
                eng.update_probs(b, c)
            counts = orc.recalculate_feature_counts(wl.features, groups, source)
            want.append(orc.mixture_loglik(wl.features, wl.na_values, groups, counts, wl.concentration, weights))
        got = eng.mixture_loglik_batch(0, B)
        np.testing.assert_allclose(got, np.array(want), rtol=MIX_RTOL)
        for kernel in (MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_ONEHOT, MIXTURE_ONEHOT_GENERAL):
            eng.set_option(kernel=kernel)
            np.testing.assert_allclose(eng.mixture_loglik_batch(0, B), np.array(want), rtol=MIX_RTOL)
        eng.set_option(kernel=MIXTURE_PACKED)
        assert len(set(np.round(got, 6))) == B
        eng.mixture_loglik_batch_async(1, 3)
        eng.sync()
        assert np.array_equal(eng.fetch_results(1, 3), got[1:4])


def test_edge_cases():
    # every observation NA -> LL = 0; object in no group of any component handled
    feats = np.zeros((6, 5, 3), dtype=bool)
    with Engine(feats, [2, 1], n_slots=1) as eng:
        groups = [np.zeros((2, 6), dtype=bool), np.ones((1, 6), dtype=bool)]
        groups[0][0, :2] = True
        eng.load_state(0, groups, np.full((5, 2), 0.5, dtype=np.float32),
                       probs=[np.full((2, 5, 3), 1 / 3, dtype=np.float32), np.full((1, 5, 3), 1 / 3, dtype=np.float32)])
        assert eng.mixture_loglik(0) == 0.0
        assert eng.info()["n_na"] == 30
    # a zero-probability observed state gives -inf like the reference's log(0)
    feats = np.zeros((3, 1, 2), dtype=bool)
    feats[0, 0, 0] = feats[1, 0, 0] = feats[2, 0, 1] = True
    for log_mode in (LOG_PER_OBS, LOG_PRODUCT):
        with Engine(feats, [1], n_slots=1) as eng:
            eng.set_option(log_mode=log_mode)
            eng.load_state(0, [np.ones((1, 3), dtype=bool)], np.ones((1, 1), dtype=np.float32),
                           probs=[np.array([[[1.0, 0.0]]], dtype=np.float32)])
            assert eng.mixture_loglik(0) == -np.inf
            assert np.array_equal(eng.observation_lh(0).ravel(), [1.0, 1.0, 0.0])
    # non-one-hot features are rejected (the reference's encode_states never produces them)
    bad = np.zeros((2, 2, 3), dtype=bool)
    bad[0, 0, :2] = True
    with pytest.raises(EngineError, match="not one-hot"):
        Engine(bad, [1])
    # state errors are reported, not computed around
    with Engine(feats, [1], n_slots=1) as eng:
        with pytest.raises(EngineError, match="groups not set|not set"):
            eng.mixture_loglik(0)
    # normalize() assert: a row with non-positive sum
    with Engine(feats, [1], n_slots=1) as eng:
        eng.set_concentration(0, np.zeros((1, 2)))
        eng.set_groups(0, 0, np.zeros((1, 3), dtype=bool))
        eng.set_source(0, np.zeros((3, 1, 1), dtype=bool))
        eng.recount(0)
        with pytest.raises(EngineError, match="non-positive sum"):
            eng.update_probs(0, 0)


def test_known_answers_on_device():
    """The reference's commented-out TestLikelihood cases (test/test_model.py:157-207)."""
    with open(GOLDEN / "known_answers.json") as fh:
        cases = {c["name"]: c for c in json.load(fh)}
    feats = np.array([[[True, False]], [[True, False]], [[False, True]]])
    groups = np.ones((1, 3), dtype=bool)
    with Engine(feats, [1], n_slots=1) as eng:
        for name, table in [("uniform_0.125", (0.5, 0.5)), ("skewed_0.25x0.75^2", (0.75, 0.25)), ("zero", (1.0, 0.0))]:
            out = np.empty((3, 1))
            eng.component_lh(np.array([[table]], dtype=np.float32), groups, np.arange(1), out)
            assert out.ravel().tolist() == cases[name]["per_obs"]
            assert out.prod() == cases[name]["expected"]
        eng.load_state(0, [groups], np.ones((1, 1), dtype=np.float32), probs=[np.array([[[0.75, 0.25]]], dtype=np.float32)])
        assert np.isclose(eng.mixture_loglik(0), np.log(0.25 * 0.75 ** 2), rtol=1e-14)
    with Engine(feats, [1, 1], n_slots=1) as eng:
        cl = np.array([[True, True, False]])
        eng.load_state(0, [cl, groups], np.array([[0.5, 0.5]], dtype=np.float32),
                       probs=[np.array([[[1.0, 0.0]]], dtype=np.float32), np.array([[[0.5, 0.5]]], dtype=np.float32)])
        assert eng.weights_normalized(0).tolist() == cases["weights_pattern"]["expected"]


def test_deferred_checks_surface_at_next_sync():
    """SBE_OPT_DEFERRED_CHECKS: the normalize() assert raised by a kernel is reported by the next
    synchronizing call instead of stalling the state-setting call."""
    feats = np.zeros((3, 1, 2), dtype=bool)
    feats[0, 0, 0] = feats[1, 0, 0] = feats[2, 0, 1] = True
    with Engine(feats, [1], n_slots=1) as eng:
        eng.set_option(deferred_checks=True)
        eng.set_concentration(0, np.zeros((1, 2)))
        eng.set_groups(0, 0, np.zeros((1, 3), dtype=bool))
        eng.set_source(0, np.zeros((3, 1, 1), dtype=bool))
        eng.recount(0)
        eng.update_probs(0, 0)                      # does not raise here
        with pytest.raises(EngineError, match="non-positive sum"):
            eng.sync()
        eng.sync()                                  # reported once
        eng.set_concentration(0, np.ones((1, 2)))
        eng.set_groups(0, 0, np.ones((1, 3), dtype=bool))
        eng.update_probs(0, 0)                      # counts 0 + prior 1 -> tables (0.5, 0.5)
        eng.set_weights(0, np.ones((1, 1), dtype=np.float32))
        assert np.isclose(eng.mixture_loglik(0), 3 * np.log(0.5), rtol=1e-14)
        eng.set_option(deferred_checks=False)


def test_lgamma_accuracy():
    """The engine's lgamma (recurrence up to y >= 8, Stirling series, two fp64 logs) against SciPy's gammaln over the
    arguments the Dirichlet-categorical terms produce -- concentrations from 1e-6 to 1e4, counts + concentrations up
    to 1e7 -- and the edge of its domain: absolute error <= 4e-15 * max(1, |lgamma|) (+ 1e-15 near the zeros at 1, 2)."""
    from scipy.special import gammaln
    rng = np.random.default_rng(1)
    x = np.concatenate([
        np.exp(rng.uniform(np.log(1e-6), np.log(1e4), 200000)),
        rng.integers(0, 5000, 100000) + np.exp(rng.uniform(np.log(1e-3), np.log(50.0), 100000)),
        rng.uniform(0.5, 9.5, 100000), 1.0 + rng.uniform(-1e-6, 1e-6, 1000), 2.0 + rng.uniform(-1e-6, 1e-6, 1000),
        np.array([1.0, 2.0, 0.5, 1.5, 7.999999999, 8.0, 8.000000001, 1e-300, 1e7, 1e15, 170.5]),
    ])
    with Engine(np.zeros((1, 1, 1), dtype=bool), [1], n_slots=1) as eng:
        got = eng.test_lgamma(x)
        want = gammaln(x)
        err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        assert err.max() <= 1e-14, (err.max(), x[err.argmax()], got[err.argmax()], want[err.argmax()])
        assert abs(got[x == 1.0][0]) <= 1e-14 and abs(got[x == 2.0][0]) <= 1e-14
        special = eng.test_lgamma(np.array([0.0, -0.5, np.inf, np.nan]))
        assert special[0] == np.inf and np.isfinite(special[1]) and special[2] == np.inf and np.isnan(special[3])


def test_fast_log_accuracy():
    """The fp64 log of the group-tuple table build: < 1 ulp against NumPy's log over positive normal
    doubles (probability-like values, the whole exponent range, values around 1), and the special
    cases go through the library log (0 -> -inf, negative -> NaN)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([
        rng.random(200000),                                   # (0, 1)
        np.float32(rng.random(50000)).astype(np.float64) * np.float32(rng.random(50000)).astype(np.float64),
        np.exp(rng.uniform(-700, 700, 50000)),                # whole exponent range
        1.0 + rng.uniform(-1e-3, 1e-3, 50000),                # around 1 (cancellation)
        np.array([1.0, 0.5, 2.0, np.sqrt(2), np.sqrt(0.5), np.nextafter(1, 2), np.nextafter(1, 0), 2.2250738585072014e-308]),
    ])
    x = x[x > 0]
    with Engine(np.zeros((1, 1, 1), dtype=bool), [1], n_slots=1) as eng:
        fast, lib = eng.test_fast_log(x)
        want = np.log(x)
        ulp = np.spacing(np.abs(want))
        ulp[want == 0] = np.spacing(1.0) / 2
        err = np.abs(fast - want) / ulp
        assert err.max() <= 1.0, (err.max(), x[err.argmax()])
        assert fast[x == 1.0][0] == 0.0
        special, _ = eng.test_fast_log(np.array([0.0, -1.0, np.inf, np.nan, 5e-324]))
        assert special[0] == -np.inf and np.isnan(special[1]) and special[2] == np.inf and np.isnan(special[3])
        assert np.isclose(special[4], np.log(5e-324), rtol=1e-15)
        # table-driven log of k_mixture_tuple64: <= 1 ulp + 2^-53 absolute (the absolute term shows only where
        # k*ln2 + log c cancels, just below 1); log(1) = 0 exactly; special values through the library log
        xs = np.concatenate([x, 1.0 - rng.random(100000) * 2.0 ** -8, 1.0 + rng.random(100000) * 2.0 ** -7,
                             1.0 + rng.uniform(-1e-9, 1e-9, 10000)])
        tab = eng.test_tab_log(xs)
        want = np.log(xs)
        ulp = np.spacing(np.abs(want))
        ulp[want == 0] = 0.0
        err = np.abs(tab - want) / (ulp + 2.0 ** -53)
        assert err.max() <= 1.5, (err.max(), xs[err.argmax()])        # 1.5: NumPy's own log is within 0.5 ulp
        assert tab[xs == 1.0][0] == 0.0
        special = eng.test_tab_log(np.array([0.0, -1.0, np.inf, np.nan, 5e-324]))
        assert special[0] == -np.inf and np.isnan(special[1]) and special[2] == np.inf and np.isnan(special[3])
        assert np.isclose(special[4], np.log(5e-324), rtol=1e-15)
