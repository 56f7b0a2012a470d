"""The BENCHMARKED path under `-m gpu`, as benchmarked: the headline workload (BASELINE.json
configs[2]: 1000 x 200 x 10, K = 5, C = 2) through the engine's DEFAULT kernel selection, every slot of the launch a DISTINCT
state, against the oracle.  Reference expression: sbayes/sampling/loggers.py:355-357 over sbayes/model/likelihood.py:104-133,
171-190.  Tolerance: 1e-10 relative (north_star).

The selection at this shape (sbe_engine_internal.hip.h: launch_mixture / mfma_geometry): 31 states per launch -> k_mixture_tuple64
(vector pipe); from 32 states on (6.4 M observations per launch) k_mixture_tuple_mfma -- with FOUR slots per block while that
geometry needs fewer rounds x passes x M tiles (32 .. 512 states: 4 slots x one M tile), with 16 slots per block beyond (513, 1024:
MT = 3 x 16 k-blocks of 64 objects x 63 column tiles -- bench.py's kernel and geometry).  A
fixed-seed 60-second slice of tools/fuzz_gpu.py's "big" generator follows (the open-ended fuzzer itself is not part of the suite)."""
import sys
import time
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from oracle import sbayes_oracle as orc                               # noqa: E402  (checker)
from sbayes_amd.engine import MIXTURE_PACKED, Engine                   # noqa: E402
from sbayes_amd.synthetic import make_state, make_workload            # noqa: E402

pytestmark = pytest.mark.gpu

N_SLOTS = 1024


@pytest.fixture(scope="module")
def headline_engine():
    """1024 distinct headline states: slot b = make_state(seed 5000 + b) -- its own clusters, weights and source assignment (the
    reference's recipe: sample_categorical(normalize_weights(...)), conditionals.py:370-384) -- counts and tables on the device."""
    wl = make_workload("headline")
    conf_groups = wl.groups[1:]
    K = wl.clusters.shape[0]
    eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=N_SLOTS)
    for c in range(wl.n_components):
        eng.set_concentration(c, wl.concentration[c])
    eng.set_option(deferred_checks=True)
    states = []
    for b in range(N_SLOTS):
        clusters, weights, source = make_state(wl.features, conf_groups, K, 5000 + b)
        eng.load_state(b, [clusters, *conf_groups], weights, source=source)
        for c in range(wl.n_components):
            eng.update_probs(b, c)
        states.append((clusters, weights, source))
    eng.set_option(deferred_checks=False)
    yield wl, eng, states
    eng.close()


def oracle_value(wl, state):
    clusters, weights, source = state
    groups = [clusters, *wl.groups[1:]]
    counts = orc.recalculate_feature_counts(wl.features, groups, source)
    return float(orc.mixture_loglik(wl.features, wl.na_values, groups, counts, wl.concentration, weights))


@pytest.mark.parametrize("B,kernel", [(31, "k_mixture_tuple64"), (32, "k_mixture_tuple_mfma"), (319, "k_mixture_tuple_mfma"), (512, "k_mixture_tuple_mfma"),
                                      (513, "k_mixture_tuple_mfma"), (1024, "k_mixture_tuple_mfma")])
def test_headline_default_selection_distinct_slots_vs_oracle(headline_engine, B, kernel):
    wl, eng, states = headline_engine
    eng.set_option(kernel=MIXTURE_PACKED)                                      # the default: what bench.py runs
    got = eng.mixture_loglik_batch(0, B)
    assert kernel in eng.last_mixture_kernel(), eng.last_mixture_kernel()      # both sides of the threshold
    if "mfma" in kernel:                       # 6 tuples: K = 5 clusters + "no cluster", one universal group
        assert ("4 slots x M tiles 1" if B <= 512 else "16 slots x M tiles 3") in eng.last_mixture_kernel(), eng.last_mixture_kernel()
    assert np.all(np.isfinite(got)) and len(set(got.tolist())) == B            # every slot its own state
    rng = np.random.default_rng(B)
    picks = np.arange(B) if B <= 32 else np.unique(np.concatenate([[0, 1, 15, 16, B - 17, B - 16, B - 1], rng.integers(0, B, size=24)]))
    assert picks.size >= 16
    want = np.array([oracle_value(wl, states[b]) for b in picks])
    np.testing.assert_allclose(got[picks], want, rtol=1e-10, atol=0.0)
    # the asynchronous route bench.py times returns the same bits; so does a second launch (fixed reduction order)
    eng.mixture_loglik_batch_async(0, B)
    assert np.array_equal(eng.fetch_results(0, B), got)
    # and the two kernel forms agree with each other on EVERY slot far inside the tolerance (one of them is oracle-checked above)
    other = eng.mixture_loglik_batch(0, 31)                      # (the vector-pipe form)
    n = min(B, other.size)
    np.testing.assert_allclose(got[:n], other[:n], rtol=1e-12)


def test_readback_of_resident_state_matches_what_was_loaded(headline_engine):
    """sbe_get_group_ids / sbe_get_weights / sbe_get_source_rows (what bench.py's parity gate rebuilds a slot's state from)
    return what the DEVICE holds, and that is what was loaded."""
    wl, eng, states = headline_engine
    for b in (0, 511, 1023):
        clusters, weights, source = states[b]
        ids = eng.get_group_ids(b, 0)
        assert np.array_equal(ids, np.where(clusters.any(axis=0), clusters.argmax(axis=0), -1))
        assert np.array_equal(eng.get_group_ids(b, 1), np.zeros(wl.shape[0], dtype=np.int32))       # the universal group
        assert np.array_equal(eng.get_weights(b), weights)
        rows = eng.get_source_rows(b, np.arange(wl.shape[0], dtype=np.int32))
        assert np.array_equal(rows.astype(bool), source)


def test_fixed_seed_slice_of_the_big_fuzz_generator():
    """60 seconds of tools/fuzz_gpu.py --big at a fixed seed: shapes of 600-4000 objects x 60-270 features, batches up to 700,
    through every kernel form (the matrix-pipe form forced wherever it applies) against the oracle at 1e-10 relative +
    1e-16 per observation (the fuzzer's tolerance: see one_case)."""
    from tools.fuzz_gpu import one_case
    rng = np.random.default_rng(20260601)
    stats = {"cases": 0, "evals": 0, "steps": 0, "gibbs": 0}
    t0 = time.time()
    while time.time() - t0 < 60.0 or stats["cases"] < 2:
        one_case(rng, stats, big=True)
    assert stats["cases"] >= 2 and stats["evals"] > 0
    print(f"[fuzz slice] {stats} in {time.time() - t0:.0f} s")


def test_bench_line_is_self_consistent_and_self_proving():
    """bench.py at a small batch, legs included: the fraction is a fraction, the kernel fits inside the step,
    the TIMED results and both legs' results passed the oracle gate (the run fails otherwise), the legs report their figures."""
    import json
    import subprocess
    res = subprocess.run([sys.executable, str(REPO / "bench.py"), "--batch", "512", "--steps", "5", "--warmup", "2", "--no-secondary",
                          "--no-cpu-baseline", "--legs"], capture_output=True, text=True, timeout=600, cwd=str(REPO))
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    rf = line["roofline"]
    assert 0.0 < rf["frac"] <= 1.0 and rf["frac_contract"] >= rf["frac"]
    assert rf["kernel_avg_us"] <= line["ms_per_step"] * 1000.0 * 1.0005
    assert "k_mixture_tuple_mfma" in rf["kernel"] and line["parity_timed_kernel"]["kernel"] == rf["kernel"]
    assert line["parity_timed_kernel_max_rel_err"] <= 1e-10 and len(line["parity_timed_kernel"]["slots"]) >= 8
    hbm, chg = line["hbm_regime"], line["changing_tables"]
    assert hbm["evals_per_launch"] == 1024 and hbm["parity_max_rel_err"] <= 1e-10 and 0.0 < hbm["frac"] <= 1.0
    assert hbm["kernel_avg_us"] <= hbm["ms_per_step"] * 1000.0 * 1.0005
    assert chg["states"] == 512 and chg["parity_max_rel_err"] <= 1e-10 and chg["evals_per_s"] > 0
    assert line["value_hbm_regime"] == hbm["evals_per_s"] and line["value_changing_tables"] == chg["evals_per_s"]
