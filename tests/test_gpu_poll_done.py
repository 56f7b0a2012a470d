"""Completion by flag (signal_done / wait_done, include/sbe_engine.h "Threading"): latency-bound calls end when the last
block of their final kernel stores a sequence number into host-mapped memory, and the host reads the results -- also
in host-mapped memory -- right after seeing it.  If the flag could overtake a result, the host would read the value the
PREVIOUS call left in that place.  These tests make every call's result differ from the previous one's (the state is
flipped between two variants before every call) and repeat a few thousand times per path: one stale read fails them.
SBE_POLL_DONE=0 (the runtime's stream wait everywhere) must give the same values."""
import os
import subprocess
import sys

import numpy as np
import pytest

from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload
from tests._fake_engine import FakeEngine

pytestmark = pytest.mark.gpu
REPS = int(os.environ.get("SBE_POLL_REPS", "20000"))        # (soak runs: SBE_POLL_REPS=2000000)


def _setup(shape):
    from oracle import sbayes_oracle as orc
    wl = make_workload("poll", shape=shape)
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
    eng.set_option(deferred_checks=True)            # setters do not synchronise: the result call is the only wait
    for c in range(len(wl.groups)):
        eng.update_probs(0, c)
    return eng, fake, wl, counts


@pytest.mark.parametrize("shape", [(40, 12, 4, 2, (), False), (1000, 200, 10, 5, (), False)], ids=["small", "headline"])
def test_flipped_state_never_reads_a_stale_result(shape):
    eng, fake, wl, counts = _setup(shape)
    try:
        rng = np.random.default_rng(1)
        F, C = wl.weights.shape
        w2 = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
        weights = [wl.weights, w2]
        rows = [counts[0][:1].copy(), counts[0][:1] + 3.0]                     # the first cluster's count row, two variants
        objs = np.arange(min(5, wl.source.shape[0]))
        src = [wl.source[objs].copy(), np.roll(wl.source[objs], 1, axis=-1)]
        src[1][~wl.features[objs].any(-1)] = False
        # expected values of both variants, from the engine itself with the runtime's wait semantics (first call of each
        # variant after a full synchronisation), cross-checked against the double
        want = {"mix": [], "lh_all": [], "sprior": [], "slf": []}
        for v in (0, 1):
            eng.set_weights(0, weights[v]); eng.set_counts_rows(0, [0], rows[v]); eng.set_source_rows(0, objs, src[v])
            fake.set_weights(0, weights[v]); fake.set_counts_rows(0, [0], rows[v]); fake.set_source_rows(0, objs, src[v])
            eng.update_probs(0, 0)
            eng.sync()
            want["mix"].append(eng.mixture_loglik(0)); eng.sync()
            want["lh_all"].append(eng.collapsed_loglik_all(0)); eng.sync()
            want["sprior"].append(eng.source_prior(0)); eng.sync()
            want["slf"].append(eng.source_lh_by_feature(0)); eng.sync()
            np.testing.assert_allclose(want["lh_all"][v], fake.collapsed_loglik_all(0), rtol=2e-6, atol=1e-6)
            np.testing.assert_allclose(want["sprior"][v], fake.source_prior(0), rtol=2e-6, atol=1e-6)
        assert want["mix"][0] != want["mix"][1] and not np.array_equal(want["lh_all"][0], want["lh_all"][1])
        assert not np.array_equal(want["sprior"][0], want["sprior"][1]) and not np.array_equal(want["slf"][0], want["slf"][1])
        for i in range(REPS):
            v = i & 1
            eng.set_weights(0, weights[v])
            eng.set_source_rows(0, objs, src[v])
            assert eng.source_lh_by_feature(0).tobytes() == want["slf"][v].tobytes(), i
            assert eng.source_prior(0).tobytes() == want["sprior"][v].tobytes(), i
            eng.set_counts_rows(0, [0], rows[v])
            assert eng.collapsed_loglik_all(0).tobytes() == want["lh_all"][v].tobytes(), i
            eng.update_probs(0, 0)
            assert eng.mixture_loglik(0) == want["mix"][v], i
    finally:
        eng.close()


def test_one_call_steps_alternate_without_stale_results():
    """sbe_step / sbe_step_delta: the same two proposals alternately; per-group values, mixture value and flags of every
    step equal the first evaluation of that proposal."""
    eng, fake, wl, counts = _setup((300, 70, 6, 3, (), False))
    try:
        for c in range(len(wl.groups)):
            eng.set_groups(1, c, wl.groups[c]); eng.set_counts(1, c, counts[c])
        eng.set_source(1, wl.source); eng.set_weights(1, wl.weights)
        cl = [wl.groups[0].copy(), wl.groups[0].copy()]
        moved = np.flatnonzero(cl[0].any(axis=0))[:4]
        cl[1][:, moved] = False
        cl[1][(cl[0][:, moved].argmax(axis=0) + 1) % cl[0].shape[0], moved] = True
        first = {}
        cur, cand = 0, 1
        for i in range(REPS):
            v = i & 1
            glh, mix, changed = eng.step(cur, cand, clusters=cl[v])
            key = (v, i > 0)                                  # (the very first step starts from the unmoved state)
            if key not in first:
                first[key] = (glh.copy(), mix, changed.copy())
            else:
                assert glh.tobytes() == first[key][0].tobytes() and mix == first[key][1] and np.array_equal(changed, first[key][2]), i
            cur, cand = cand, cur
        assert first[(0, True)][1] != first[(1, True)][1]
    finally:
        eng.close()


def test_same_values_with_the_runtime_wait():
    """SBE_POLL_DONE=0 in a fresh process: the fallback path (every wait is hipStreamSynchronize) gives the same numbers."""
    code = ("import numpy as np; from sbayes_amd.engine import Engine; from sbayes_amd.synthetic import make_workload, make_state\n"
            "from oracle import sbayes_oracle as orc\n"
            "wl = make_workload('p', shape=(120, 30, 5, 3, (), False)); n_groups = [g.shape[0] for g in wl.groups]\n"
            "counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)\n"
            "eng = Engine(wl.features, n_groups, n_slots=2)\n"
            "for c in range(len(wl.groups)):\n"
            "    eng.set_concentration(c, wl.concentration[c]); eng.set_groups(0, c, wl.groups[c]); eng.set_counts(0, c, counts[c]); eng.update_probs(0, c)\n"
            "eng.set_source(0, wl.source); eng.set_weights(0, wl.weights)\n"
            "print(repr(eng.mixture_loglik(0)), eng.collapsed_loglik_all(0).tobytes().hex(), eng.source_prior(0).tobytes().hex())\n")
    outs = []
    for poll in ("0", "1"):
        env = dict(os.environ, SBE_POLL_DONE=poll)
        outs.append(subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True, env=env,
                                   cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), timeout=300).stdout)
    assert outs[0] == outs[1] and len(outs[0]) > 100
