"""Parity of the one-call MCMC steps (sbe_step in both forms, sbe_gibbs_step) against the CPU oracle at shapes
with several feature tiles and several object chunks -- the fixed-seed slice of tools/fuzz_gpu.py that belongs in
the driver-run suite: counts, probability tables, per-group collapsed values, changed-group
flags, mixture scalar, new source rows, log_q / log_q_back.

Tolerances: integer counts, float32 tables, source rows and flags bit-exact; mixture log-likelihood 1e-10 relative
(north_star); per-group collapsed values 2e-6 relative (float32 per feature in the reference);
log_q / log_q_back 3e-6 relative (fp64 sums of the logs of float32 probabilities here, float32 sums in the reference)."""
import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd.engine import Engine, EngineError
from sbayes_amd.synthetic import make_state, make_workload
from tests.test_gpu_shapes import random_case

pytestmark = pytest.mark.gpu

# N, F, S, groups per component, na_rate, objects whose source rows change in the step
STEP_SHAPES = [
    (1000, 65,  4,  [3, 1],          0.03, 20),
    (1200, 130, 6,  [5, 1],          0.03, 30),
    (1000, 200, 10, [5, 1],          0.03, 20),      # the headline shape with random data
    (1500, 500, 5,  [4, 1],          0.02, 12),      # F = 500: eight tiles, the reference's np.prod underflows here (H5)
    (2003, 70,  3,  [7, 1],          0.03, 40),      # ragged object tail
    (1100, 129, 7,  [4, 1, 6],       0.05, 25),      # C = 3
    (1024, 200, 8,  [6, 1, 5, 3],    0.03, 16),      # C = 4
    (3000, 96,  12, [10, 1, 20, 20], 0.03, 30),      # thousands of group tuples: general kernel
    (1000, 64,  2,  [2, 1],          0.3,  300),     # > 256 changed source rows: falls to the call-by-call form
    (1300, 260, 3,  [3, 1],          0.0,  600),     # > 256 changed rows, five tiles
    (257,  130, 6,  [3, 1, 2, 2, 5], 0.02, 9),       # C = 5: runtime-C instantiation
    (400,  33,  33, [2, 1, 2],       0.02, 10),      # S = 33
    (1000, 200, 10, [5],             0.03, 20),      # C = 1 (clusters only; every object is in a cluster)
    (64,   300, 4,  [2, 1],          0.1,  64),      # every object's rows change
    (5,    2,   2,  [1, 1],          0.2,  2),       # test_files-sized
    (1777, 77,  40, [8, 1],          0.03, 20),      # S = 40
    (1000, 192, 10, [5, 1],          0.03, 1),       # exactly three full tiles; single moved object
    (1000, 193, 10, [5, 1],          0.03, 0),       # cluster move only, 1-wide ragged tile
    (4000, 80,  5,  [12, 1, 9],      0.03, 50),
    (1000, 200, 10, [5, 1],          0.5,  20),      # half of the observations NA
    (2500, 140, 9,  [20, 1],         0.03, 35),      # K = 20
    (1000, 65,  127, [2, 1],         0.03, 5),       # S = 127: largest state count of the 64-wide tuple kernel
]


def _expected(feats, na, groups, source, conc, weights):
    counts = orc.recalculate_feature_counts(feats, groups, source)
    probs = [orc.component_probs(counts[c], conc[c]) for c in range(len(groups))]
    glh = np.concatenate([orc.collapsed_group_logliks(counts[c], conc[c]) for c in range(len(groups))])
    with np.errstate(divide="ignore", invalid="ignore"):
        w = orc.normalize_weights(weights, orc.has_components(groups))
        lh = orc.likelihood_per_component(feats, na, groups, counts, conc)
        mix = np.log(orc.mixture_observation_lh(w, lh))[~na].sum()
    return counts, probs, glh, mix


def _check_candidate(eng, slot, glh, mix, want, tag):
    counts, probs, want_glh, want_mix = want
    for c in range(len(counts)):
        assert np.array_equal(eng.get_counts(slot, c), counts[c]), (tag, "counts", c)
        assert np.array_equal(eng.get_probs(slot, c), probs[c]), (tag, "probs", c)
    np.testing.assert_allclose(glh, want_glh, rtol=2e-6, atol=1e-6, err_msg=str(tag))
    assert np.isfinite(want_mix), tag
    assert abs(mix - want_mix) <= 1e-10 * abs(want_mix), (tag, mix, want_mix)


def _propose(rng, feats, na, groups, source, weights, n_rows, move):
    """A random MCMC-like delta: a few objects change cluster, `n_rows` objects (the moved ones among them) get new
    source rows over the components they still have, every third call also new weights."""
    N, F, _ = feats.shape
    C = len(groups)
    clusters = groups[0].copy()
    K = clusters.shape[0]
    moved = np.zeros(0, dtype=np.int64)
    if move and C >= 2:
        moved = np.unique(rng.integers(0, N, size=min(4, N)))
        for n in moved:
            clusters[:, n] = False
            k = int(rng.integers(0, K + 1))
            if k < K:
                clusters[k, n] = True
    new_groups = [clusters] + groups[1:]
    hc = orc.has_components(new_groups)
    objs = np.union1d(moved, rng.choice(N, size=min(n_rows, N), replace=False) if n_rows else moved).astype(np.int32)
    new_source = source.copy()
    rows = None
    if objs.size:
        idx = np.argmax(rng.random((objs.size, F, C)) * hc[objs][:, None, :], axis=-1)
        rows = np.eye(C, dtype=bool)[idx]
        rows[na[objs]] = False
        new_source[objs] = rows
    return clusters, new_groups, objs, rows, new_source


@pytest.mark.parametrize("shape", STEP_SHAPES, ids=lambda s: f"N{s[0]}F{s[1]}S{s[2]}C{len(s[3])}r{s[5]}")
def test_one_call_steps_against_oracle(shape):
    N, F, S, n_groups, na_rate, n_rows = shape
    rng = np.random.default_rng(1000 + N + 7 * F + 13 * S + n_rows)
    feats, groups, weights, source, conc = random_case(rng, N, F, S, n_groups, na_rate)
    na = ~feats.any(-1)
    C = len(n_groups)
    tag = f"N{N} F{F} S{S} groups{n_groups}"
    with Engine(feats, n_groups, n_slots=3) as eng:
        for c in range(C):
            eng.set_concentration(c, conc[c])
        eng.load_state(0, groups, weights, source=source)
        for c in range(C):
            eng.update_probs(0, c)
        want0 = _expected(feats, na, groups, source, conc, weights)
        got0 = eng.mixture_loglik(0)
        assert abs(got0 - want0[3]) <= 1e-10 * abs(want0[3]), (tag, "initial")

        # ---- sbe_step: three chained steps (accept, reject, accept), both forms on the same delta --------------
        cur, cand, spare = 0, 1, 2
        cur_groups, cur_source, cur_weights = groups, source, weights
        for i_step in range(3):
            clusters, new_groups, objs, rows, new_source = _propose(rng, feats, na, cur_groups, cur_source, cur_weights,
                                                                   n_rows, move=True)
            new_weights = cur_weights
            kw = {}
            if C >= 2 and not np.array_equal(clusters, cur_groups[0]):
                kw["clusters"] = clusters
            if objs.size:
                kw.update(changed_objects=objs, source_rows=rows)
            if i_step == 1:
                new_weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
                kw["weights"] = new_weights
            want = _expected(feats, na, new_groups, new_source, conc, new_weights)
            cur_counts = orc.recalculate_feature_counts(feats, cur_groups, cur_source)
            changed_want = [np.any(want[0][c] != cur_counts[c], axis=(1, 2)) for c in range(C)]   # state.py:349-350
            outs = []
            for form, dst in ((0, cand), (1, spare)):
                eng.set_option(step_form=form)
                glh, mix, changed = eng.step(cur, dst, **kw)
                _check_candidate(eng, dst, glh, mix, want, (tag, "step", i_step, "form", form))
                assert np.array_equal(changed, np.concatenate(changed_want)), (tag, "changed flags", form)
                outs.append((glh, mix))
            assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1], (tag, "forms differ")
            # the current slot is untouched by the proposals
            assert eng.mixture_loglik(cur) == (got0 if i_step == 0 else cur_mix), (tag, "current slot changed")
            if i_step != 1:                                   # accept
                cur, cand = cand, cur
                cur_groups, cur_source, cur_weights, cur_mix = new_groups, new_source, new_weights, outs[0][1]
            elif i_step == 1:
                cur_mix = eng.mixture_loglik(cur)
        eng.set_option(step_form=0)

        # ---- sbe_gibbs_step with the caller's uniforms: the oracle's draw, number for number -------------------
        n_sub = max(1, min(N, n_rows if n_rows else 20))
        objects = np.sort(rng.choice(N, size=n_sub, replace=False)).astype(np.int32)
        z = rng.random((n_sub, F))
        cur_counts = orc.recalculate_feature_counts(feats, cur_groups, cur_source)
        new_source, want_q, want_qb, new_counts = orc.gibbs_source_propose(
            feats, na, cur_groups, cur_counts, conc, cur_weights, cur_source, objects, z)
        try:
            lq, lqb, glh, mix, changed = eng.gibbs_step(cur, cand, objects, z=z)
        except EngineError as exc:                            # payload limit of the one-call form
            assert "too large" in str(exc), exc
            return
        drawn = eng.get_source_rows(cand, np.arange(N, dtype=np.int32))
        assert np.array_equal(drawn, new_source), (tag, "gibbs source rows")
        want = _expected(feats, na, cur_groups, new_source, conc, cur_weights)
        for c in range(C):
            assert np.array_equal(want[0][c], new_counts[c])
        _check_candidate(eng, cand, glh, mix, want, (tag, "gibbs"))
        assert abs(lq - want_q) <= 3e-6 * abs(want_q) + 1e-12, (tag, lq, want_q)
        assert abs(lqb - want_qb) <= 3e-6 * abs(want_qb) + 1e-12, (tag, lqb, want_qb)
        changed_want = np.concatenate([np.any(new_counts[c] != cur_counts[c], axis=(1, 2)) for c in range(C)])
        assert np.array_equal(changed, changed_want), (tag, "gibbs changed flags")


@pytest.mark.parametrize("name", ["headline", "stress"])
def test_one_call_steps_at_baseline_workloads(name):
    """sbe_step / sbe_gibbs_step on the BASELINE.json workloads themselves (configs[2] and configs[4])."""
    wl = make_workload(name)
    feats, na = wl.features, wl.na_values
    N, F, S = wl.shape
    C = wl.n_components
    n_groups = [g.shape[0] for g in wl.groups]
    rng = np.random.default_rng(77)
    with Engine(feats, n_groups, n_slots=2) as eng:
        for c in range(C):
            eng.set_concentration(c, wl.concentration[c])
        eng.load_state(0, wl.groups, wl.weights, source=wl.source)
        for c in range(C):
            eng.update_probs(0, c)
        eng.mixture_loglik(0)
        clusters, new_groups, objs, rows, new_source = _propose(rng, feats, na, wl.groups, wl.source, wl.weights, 20, True)
        want = _expected(feats, na, new_groups, new_source, wl.concentration, wl.weights)
        for form in (0, 1):
            eng.set_option(step_form=form)
            glh, mix, _ = eng.step(0, 1, clusters=clusters, changed_objects=objs, source_rows=rows)
            _check_candidate(eng, 1, glh, mix, want, (name, "step form", form))
        eng.set_option(step_form=0)
        objects = np.sort(rng.choice(N, size=20, replace=False)).astype(np.int32)
        z = rng.random((20, F))
        counts0 = orc.recalculate_feature_counts(feats, wl.groups, wl.source)
        new_source, want_q, want_qb, new_counts = orc.gibbs_source_propose(
            feats, na, wl.groups, counts0, wl.concentration, wl.weights, wl.source, objects, z)
        lq, lqb, glh, mix, _ = eng.gibbs_step(0, 1, objects, z=z)
        assert np.array_equal(eng.get_source_rows(1, objects), new_source[objects])
        want = _expected(feats, na, wl.groups, new_source, wl.concentration, wl.weights)
        _check_candidate(eng, 1, glh, mix, want, (name, "gibbs"))
        assert abs(lq - want_q) <= 3e-6 * abs(want_q) and abs(lqb - want_qb) <= 3e-6 * abs(want_qb)
