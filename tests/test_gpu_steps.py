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


# ---- batched multi-chain step (sbe_step_batch) -----------------------------------------------------------------
def _batch_equals_single_steps(workload, B, n_sweeps, seed=8, distinct_states=None, expect_kernel=None):
    """B chains stepped by ONE sbe_step_batch call against B sbe_step calls and, at the end, the oracle."""
    wl = make_workload(workload)
    feats, na = wl.features, wl.na_values
    N, F, S = wl.shape
    C = wl.n_components
    rng = np.random.default_rng(seed)
    distinct_states = B if distinct_states is None else distinct_states
    n_groups = [g.shape[0] for g in wl.groups]
    with Engine(feats, n_groups, n_slots=2 * B) as eb, Engine(feats, n_groups, n_slots=2) as es:
        for eng in (eb, es):
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
        states = []
        for i in range(B):
            clusters, weights, source = (wl.clusters, wl.weights, wl.source) if i % distinct_states == 0 else \
                (states[i % distinct_states] if i >= distinct_states else
                 make_state(feats, wl.groups[1:], wl.clusters.shape[0], seed=300 + i))
            eb.load_state(2 * i, [clusters] + wl.groups[1:], weights, source=source)
            for c in range(C):
                eb.update_probs(2 * i, c)
            states.append((clusters, weights, source))
        cur = np.arange(0, 2 * B, 2, dtype=np.int32)
        cand = cur + 1
        for sweep in range(n_sweeps):
            cl = np.stack([s[0] for s in states]).copy()
            cm = np.zeros(B, dtype=bool)
            wts = np.zeros((B, F, C), dtype=np.float32)
            wm = np.zeros(B, dtype=bool)
            ptr, objs_all, rows_all, new_states = [0], [], [], []
            for i in range(B):
                clusters, weights, source = states[i]
                kind = (i + sweep) % 4
                new_clusters, new_source, new_weights = clusters, source, weights
                objs = np.zeros(0, dtype=np.int32)
                rows = np.zeros((0, F, C), dtype=bool)
                if kind in (0, 1):
                    new_clusters, new_groups, objs, rows, new_source = _propose(
                        rng, feats, na, [clusters] + wl.groups[1:], source, weights, 12 if kind == 0 else 0, True)
                    cl[i], cm[i] = new_clusters, True
                elif kind == 2:
                    _c2, new_groups, objs, rows, new_source = _propose(rng, feats, na, [clusters] + wl.groups[1:], source,
                                                                      weights, 25, False)
                if kind in (1, 3) and i % 2 == 0:
                    new_weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
                    wts[i], wm[i] = new_weights, True
                objs_all.append(np.asarray(objs, dtype=np.int32))
                rows_all.append(rows if rows is not None else np.zeros((0, F, C), dtype=bool))
                ptr.append(ptr[-1] + len(objs))
                new_states.append((new_clusters, new_weights, new_source))
            glh, mix, changed = eb.step_batch(cur, cand, cl, cm, np.array(ptr, dtype=np.int32), np.concatenate(objs_all),
                                              np.concatenate(rows_all), wts, wm)
            if expect_kernel is not None:
                assert expect_kernel in eb.last_mixture_kernel(), eb.last_mixture_kernel()
            for i in range(B):
                clusters, weights, source = states[i]
                es.load_state(0, [clusters] + wl.groups[1:], weights, source=source)
                for c in range(C):
                    es.update_probs(0, c)
                es.mixture_loglik(0)
                kw = {}
                if cm[i]:
                    kw["clusters"] = cl[i]
                if len(objs_all[i]):
                    kw.update(changed_objects=objs_all[i], source_rows=rows_all[i])
                if wm[i]:
                    kw["weights"] = wts[i]
                g1, m1, c1 = es.step(0, 1, **kw)
                assert np.array_equal(glh[i], g1) and np.array_equal(changed[i], c1), (sweep, i)
                assert abs(mix[i] - m1) <= 1e-13 * abs(m1), (sweep, i, mix[i], m1)
                for c in range(C):
                    assert np.array_equal(eb.get_counts(int(cand[i]), c), es.get_counts(1, c))
                    assert np.array_equal(eb.get_probs(int(cand[i]), c), es.get_probs(1, c))
            accept = rng.random(B) < 0.7
            cur, cand = np.where(accept, cand, cur).astype(np.int32), np.where(accept, cur, cand).astype(np.int32)
            states = [new_states[i] if accept[i] else states[i] for i in range(B)]
        # the resident states after the sweeps equal a from-scratch evaluation by the oracle
        for i in (0, B - 1):
            clusters, weights, source = states[i]
            want = _expected(feats, na, [clusters] + wl.groups[1:], source, wl.concentration, weights)
            got = eb.mixture_loglik(int(cur[i]))
            assert abs(got - want[3]) <= 1e-10 * abs(want[3])
            for c in range(C):
                assert np.array_equal(eb.get_counts(int(cur[i]), c), want[0][c])


@pytest.mark.parametrize("parts", [None, "2", "5"])
def test_step_batch_equals_single_steps(parts, monkeypatch):
    """B chains stepped by ONE sbe_step_batch call give what B sbe_step calls give: counts, tables, per-group collapsed
    values and changed-group flags bit for bit, the mixture scalar to rounding (its block geometry depends on the
    launch's batch size); mixed deltas -- cluster moves, source rows, weights, nothing -- accepted and rejected.
    `parts`: the pipelined form large batches take (from 128 chains on: the host halves of part k+1 under the device work
    of part k), forced here on 12 chains through SBE_STEP_PARTS."""
    if parts is not None:
        monkeypatch.setenv("SBE_STEP_PARTS", parts)
    _batch_equals_single_steps("headline", 12, 4)


def test_step_batch_through_the_matrix_pipe_kernel(monkeypatch):
    """The batched step's mixture launch takes its slots from a LIST (the candidates of the chains, interleaved with their
    current slots); with the matrix-pipe form chosen for it (SBE_MFMA_MIN_BATCH lowered: 24 chains) every chain's scalar equals
    the single step's to rounding, accepted and rejected proposals alike."""
    monkeypatch.setenv("SBE_MFMA_MIN_BATCH", "8")
    _batch_equals_single_steps("headline", 24, 3, seed=21, expect_kernel="k_mixture_tuple_mfma")


def test_step_batch_natural_two_part_pipeline_160_chains(monkeypatch):
    """the NATURAL two-part pipeline (>= 128 chains, include/sbe_engine.h) and the 16-chain payload
    chunks -- 160 chains without SBE_STEP_PARTS: ten full chunks over two parts -- against single sbe_step calls."""
    monkeypatch.delenv("SBE_STEP_PARTS", raising=False)
    _batch_equals_single_steps("headline", 160, 2, seed=18, distinct_states=8)


def test_step_batch_stress_workload_8_chains(monkeypatch):
    """the batch step at the stress shape (5000 x 500 x 20, C = 4: the general rows kernel, several
    feature tiles and object chunks per chain) with 8 chains, against single sbe_step calls and the oracle."""
    monkeypatch.delenv("SBE_STEP_PARTS", raising=False)
    _batch_equals_single_steps("stress", 8, 2, seed=28, distinct_states=3)


@pytest.mark.parametrize("form", ["matrix", "delta"])
@pytest.mark.parametrize("name,n_chains", [("south_america", 64), ("headline", 16)])
def test_step_batch_replays_interleaved_reference_traces(name, n_chains, form):
    """(form = delta: the same sweeps through ResidentChainBatch.step_delta / sbe_step_batch_delta -- the moved objects
    and their new clusters instead of cluster matrices; the candidates are patched in O(delta).)
    n_chains copies of the recorded reference MCMC trace, chain i lagging i steps behind chain i-1, stepped together
    by sbe_step_batch through ResidentChainBatch: every chain reproduces the reference's recorded collapsed and mixture
    log-likelihood at every step (chains that have not started or have finished pass an empty delta)."""
    from sbayes_amd import model as sbm
    from sbayes_amd.resident import ResidentChainBatch
    from tests.test_gpu_dropin import build, load_case
    fx, tr = load_case(name)
    model, sample = build(fx)
    n_steps = min(tr.n_steps, 120)
    F, C = fx.features.shape[1], fx.n_comp
    batch = ResidentChainBatch(model, [sample] * n_chains)
    try:
        deltas = []                                                # per trace step: (clusters or None, (objs, rows), weights or None)
        prev_c, prev_w = fx.groups[0], fx.weights
        for t in range(n_steps):
            c, w = tr.clusters(t), tr.weights[t]
            objs, rows = tr.source_delta(t)
            deltas.append((c if not np.array_equal(c, prev_c) else None, (objs, rows),
                           w if not np.array_equal(w, prev_w) else None))
            prev_c, prev_w = c, w
        for sweep in range(n_steps + n_chains - 1):
            ts = [sweep - i for i in range(n_chains)]
            live = [0 <= t < n_steps for t in ts]
            if form == "matrix":
                ll, glh, mix = batch.step(clusters=[deltas[t][0] if ok else None for t, ok in zip(ts, live)],
                                          source_rows=[deltas[t][1] if ok else None for t, ok in zip(ts, live)],
                                          weights=[deltas[t][2] if ok else None for t, ok in zip(ts, live)])
            else:
                mptr, mobj, mcl, ptr, objs_l, rows_l = [0], [], [], [0], [], []
                w = np.zeros((n_chains, F, C), dtype=np.float32)
                wm = np.zeros(n_chains, dtype=bool)
                for i, (t, ok) in enumerate(zip(ts, live)):
                    moved = np.zeros(0, dtype=np.int64)
                    if ok and deltas[t][0] is not None:
                        prev = tr.clusters(t - 1) if t > 0 else fx.groups[0]
                        moved = np.flatnonzero((deltas[t][0] != prev).any(axis=0))
                        mcl.append(np.where(deltas[t][0][:, moved].any(axis=0), deltas[t][0][:, moved].argmax(axis=0), -1))
                    else:
                        mcl.append(np.zeros(0, dtype=np.int64))
                    mobj.append(moved)
                    mptr.append(mptr[-1] + moved.size)
                    objs, rows = deltas[t][1] if ok else (np.zeros(0, dtype=np.int32), np.zeros((0, F, C), dtype=bool))
                    objs_l.append(np.asarray(objs, dtype=np.int32))
                    rows_l.append(np.asarray(rows, dtype=bool).reshape(-1, F, C))
                    ptr.append(ptr[-1] + len(objs))
                    if ok and deltas[t][2] is not None:
                        w[i], wm[i] = deltas[t][2], True
                ll, glh, mix = batch.step_delta(np.array(mptr), np.concatenate(mobj), np.concatenate(mcl), np.array(ptr),
                                                np.concatenate(objs_l), np.concatenate(rows_l), w, wm)
            for i, (t, ok) in enumerate(zip(ts, live)):
                if ok:
                    assert abs(ll[i] - tr.last_lh[t]) <= 1e-6 * abs(tr.last_lh[t]), (sweep, i, t)
                    np.testing.assert_allclose(glh[i], tr.group_lh[t], rtol=1e-6, atol=1e-6)
                    assert abs(mix[i] - tr.mixture_ll[t]) <= 1e-10 * abs(tr.mixture_ll[t]), (sweep, i, t)
            batch.accept(np.array(live))
        final_counts = orc.recalculate_feature_counts(fx.features, [tr.clusters(n_steps - 1)] + fx.groups[1:], tr.source(n_steps - 1))
        for c in range(fx.n_comp):
            assert np.array_equal(batch.counts(0, c), final_counts[c])
            assert np.array_equal(batch.counts(n_chains - 1, c), final_counts[c])
    finally:
        batch.close()
        from sbayes_amd.registry import release_all
        release_all()


@pytest.mark.parametrize("name", ["headline", "south_america_like", "many_tuples"])
def test_incremental_pattern_and_tuple_update_equals_full_derivation(name):
    """Round 3: a step that moves a few objects between clusters updates the candidate's has_components pattern ids and
    group-tuple tables for those objects only (sbe_engine.hip: update_patterns_and_tuples); SBE_OPT_STEP_DERIVE = 1
    re-derives them from all N objects as before.  Forty chained steps (accept / reject mixed; moves that empty a tuple,
    create one, empty a CLUSTER -- the pattern set changes and the full derivation takes over) on two engines, one per
    mode: counts, tables, flags and per-group values bit for bit, the mixture scalar to rounding (the tuple numbering
    may differ, the looked-up values may not), and at the end the oracle."""
    if name == "headline":
        wl = make_workload("headline")
    elif name == "south_america_like":                     # three components, small clusters that do get emptied
        wl = make_workload("sa_like", shape=(120, 70, 5, 3, (6,), True))
    else:                                                  # dozens of tuples: indices are vacated and reused
        wl = make_workload("tuples", shape=(400, 64, 4, 6, (3, 2), False))
    feats, na = wl.features, wl.na_values
    N, F, S = wl.shape
    C = wl.n_components
    K = wl.clusters.shape[0]
    rng = np.random.default_rng(77)
    n_groups = [g.shape[0] for g in wl.groups]
    with Engine(feats, n_groups, n_slots=2) as ea, Engine(feats, n_groups, n_slots=2) as eb:
        eb.set_option(step_derive=1)
        for eng in (ea, eb):
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
            eng.load_state(0, wl.groups, wl.weights, source=wl.source)
            for c in range(C):
                eng.update_probs(0, c)
            eng.mixture_loglik(0)
        cur, cand = 0, 1
        groups, source, weights = list(wl.groups), wl.source, wl.weights
        for i_step in range(40):
            if i_step % 10 == 7:                           # empty a whole cluster: a has_components pattern may vanish
                clusters = groups[0].copy()
                clusters[int(rng.integers(0, K))] = False
                objs = np.zeros(0, dtype=np.int32)
                rows, new_source = None, source
            else:
                clusters, _g, objs, rows, new_source = _propose(rng, feats, na, groups, source, weights,
                                                               int(rng.integers(0, 12)), True)
            kw = {"clusters": clusters}
            if len(objs):
                kw.update(changed_objects=objs, source_rows=rows)
            ga, ma, ca = ea.step(cur, cand, **kw)
            gb, mb, cb = eb.step(cur, cand, **kw)
            assert np.array_equal(ga, gb) and np.array_equal(ca, cb), i_step
            assert abs(ma - mb) <= 1e-13 * abs(mb), (i_step, ma, mb)
            for c in range(C):
                assert np.array_equal(ea.get_counts(cand, c), eb.get_counts(cand, c))
                assert np.array_equal(ea.get_probs(cand, c), eb.get_probs(cand, c))
            if rng.random() < 0.7:
                cur, cand = cand, cur
                groups, source = [clusters] + groups[1:], new_source
        want = _expected(feats, na, groups, source, wl.concentration, weights)
        for eng in (ea, eb):
            got = eng.mixture_loglik(cur)
            assert abs(got - want[3]) <= 1e-10 * abs(want[3])


def test_step_batch_reports_the_malformed_chain_by_index():
    """every chain of a batch has its own data-check words.  One chain's proposal carries a source row with
    two components set: the error names THAT chain, and the other chains' results were delivered all the same."""
    wl = make_workload("cfg1")
    feats = wl.features
    N, F, S = wl.shape
    C = wl.n_components
    B = 5
    n_groups = [g.shape[0] for g in wl.groups]
    with Engine(feats, n_groups, n_slots=2 * B) as eb, Engine(feats, n_groups, n_slots=2) as es:
        for eng in (eb, es):
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
        for i in range(B):
            eb.load_state(2 * i, wl.groups, wl.weights, source=wl.source)
            for c in range(C):
                eb.update_probs(2 * i, c)
        es.load_state(0, wl.groups, wl.weights, source=wl.source)
        for c in range(C):
            es.update_probs(0, c)
        es.mixture_loglik(0)
        objs = np.array([1, 4], dtype=np.int32)
        good = wl.source[objs].copy()
        bad = good.copy()
        bad[0, 0, :] = True                                              # two components set in one observation
        rows = np.concatenate([bad if i == 3 else good for i in range(B)])
        ptr = np.arange(0, 2 * B + 1, 2, dtype=np.int32)
        cur = np.arange(0, 2 * B, 2, dtype=np.int32)
        with pytest.raises(EngineError, match="chain 3: source is not one-hot") as exc:
            eb.step_batch(cur, cur + 1, None, None, ptr, np.tile(objs, B), rows)
        assert "chain 3" in str(exc.value)
        # the batch still works afterwards, and a clean sweep equals the single step
        glh, mix, changed = eb.step_batch(cur, cur + 1, None, None, ptr, np.tile(objs, B), np.concatenate([good] * B))
        g1, m1, c1 = es.step(0, 1, changed_objects=objs, source_rows=good)
        for i in range(B):
            assert np.array_equal(glh[i], g1) and abs(mix[i] - m1) <= 1e-13 * abs(m1)


@pytest.mark.parametrize("name,B", [("headline", 12), ("many_tuples", 6), ("stress_like", 4)])
def test_step_batch_delta_equals_matrix_form(name, B):
    """sbe_step_batch_delta (proposals as moved objects + changed rows; candidates PATCHED in O(delta)) against
    sbe_step_batch (full cluster matrices; candidates rebuilt): eight sweeps of mixed proposals -- cluster moves, moves out
    of every cluster, source rows, weights, nothing, a whole cluster emptied (the pattern set changes: that chain falls
    back inside the call) -- with random accepts.  Counts, tables, per-group values and flags bit for bit, the mixture
    scalar to rounding; at the end the oracle.  The first sweep of the delta engine is the fallback (no records yet)."""
    if name == "headline":
        wl = make_workload("headline")
    elif name == "many_tuples":
        wl = make_workload("tuples", shape=(400, 64, 4, 6, (3, 2), False))
    else:                                                  # C = 4, hundreds of tuples: the rows kernel evaluates the candidates
        wl = make_workload("stress_like", shape=(900, 150, 12, 8, (15, 15), False))
    feats, na = wl.features, wl.na_values
    N, F, S = wl.shape
    C = wl.n_components
    K = wl.clusters.shape[0]
    rng = np.random.default_rng(91)
    n_groups = [g.shape[0] for g in wl.groups]
    with Engine(feats, n_groups, n_slots=2 * B) as ea, Engine(feats, n_groups, n_slots=2 * B) as eb:
        states = []
        for i in range(B):
            clusters, weights, source = (wl.clusters, wl.weights, wl.source) if i == 0 else \
                make_state(feats, wl.groups[1:], K, seed=500 + i)
            states.append((clusters, weights, source))
        for eng in (ea, eb):
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
            for i, (clusters, weights, source) in enumerate(states):
                eng.load_state(2 * i, [clusters] + wl.groups[1:], weights, source=source)
                for c in range(C):
                    eng.update_probs(2 * i, c)
        cur = np.arange(0, 2 * B, 2, dtype=np.int32)
        cand = cur + 1
        for sweep in range(8):
            cl = np.stack([s[0] for s in states]).copy()
            wts = np.zeros((B, F, C), dtype=np.float32)
            wm = np.zeros(B, dtype=bool)
            mptr, mobj, mcl, ptr, objs_all, rows_all, new_states = [0], [], [], [0], [], [], []
            for i in range(B):
                clusters, weights, source = states[i]
                kind = (i + sweep) % 5
                new_clusters, new_source, new_weights = clusters, source, weights
                objs, rows = np.zeros(0, dtype=np.int32), np.zeros((0, F, C), dtype=bool)
                if kind in (0, 1):
                    new_clusters, _g, objs, rows, new_source = _propose(rng, feats, na, [clusters] + wl.groups[1:], source, weights,
                                                                       12 if kind == 0 else 0, True)
                elif kind == 2:
                    _c2, _g, objs, rows, new_source = _propose(rng, feats, na, [clusters] + wl.groups[1:], source, weights, 25, False)
                elif kind == 3 and sweep >= 2:                          # a whole cluster emptied
                    new_clusters = clusters.copy()
                    new_clusters[int(rng.integers(0, K))] = False
                if kind in (1, 4) and i % 2 == 0:
                    new_weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
                    wts[i], wm[i] = new_weights, True
                cl[i] = new_clusters
                moved = np.flatnonzero((new_clusters != clusters).any(axis=0))
                if sweep % 2 == 1 and moved.size < N:                   # sometimes also list an object that does not move
                    moved = np.union1d(moved, [int(rng.integers(0, N))])
                mobj.append(moved.astype(np.int32))
                mcl.append(np.where(new_clusters[:, moved].any(axis=0), new_clusters[:, moved].argmax(axis=0), -1).astype(np.int32))
                mptr.append(mptr[-1] + moved.size)
                objs_all.append(np.asarray(objs, dtype=np.int32))
                rows_all.append(rows if rows is not None and len(objs) else np.zeros((0, F, C), dtype=bool))
                ptr.append(ptr[-1] + len(objs))
                new_states.append((new_clusters, new_weights, new_source))
            ptr_a = np.array(ptr, dtype=np.int32)
            ga, ma, ca = ea.step_batch(cur, cand, cl, None, ptr_a, np.concatenate(objs_all), np.concatenate(rows_all), wts, wm)
            gb, mb, cb = eb.step_batch_delta(cur, cand, np.array(mptr, dtype=np.int32), np.concatenate(mobj), np.concatenate(mcl),
                                             ptr_a, np.concatenate(objs_all), np.concatenate(rows_all), wts, wm)
            assert np.array_equal(ga, gb) and np.array_equal(ca, cb), (name, sweep)
            assert np.all(np.abs(ma - mb) <= 1e-13 * np.abs(ma)), (name, sweep, ma, mb)
            for i in (0, B // 2, B - 1):
                for c in range(C):
                    assert np.array_equal(ea.get_counts(int(cand[i]), c), eb.get_counts(int(cand[i]), c)), (sweep, i, c)
                    assert np.array_equal(ea.get_probs(int(cand[i]), c), eb.get_probs(int(cand[i]), c)), (sweep, i, c)
            accept = rng.random(B) < 0.6
            cur, cand = np.where(accept, cand, cur).astype(np.int32), np.where(accept, cur, cand).astype(np.int32)
            states = [new_states[i] if accept[i] else states[i] for i in range(B)]
        for i in (0, B - 1):
            clusters, weights, source = states[i]
            want = _expected(feats, na, [clusters] + wl.groups[1:], source, wl.concentration, weights)
            for eng in (ea, eb):
                got = eng.mixture_loglik(int(cur[i]))
                assert abs(got - want[3]) <= 1e-10 * abs(want[3])
                for c in range(C):
                    assert np.array_equal(eng.get_counts(int(cur[i]), c), want[0][c])


@pytest.mark.parametrize("name", ["headline", "many_tuples"])
def test_single_step_delta_equals_matrix_form(name):
    """sbe_step_delta (ResidentChain-level: moved objects + changed rows, candidate patched in O(delta)) against sbe_step
    (cluster matrix): thirty chained steps with accepts and rejects on two engines; counts, tables, per-group values
    and flags bit for bit, the mixture scalar to rounding, the oracle at the end."""
    wl = make_workload("headline") if name == "headline" else make_workload("tuples", shape=(400, 64, 4, 6, (3, 2), False))
    feats, na = wl.features, wl.na_values
    N, F, S = wl.shape
    C = wl.n_components
    rng = np.random.default_rng(17)
    n_groups = [g.shape[0] for g in wl.groups]
    with Engine(feats, n_groups, n_slots=2) as ea, Engine(feats, n_groups, n_slots=2) as eb:
        for eng in (ea, eb):
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
            eng.load_state(0, wl.groups, wl.weights, source=wl.source)
            for c in range(C):
                eng.update_probs(0, c)
            eng.mixture_loglik(0)
        cur, cand = 0, 1
        groups, source, weights = list(wl.groups), wl.source, wl.weights
        for i_step in range(30):
            clusters, _g, objs, rows, new_source = _propose(rng, feats, na, groups, source, weights, int(rng.integers(0, 15)),
                                                           i_step % 3 != 2)
            new_weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32) if i_step % 7 == 3 else None
            kw = {"clusters": clusters}
            if len(objs):
                kw.update(changed_objects=objs, source_rows=rows)
            if new_weights is not None:
                kw["weights"] = new_weights
            ga, ma, ca = ea.step(cur, cand, **kw)
            moved = np.flatnonzero((clusters != groups[0]).any(axis=0))
            mcl = np.where(clusters[:, moved].any(axis=0), clusters[:, moved].argmax(axis=0), -1)
            gb, mb, cb = eb.step_delta(cur, cand, moved, mcl, objs if len(objs) else None, rows if len(objs) else None, new_weights)
            assert np.array_equal(ga, gb) and np.array_equal(ca, cb), i_step
            assert abs(ma - mb) <= 1e-13 * abs(ma), (i_step, ma, mb)
            for c in range(C):
                assert np.array_equal(ea.get_counts(cand, c), eb.get_counts(cand, c))
                assert np.array_equal(ea.get_probs(cand, c), eb.get_probs(cand, c))
            if rng.random() < 0.6:
                cur, cand = cand, cur
                groups, source = [clusters] + groups[1:], new_source
                weights = new_weights if new_weights is not None else weights
        want = _expected(feats, na, groups, source, wl.concentration, weights)
        for eng in (ea, eb):
            got = eng.mixture_loglik(cur)
            assert abs(got - want[3]) <= 1e-10 * abs(want[3])
