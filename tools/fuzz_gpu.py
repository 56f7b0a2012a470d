#!/usr/bin/env python3
"""Randomised differential test on the GPU box: random shapes / states / batch sizes through every fused-kernel form,
both sbe_step forms and the one-call Gibbs step, against the CPU oracle.  Not part of the pytest suite (open-ended
run time):  python tools/fuzz_gpu.py --seconds 300 [--seed 0]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import sbayes_oracle as orc                                       # noqa: E402  (checker only)
from sbayes_amd.engine import (MIXTURE_ONEHOT, MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED, MIXTURE_PACKED_GENERAL,   # noqa: E402
                               MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA, MIXTURE_PACKED_V2, Engine, EngineError)
from tests.test_gpu_shapes import random_case                                  # noqa: E402


def random_state(rng, feats, groups0, n_groups):
    N, F, _ = feats.shape
    C = len(n_groups)
    if C == 1:
        groups = groups0
    else:
        a = rng.integers(0, n_groups[0] + rng.integers(0, n_groups[0] + 1), size=N)
        groups = [np.stack([a == k for k in range(n_groups[0])])] + groups0[1:]
    weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
    hc = orc.has_components(groups)
    covered = hc.any(axis=1)
    src_idx = np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)
    source = np.eye(C, dtype=bool)[src_idx]
    source[~feats.any(-1)] = False
    source[~covered] = False
    return groups, weights, source, covered


def forms_case(rng, eng, feats, n_groups, conc, state, tag, stats):
    """collapsed_loglik_all / source_prior / source_lh_by_feature / source_posterior / normalize_weights / counts_delta /
    given_unchanged_lh / cluster_posterior_marginals / jump_lh_resident of the slot-0 state at random subsets."""
    from tests._fake_engine import FakeEngine
    groups, weights, source, counts = state
    N, F, C = source.shape
    K = n_groups[0]
    fake = FakeEngine(feats, n_groups)
    unif = feats.any(axis=0).astype(np.float64)                     # (the cluster prior's uniform concentration: 1 on seen states)
    unif[~unif.any(axis=1), 0] = 1.0
    for c in range(C):
        fake.set_concentration(c, conc[c]); fake.set_groups(0, c, groups[c]); fake.set_counts(0, c, counts[c])
    fake.set_source(0, source); fake.set_weights(0, weights); fake.set_uniform_counts(unif)
    eng.set_uniform_counts(unif)
    np.testing.assert_allclose(eng.collapsed_loglik_all(0), fake.collapsed_loglik_all(0), rtol=2e-6, atol=1e-5, err_msg=tag)
    with np.errstate(divide="ignore"):
        np.testing.assert_allclose(eng.source_prior(0), fake.source_prior(0), rtol=2e-6, atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(eng.source_lh_by_feature(0), fake.source_lh_by_feature(0), rtol=max(3e-5, N * 2.0 ** -25), atol=1e-4, err_msg=tag)
    hc = rng.random((int(rng.integers(1, 40)), C)) < 0.6
    hc[:, 1] = True
    assert np.array_equal(eng.normalize_weights(weights, hc), fake.normalize_weights(weights, hc)), (tag, "normalize_weights")
    objs = np.unique(rng.integers(0, N, size=int(rng.integers(1, min(N, 60) + 1)))).astype(np.int32)
    assert np.array_equal(eng.source_posterior(0, objs), fake.source_posterior(0, objs)), (tag, "source_posterior")
    i_cl = int(rng.integers(0, K))
    with np.errstate(divide="ignore", invalid="ignore"):
        want = fake.given_unchanged_lh(0, i_cl, objs)
    assert np.array_equal(eng.given_unchanged_lh(0, i_cl, objs), want), (tag, "given_unchanged_lh")
    # the literal a3 / a1 surfaces (streamed by the kernel from 512 KB on, chunk after chunk): bit for bit, strided view, partial update
    lh_want = orc.likelihood_per_component(feats, ~feats.any(-1), groups, counts, conc)
    assert np.array_equal(eng.likelihood_per_component(0), lh_want), (tag, "likelihood_per_component")
    probs0 = eng.get_probs(0, 0)
    changed = np.flatnonzero(rng.random(K) < 0.6).astype(np.int64)
    view = np.full((N, F, 3), -5.0)
    want_v = view.copy()
    eng.component_lh(probs0, groups[0], changed, view[..., 1])
    orc.compute_component_likelihood(feats, probs0, groups[0], changed, want_v[..., 1])
    assert np.array_equal(view, want_v), (tag, "component_lh")
    stats["surfaces"] = stats.get("surfaces", 0) + 1
    # ClusterOperator.gibbs_sample_source in one call (round 4): drawn components, p[drawn], p_back[old source] -- bit for bit at T = 1
    hc_new = np.stack([g[:, objs].any(axis=0) for g in groups], axis=1)
    hc_old = hc_new.copy()
    flip = rng.random(objs.size) < 0.5
    hc_old[flip, 0] = ~hc_old[flip, 0]
    if C > 1:
        hc_old[:, 1] = hc_new[:, 1] = True               # (an object in no component at all has no weight row)
    so_g = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
    zz = rng.random((objs.size, F))
    if (hc_new.any(axis=1) & hc_old.any(axis=1)).all():
        from_prior = bool(rng.integers(0, 2))
        with np.errstate(divide="ignore", invalid="ignore"):
            want_g = fake.given_unchanged_gibbs(0, i_cl, objs, hc_new, hc_old, so_g, zz, from_prior=from_prior)
        degenerate = not from_prior and not np.isfinite(want_g[1]).all()   # (a posterior row with zero mass: the reference asserts)
        if not degenerate:
            got_g = eng.given_unchanged_gibbs(0, i_cl, objs, hc_new, hc_old, so_g, zz, from_prior=from_prior)
            for a_, b_, what in zip(got_g, want_g, ("ids", "p[drawn]", "p_back[old]")):
                ok_ = np.isfinite(b_) if b_.dtype != np.uint8 else np.ones(b_.shape, dtype=bool)
                assert np.array_equal(a_[ok_], b_[ok_]), (tag, "given_unchanged_gibbs", what)
            stats["cluster_gibbs"] = stats.get("cluster_gibbs", 0) + 1
            # ... the same call with its count delta, plain and with the slot FOLLOWING (sbe_given_unchanged_gibbs_apply): the same
            # five arrays; the following slot ends as the explicit patches of the same data leave it
            gid_n = np.stack([np.where(g[:, objs].any(axis=0), g[:, objs].argmax(axis=0) + eng.group_offsets[c_], -1).astype(np.int32)
                              for c_, g in enumerate(groups)])
            gid_o = gid_n.copy()
            gid_o[0] = np.where(hc_old[:, 0], np.maximum(gid_n[0], 0), -1)
            with_c = eng.given_unchanged_gibbs(0, i_cl, objs, hc_new, hc_old, so_g, zz, from_prior=from_prior, gid_old=gid_o, gid_new=gid_n)
            sa_, sb_ = eng.n_slots - 1, eng.n_slots - 2
            eng.copy_slot(sa_, 0); eng.copy_slot(sb_, 0)
            eng.update_probs(sa_, range(C)); eng.update_probs(sb_, range(C))
            off_ = eng.group_offsets
            comp_ = np.searchsorted(off_, with_c[3], side="right") - 1
            rows_n = (np.stack([counts[c_][g_ - off_[c_]] for g_, c_ in zip(with_c[3], comp_)]) + with_c[4]) if with_c[3].size else None
            if rows_n is None or (rows_n >= 0).all():
                fol = eng.given_unchanged_gibbs(sa_, i_cl, objs, hc_new, hc_old, so_g, zz, from_prior=from_prior, gid_old=gid_o, gid_new=gid_n,
                                                follow=True, update_probs=True)
                for a_, b_, what in zip(fol, with_c, ("ids", "sel", "back", "touched", "rows")):
                    assert np.array_equal(a_, b_, equal_nan=True) if a_.dtype.kind == "f" else np.array_equal(a_, b_), (tag, "cluster gibbs following", what)
                ok_rows = True
                if with_c[3].size:
                    try:
                        eng.set_counts_rows(sb_, with_c[3], rows_n, update_probs=True)
                        eng.sync()
                    except Exception as exc:
                        ok_rows = False
                        assert "normali" in str(exc).lower(), (tag, exc)
                    eng.set_source_rows(sb_, objs, with_c[0][..., None] == np.arange(C, dtype=np.uint8))
                if ok_rows:
                    for c_ in range(C):
                        assert np.array_equal(eng.get_counts(sa_, c_), eng.get_counts(sb_, c_)), (tag, "cluster gibbs: following counts", c_)
                        assert np.array_equal(eng.get_probs(sa_, c_), eng.get_probs(sb_, c_)), (tag, "cluster gibbs: following tables", c_)
                    assert np.array_equal(eng.get_source_rows(sa_, objs), eng.get_source_rows(sb_, objs)), (tag, "cluster gibbs: following source rows")
                    stats["gibbs_follow"] = stats.get("gibbs_follow", 0) + 1
                else:
                    try:
                        eng.sync()
                    except Exception:
                        pass
    with np.errstate(divide="ignore", invalid="ignore"):
        want = fake.cluster_posterior_marginals(0, i_cl, objs)
        got = eng.cluster_posterior_marginals(0, i_cl, objs)
    ok = np.isfinite(want) & (want > -690.0)      # (and above the denormal range: the product keeps its digits)
    # (the double takes the log of a linear-space product over the features, which underflows beyond F ~ 75 --
    #  SURVEY.md H5 -- where the device's sum of logs stays finite: compared where the reference value exists)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9, err_msg=tag)
    # update_feature_counts in delta form: the subset's clusters and source rows change
    off = eng.group_offsets
    new_clusters = groups[0].copy()
    new_clusters[:, objs] = False
    move = rng.integers(0, K + 1, size=objs.size)
    new_clusters[move[move < K], objs[move < K]] = True
    pick = rng.integers(0, C + 1, size=(objs.size, F))
    new_rows = pick[..., None] == np.arange(C)
    ids = lambda g, o: np.where(g[:, objs].any(axis=0), g[:, objs].argmax(axis=0) + o, -1).astype(np.int32)   # noqa: E731
    gid_old = np.stack([ids(groups[c], off[c]) for c in range(C)])
    gid_new = gid_old.copy()
    gid_new[0] = ids(new_clusters, 0)
    so = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
    sn = np.where(new_rows.any(-1), new_rows.argmax(-1), 255).astype(np.uint8)
    t_got, d_got = eng.counts_delta(objs, gid_old, gid_new, so, sn)
    t_want, d_want = fake.counts_delta(objs, gid_old, gid_new, so, sn)
    assert np.array_equal(t_got, t_want) and np.array_equal(d_got, d_want), (tag, "counts_delta")
    # ... and a slot holding the old counts follows the difference (sbe_counts_delta_apply) = the rows patched the explicit way
    if (counts[0] >= 0).all():
        sa_, sb_ = eng.n_slots - 1, eng.n_slots - 2
        eng.copy_slot(sa_, 0); eng.copy_slot(sb_, 0)
        eng.update_probs(sa_, range(C)); eng.update_probs(sb_, range(C))
        t_f, d_f = eng.counts_delta(objs, gid_old, gid_new, so, sn, follow_slot=sa_, update_probs=True, update_source=True)
        assert np.array_equal(eng.get_source_rows(sa_, objs), new_rows), (tag, "following source rows")
        assert np.array_equal(t_f, t_want) and np.array_equal(d_f, d_want), (tag, "counts_delta with a following slot")
        comp_of = np.searchsorted(off, t_want, side="right") - 1
        rows_f = np.stack([counts[c_][g_ - off[c_]] for g_, c_ in zip(t_want, comp_of)]) + d_want if t_want.size else np.zeros((0,) + counts[0].shape[1:], np.float32)
        ok_rows = True
        try:
            eng.set_counts_rows(sb_, t_want, rows_f, update_probs=True)
            eng.sync()
        except Exception as exc:                      # (a row that normalises to nothing: both forms must say so)
            ok_rows = False
            assert "normali" in str(exc).lower(), (tag, exc)
        if ok_rows:
            for c_ in range(C):
                assert np.array_equal(eng.get_counts(sa_, c_), eng.get_counts(sb_, c_)), (tag, "following counts", c_)
                assert np.array_equal(eng.get_probs(sa_, c_), eng.get_probs(sb_, c_)), (tag, "following probability rows", c_)
            stats["follow"] = stats.get("follow", 0) + 1
        else:
            try:
                eng.sync()
            except Exception:
                pass
    members = np.flatnonzero(groups[0][i_cl]).astype(np.int32)
    if members.size and K >= 2:
        i_tg = (i_cl + 1) % K
        with np.errstate(divide="ignore", invalid="ignore"):
            want = fake.jump_lh_resident(0, i_cl, i_tg, members)
            got = eng.jump_lh_resident(0, i_cl, i_tg, members)
        ok = np.isfinite(want) & (want > -690.0)
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-9, atol=1e-9, err_msg=tag)
    # round 4, second session: the one-launch forms against the table-kernel forms (same bits), count rows with their
    # probability rows, GibbsSampleSource._propose in one call against the double's composition of the call-by-call pieces
    fused = (eng.cluster_posterior_marginals(0, i_cl, objs), eng.given_unchanged_lh(0, i_cl, objs),
             eng.jump_lh_resident(0, i_cl, (i_cl + 1) % K, members) if members.size and K >= 2 else None)
    eng.set_option(fuse_tables=False)
    plain = (eng.cluster_posterior_marginals(0, i_cl, objs), eng.given_unchanged_lh(0, i_cl, objs),
             eng.jump_lh_resident(0, i_cl, (i_cl + 1) % K, members) if members.size and K >= 2 else None)
    eng.set_option(fuse_tables=True)
    for a_, b_, what in zip(fused, plain, ("marginals", "given_unchanged_lh", "jump")):
        assert a_ is None or np.array_equal(a_, b_), (tag, "fused != table kernel in front", what)
    s1, s2 = eng.n_slots - 1, eng.n_slots - 2
    eng.copy_slot(s1, 0)
    eng.copy_slot(s2, 0)
    c_pick = int(rng.integers(0, C))
    g_pick = np.unique(rng.integers(0, n_groups[c_pick], size=2))
    rows_new = rng.integers(0, 30, size=(g_pick.size,) + counts[c_pick].shape[1:]).astype(np.float32)
    eng.set_counts_rows(s1, off[c_pick] + g_pick, rows_new, update_probs=True)
    eng.set_counts_rows(s2, off[c_pick] + g_pick, rows_new)
    eng.update_probs(s2, c_pick)
    assert np.array_equal(eng.get_probs(s1, c_pick), eng.get_probs(s2, c_pick)), (tag, "set_counts_rows(update_probs=True)")
    assert np.array_equal(eng.get_counts(s1, c_pick), eng.get_counts(s2, c_pick)), (tag, "set_counts_rows(update_probs=True): counts")
    if eng.gibbs_propose_supported():
        from_prior = bool(rng.integers(0, 2))
        with np.errstate(divide="ignore", invalid="ignore"):
            want_p = fake.gibbs_propose(0, 1, objs, zz, from_prior=from_prior)
        if np.isfinite(want_p[1]).all() and np.isfinite(want_p[2]).all():      # (a posterior row with zero mass: the reference asserts)
            got_p = eng.gibbs_propose(0, s1, objs, zz, from_prior=from_prior)
            for a_, b_, what in zip(got_p, want_p, ("ids", "p[drawn]", "p_back[old]", "touched", "count rows")):
                assert a_.shape == b_.shape and np.array_equal(a_, b_), (tag, "gibbs_propose", what)
            stats["gibbs_propose"] = stats.get("gibbs_propose", 0) + 1
            # ... and with the current slot FOLLOWING (sbe_gibbs_propose_apply): it ends as the double's candidate
            eng.copy_slot(s2, 0)
            eng.update_probs(s2, range(C))
            got_f = eng.gibbs_propose(s2, s1, objs, zz, from_prior=from_prior, follow=True)
            for a_, b_, what in zip(got_f, want_p, ("ids", "p[drawn]", "p_back[old]", "touched", "count rows")):
                assert np.array_equal(a_, b_), (tag, "gibbs_propose following", what)
            if got_f[3].size:
                assert np.array_equal(eng.get_source_rows(s2, objs), fake.get_source_rows(1, objs)), (tag, "gibbs_propose: following source rows")
                for c_ in range(C):
                    assert np.array_equal(eng.get_counts(s2, c_), fake._slot(1)["counts"][c_]), (tag, "gibbs_propose: following counts", c_)
                assert np.array_equal(eng.likelihood_per_component(s2), fake._state(1)[2]), (tag, "gibbs_propose: following tables")
    # third session: likelihood + prior in one call = the two calls, bit for bit; a slot walked through cluster moves (ids,
    # pattern / tuple tables followed on the host, per-pattern weights: one launch per sbe_set_groups) = a slot set at once
    pg, po = eng.collapsed_and_source_prior(0)
    assert np.array_equal(pg, eng.collapsed_loglik_all(0)) and np.array_equal(po, eng.source_prior(0), equal_nan=True), (tag, "collapsed_and_source_prior")
    walk = groups[0].copy()
    eng.copy_slot(s1, 0)
    try:
        for step in range(4):
            for n in rng.choice(N, size=min(N, int(rng.choice([1, 1, 2, max(1, N // 3)]))), replace=False):
                walk[:, n] = False
                k = int(rng.integers(0, K + 1))
                if k < K:
                    walk[k, n] = True
            eng.set_groups(0, 0, walk)
            eng.set_groups(s1, 0, walk)
            if C > 1:
                eng.set_groups(s1, 1, groups[1])        # (another component's ids: s1's tables are derived from all N)
            with np.errstate(divide="ignore"):
                a_, b_ = eng.source_prior(0), eng.source_prior(s1)
            assert np.array_equal(a_, b_, equal_nan=True), (tag, "set_groups walk: source_prior", step)
            assert eng.mixture_loglik(0) == eng.mixture_loglik(s1) or (np.isnan(eng.mixture_loglik(0)) and np.isnan(eng.mixture_loglik(s1))), (tag, "set_groups walk: mixture", step)
            assert np.array_equal(eng.weights_normalized(0), eng.weights_normalized(s1), equal_nan=True), (tag, "set_groups walk: weights", step)
            with np.errstate(invalid="ignore", divide="ignore"):
                want_w = orc.normalize_weights(weights, orc.has_components([walk] + list(groups[1:])))
            assert np.array_equal(eng.weights_normalized(0), want_w, equal_nan=True), (tag, "set_groups walk: weights vs oracle", step)
    finally:
        eng.set_groups(0, 0, groups[0])
    # the row uploads of a bind in one launch (sbe_set_slot_delta) = the three setters one after the other
    eng.copy_slot(s1, 0); eng.copy_slot(s2, 0)
    eng.update_probs(s1, range(C)); eng.update_probs(s2, range(C))
    moved_cl = groups[0].copy()
    for n in rng.choice(N, size=min(N, 2), replace=False):
        moved_cl[:, n] = False
        moved_cl[int(rng.integers(0, K)), n] = True
    use = [bool(rng.integers(0, 2)) for _ in range(3)]
    if sum(use) < 2:
        use = [True, True, True]
    kw = dict(groups_component=0, groups=moved_cl if use[0] else None,
              count_idx=(off[c_pick] + g_pick) if use[1] else None, count_rows=rows_new if use[1] else None, update_probs=True,
              source_objects=objs if use[2] else None, source_rows=new_rows if use[2] else None)
    eng.set_slot_delta(s1, **kw)
    if use[0]:
        eng.set_groups(s2, 0, moved_cl)
    if use[1]:
        eng.set_counts_rows(s2, off[c_pick] + g_pick, rows_new, update_probs=True)
    if use[2]:
        eng.set_source_rows(s2, objs, new_rows)
    for c_ in range(C):
        assert np.array_equal(eng.get_counts(s1, c_), eng.get_counts(s2, c_)) and np.array_equal(eng.get_probs(s1, c_), eng.get_probs(s2, c_)), (tag, "set_slot_delta", c_)
    assert np.array_equal(eng.get_source_rows(s1, objs), eng.get_source_rows(s2, objs)), (tag, "set_slot_delta: source rows")
    assert np.array_equal(eng.weights_normalized(s1), eng.weights_normalized(s2), equal_nan=True), (tag, "set_slot_delta: weights")
    a_, b_ = eng.mixture_loglik(s1), eng.mixture_loglik(s2)
    assert a_ == b_ or (np.isnan(a_) and np.isnan(b_)), (tag, "set_slot_delta: mixture")
    stats["forms"] = stats.get("forms", 0) + 1


def one_case(rng, stats, big=False):
    if big:                                             # long chunks, several block generations, ragged tiles
        N = int(rng.integers(600, 4000))
        F = int(rng.choice([rng.integers(60, 80), rng.integers(120, 140), rng.integers(180, 270)]))
        S = int(rng.integers(2, 12))
        C = int(rng.integers(1, 4))
        n_groups = [int(rng.integers(1, 6))] + [1 if c == 1 else int(rng.integers(1, 4)) for c in range(1, C)]
        B = int(rng.choice([rng.integers(8, 64), rng.integers(64, 200), rng.integers(200, 700)]))
    else:
        N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 400), rng.integers(400, 3000)]))
        F = int(rng.choice([rng.integers(1, 20), rng.integers(20, 140), rng.integers(140, 300)]))
        S = int(rng.choice([rng.integers(1, 6), rng.integers(6, 40)]))
        C = int(rng.integers(1, 5))
        n_groups = [int(rng.integers(1, 7))] + [1 if c == 1 else int(rng.integers(1, 9)) for c in range(1, C)]
        B = int(rng.choice([1, 2, rng.integers(3, 9), rng.integers(9, 40)]))
    na_rate = float(rng.choice([0.0, 0.03, 0.3]))
    feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, na_rate)
    na = ~feats.any(-1)
    tag = f"N{N} F{F} S{S} groups{n_groups} B{B} na{na_rate}"
    with Engine(feats, n_groups, n_slots=B + 5) as eng:
        for c in range(C):
            eng.set_concentration(c, conc[c])
        want, states = [], []
        for b in range(B):
            groups, weights, source, covered = random_state(rng, feats, groups0, n_groups)
            if not covered.all() and C == 1:
                return
            eng.load_state(b, groups, weights, source=source)
            for c in range(C):
                eng.update_probs(b, c)
            counts = orc.recalculate_feature_counts(feats, groups, source)
            with np.errstate(divide="ignore", invalid="ignore"):
                w = orc.normalize_weights(weights, orc.has_components(groups))
                obs = orc.mixture_observation_lh(w, orc.likelihood_per_component(feats, na, groups, counts, conc))
                want.append(np.log(obs)[~na].sum())
            states.append((groups, weights, source, counts))
        want = np.array(want)
        if not np.all(np.isfinite(want)):
            return                                  # objects without any component: the reference asserts there
        for kernel in (MIXTURE_PACKED, MIXTURE_ONEHOT, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS,
                       MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED_TUPLE_MFMA):
            eng.set_option(kernel=kernel)
            try:
                got = eng.mixture_loglik_batch(0, B)
            except EngineError as exc:
                if "not applicable" in str(exc):
                    continue
                raise
            # 1e-10 relative (north_star), with an absolute floor of 1e-16 per observation: the product form of the log
            # accumulation rounds every factor at 2^-53 of ~1, which is all that is left when the sum itself is ~0
            # (S = 1: every probability is 1, the terms are the float32 rounding of the weights)
            tol = 1e-10 * np.abs(want) + 1e-16 * N * F
            assert np.all((np.abs(got - want) <= tol) | (got == want)), (tag, kernel, got, want)
            stats["evals"] += B
        eng.set_option(kernel=MIXTURE_PACKED)
        # one-call steps from state 0: lean vs general form, counts vs the oracle
        groups, weights, source, counts = states[0]
        if C >= 2 and N >= 2:
            clusters = groups[0].copy()
            for n in rng.integers(0, N, size=min(3, N)):
                clusters[:, n] = False
                k = int(rng.integers(0, clusters.shape[0] + 1))
                if k < clusters.shape[0]:
                    clusters[k, n] = True
            new_groups = [clusters] + groups[1:]
            hc = orc.has_components(new_groups)
            if hc.any(axis=1).all():
                objs = np.unique(rng.integers(0, N, size=min(5, N))).astype(np.int32)
                moved = np.union1d(objs, np.flatnonzero((clusters != groups[0]).any(axis=0)))
                new_source = source.copy()
                idx = np.argmax(rng.random((moved.size, F, C)) * hc[moved][:, None, :], axis=-1)
                rows = np.eye(C, dtype=bool)[idx]
                rows[na[moved]] = False
                new_source[moved] = rows
                outs = []
                for form in (0, 1):
                    eng.set_option(step_form=form)
                    outs.append(eng.step(0, B, clusters=clusters, changed_objects=moved.astype(np.int32), source_rows=rows))
                (g0, m0, c0), (g1, m1, c1) = outs
                assert np.array_equal(g0, g1) and m0 == m1 and np.array_equal(c0, c1), (tag, "step forms")
                new_counts = orc.recalculate_feature_counts(feats, new_groups, new_source)
                for c in range(C):
                    assert np.array_equal(eng.get_counts(B, c), new_counts[c]), (tag, "step counts", c)
                with np.errstate(divide="ignore", invalid="ignore"):
                    w = orc.normalize_weights(weights, hc)
                    obs = orc.mixture_observation_lh(w, orc.likelihood_per_component(feats, na, new_groups, new_counts, conc))
                    want_mix = np.log(obs)[~na].sum()
                if np.isfinite(want_mix):
                    assert abs(m0 - want_mix) <= 1e-10 * abs(want_mix) + 1e-16 * N * F, (tag, "step mixture", m0, want_mix)
                stats["steps"] += 1
                # the same delta as chain 0 of a two-chain batched step (chain 1: a copy of state 0 with the source rows
                # only): every chain of the batch must reproduce its single step bit for bit
                eng.set_option(step_form=0)
                try:
                    eng.copy_slot(B + 1, 0)
                    single_b = eng.step(B + 1, B + 2, changed_objects=moved.astype(np.int32), source_rows=rows)
                    stack = np.stack([clusters, groups[0]])
                    glh, mixb, chg = eng.step_batch(np.array([0, B + 1]), np.array([B, B + 2]), stack, np.array([True, False]),
                                                    np.array([0, moved.size, 2 * moved.size], dtype=np.int32),
                                                    np.concatenate([moved, moved]).astype(np.int32), np.concatenate([rows, rows]))
                except EngineError as exc:
                    if "too large" not in str(exc):
                        raise
                else:
                    # (the mixture kernel's chunk geometry depends on the number of slots of a launch: the partial sums of
                    #  a batched eval may be grouped differently -- equal to rounding, everything else bit for bit)
                    close = lambda a, b: a == b or abs(a - b) <= 1e-13 * abs(b) + 1e-16 * N * F          # noqa: E731
                    assert np.array_equal(glh[0], g0) and np.array_equal(chg[0], c0), (tag, "batched step, chain 0", glh[0], g0)
                    assert close(mixb[0], m0), (tag, "batched step mixture, chain 0", mixb[0], m0)
                    assert np.array_equal(glh[1], single_b[0]) and close(mixb[1], single_b[1]), (tag, "batched step, chain 1")
                    stats["batched"] = stats.get("batched", 0) + 1
                    # round 3: the same two proposals once more in DELTA form (both were left pending = rejected: the
                    # candidates are patched from the records the matrix-form step left), then chained: accept chain 0,
                    # propose a second delta from the new state
                    mv0 = np.flatnonzero((clusters != groups[0]).any(axis=0)).astype(np.int32)
                    mc0 = np.where(clusters[:, mv0].any(axis=0), clusters[:, mv0].argmax(axis=0), -1).astype(np.int32)
                    args_d = (np.array([0, mv0.size, mv0.size], dtype=np.int32), mv0, mc0,
                              np.array([0, moved.size, 2 * moved.size], dtype=np.int32), np.concatenate([moved, moved]).astype(np.int32),
                              np.concatenate([rows, rows]))
                    glh_d, mix_d, chg_d = eng.step_batch_delta(np.array([0, B + 1]), np.array([B, B + 2]), *args_d)
                    assert np.array_equal(glh_d, glh) and np.array_equal(chg_d, chg), (tag, "delta batched step")
                    assert close(mix_d[0], mixb[0]) and close(mix_d[1], mixb[1]), (tag, "delta batched mixture", mix_d, mixb)
                    # accept chain 1 (its slots swap roles: B + 2 is current now), move one more object in delta form -- the
                    # candidate B + 1 is patched from the records -- and compare with the matrix form on copies (B + 3, B + 4)
                    n2 = int(rng.integers(0, N))
                    k2 = int(rng.integers(-1, clusters.shape[0]))
                    cl2 = groups[0].copy()
                    cl2[:, n2] = False
                    if k2 >= 0:
                        cl2[k2, n2] = True
                    if orc.has_components([cl2] + groups[1:]).any(axis=1).all():
                        eng.copy_slot(B + 3, B + 2)
                        g_d, m_d, c_d = eng.step_delta(B + 2, B + 1, [n2], [k2])
                        g_m, m_m, c_m = eng.step(B + 3, B + 4, clusters=cl2)
                        assert np.array_equal(g_d, g_m) and np.array_equal(c_d, c_m) and close(m_d, m_m), (tag, "chained delta step")
                        for c in range(C):
                            assert np.array_equal(eng.get_counts(B + 1, c), eng.get_counts(B + 4, c)), (tag, "chained delta counts", c)
                    stats["delta"] = stats.get("delta", 0) + 1
        # the resident operator forms of the drop-in layer (round 3) on state 0, against the oracle-backed double
        if C >= 2 and N >= 2 and not big:
            forms_case(rng, eng, feats, n_groups, conc, states[0], tag, stats)
        # one-call Gibbs step from state 0: counts consistent with the source it drew
        eng.set_option(step_form=0)
        objs = np.unique(rng.integers(0, N, size=min(6, N))).astype(np.int32)
        try:
            lq, lqb, glh, mix, changed = eng.gibbs_step(0, B, objs, z=rng.random((objs.size, F)))
        except EngineError as exc:
            if "too large" not in str(exc):
                raise
        else:
            drawn = eng.get_source_rows(B, np.arange(N, dtype=np.int32))
            others = np.setdiff1d(np.arange(N), objs)
            assert np.array_equal(drawn[others], source[others]), (tag, "gibbs untouched rows")
            assert not drawn[na].any(), (tag, "gibbs NA rows")
            new_counts = orc.recalculate_feature_counts(feats, groups, drawn)
            for c in range(C):
                assert np.array_equal(eng.get_counts(B, c), new_counts[c]), (tag, "gibbs counts", c)
            with np.errstate(divide="ignore", invalid="ignore"):
                w = orc.normalize_weights(weights, orc.has_components(groups))
                obs = orc.mixture_observation_lh(w, orc.likelihood_per_component(feats, na, groups, new_counts, conc))
                want_mix = np.log(obs)[~na].sum()
            if np.isfinite(want_mix):
                assert abs(mix - want_mix) <= 1e-10 * abs(want_mix) + 1e-16 * N * F, (tag, "gibbs mixture", mix, want_mix)
            assert np.isfinite(lq) and np.isfinite(lqb) and lq <= 1e-9 and lqb <= 1e-9, (tag, lq, lqb)
            stats["gibbs"] += 1
        # round 6: RESIDENT OVERLAP -- some objects of the last component put into a SECOND group: sbe_set_groups keeps the last
        # group as the id (what an uncached evaluation ends up with, likelihood.py:126-130), counts come from the caller (once per
        # group, counts.py:28-30), every kernel form must return the oracle's value of the overlapping sample; the ids read back
        # are the last groups; count-deriving calls refuse the marked slot
        if C >= 2 and n_groups[C - 1] >= 2 and N >= 4:
            groups, weights, source, counts = states[0]
            ov = groups[C - 1].copy()
            extra = rng.integers(0, N, size=max(1, N // 8))
            ov[rng.integers(0, ov.shape[0], size=extra.size), extra] = True
            if (ov.sum(axis=0) > 1).any():
                ov_groups = groups[:C - 1] + [ov]
                ov_counts = orc.recalculate_feature_counts(feats, ov_groups, source)
                slot = B + 1
                eng.copy_slot(slot, 0)
                eng.set_groups(slot, C - 1, ov)
                want_ids = np.full(N, -1, dtype=np.int32)
                for g in range(ov.shape[0]):
                    want_ids[ov[g]] = g
                assert np.array_equal(eng.get_group_ids(slot, C - 1), want_ids), (tag, "overlap ids")
                for c in range(C):
                    eng.set_counts(slot, c, ov_counts[c])
                    eng.update_probs(slot, c)
                with np.errstate(divide="ignore", invalid="ignore"):
                    want_ov = orc.mixture_loglik(feats, na, ov_groups, ov_counts, conc, weights)
                if np.isfinite(want_ov):
                    for kernel in (MIXTURE_PACKED, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA, MIXTURE_ONEHOT):
                        eng.set_option(kernel=kernel)
                        try:
                            got_ov = eng.mixture_loglik(slot)
                        except EngineError as exc:
                            if "not applicable" in str(exc):
                                continue
                            raise
                        assert abs(got_ov - want_ov) <= 1e-10 * abs(want_ov) + 1e-16 * N * F, (tag, "overlap", kernel, got_ov, want_ov)
                    eng.set_option(kernel=MIXTURE_PACKED)
                    try:
                        eng.recount(slot)
                        raise AssertionError((tag, "recount accepted a slot with overlapping groups"))
                    except EngineError as exc:
                        assert " is in groups " in str(exc), exc
                    stats["overlap"] = stats.get("overlap", 0) + 1
    stats["cases"] += 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--big", action="store_true", help="large shapes and batches (long chunks, several block generations)")
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    stats = {"cases": 0, "evals": 0, "steps": 0, "gibbs": 0}
    t0 = last = time.time()
    while time.time() - t0 < args.seconds:
        one_case(rng, stats, big=args.big)
        if time.time() - last > 30:
            last = time.time()
            print(f"[fuzz] {time.time() - t0:5.0f} s  {stats}", flush=True)
    print(f"[fuzz] done, no mismatch: {stats}", flush=True)


if __name__ == "__main__":
    main()
