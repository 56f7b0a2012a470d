"""Randomised shape sweep: every kernel family against the oracle on ragged / edge shapes the
fixtures do not reach -- N not a multiple of 4, F not a multiple of the tile width, S = 1 ..,
C = 1..6 (compile-time 1..4 and the runtime-C instantiation), many groups, objects in no group,
all three feature-tile widths (SBE_FT), both streamed representations, both log modes."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import sbayes_oracle as orc
from sbayes_amd.engine import (LOG_PER_OBS, LOG_PRODUCT, MIXTURE_ONEHOT, MIXTURE_PACKED, MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2,
                               MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA, Engine, EngineError)

pytestmark = pytest.mark.gpu

SHAPES = [
    # N,   F,   S,  groups per component,        na_rate
    (1,    1,   1,  [1],                         0.0),
    (3,    2,   2,  [1, 1],                      0.3),
    (5,    15,  3,  [2, 1],                      0.1),
    (7,    16,  7,  [3, 2, 1],                   0.05),
    (50,   17,  5,  [2, 1, 4],                   0.03),
    (129,  65,  4,  [4, 1, 3, 2],                0.03),
    (257,  130, 6,  [3, 1, 2, 2, 5],             0.02),      # C = 5: runtime-C kernel instantiation
    (64,   33,  33, [2, 1, 2, 3, 2, 2],          0.02),      # C = 6, S = 33
    (100,  40,  20, [30, 1, 25],                 0.03),      # many groups: narrower LDS tile
    (41,   70,  2,  [5, 1],                      0.9),       # almost everything NA
    (20,   3,   254, [2, 1],                     0.05),      # maximum state count (0xFF is NA)
    (30,   9,   3,  [2, 1, 2, 2, 2, 2, 2, 2],    0.05),      # maximum component count (8)
    (2003, 70,  3,  [7, 1],                      0.03),      # several object chunks with a ragged tail
    (300,  20,  30, [60, 1, 45],                 0.03),      # tables of a 16-feature tile exceed LDS: direct (L2) gathers
]


def random_case(rng, N, F, S, n_groups, na_rate):
    x = rng.integers(0, S, size=(N, F))
    na = rng.random((N, F)) < na_rate
    feats = np.zeros((N, F, S), dtype=bool)
    nn, ff = np.nonzero(~na)
    feats[nn, ff, x[nn, ff]] = True
    groups = []
    for c, G in enumerate(n_groups):
        if c == 1:
            g = np.ones((G, N), dtype=bool) if G == 1 else None
        if c != 1 or G != 1:
            a = rng.integers(0, G + (1 if c != 1 else 0) + (G if c == 0 else 0), size=N)   # some objects in no group
            g = np.stack([a == k for k in range(G)])
        groups.append(g)
    C = len(n_groups)
    if C == 1:                                   # a lone component must cover every object with data
        groups[0] = np.stack([np.arange(N) % n_groups[0] == k for k in range(n_groups[0])])
    weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
    hc = orc.has_components(groups)
    w = hc[:, None, :] * weights[None].astype(np.float64)
    tot = w.sum(-1, keepdims=True)
    w = np.divide(w, tot, out=np.full_like(w, 1.0 / C), where=tot > 0)
    cdf = np.cumsum(w, -1)
    src_idx = np.argmax(rng.random((N, F, 1)) < cdf / cdf[..., -1:], axis=-1)
    source = np.eye(C, dtype=bool)[src_idx]
    source[~feats.any(-1)] = False
    source[~hc.any(1)] = False
    source &= hc[:, None, :]
    conc = [rng.choice([0.5, 1.0, 2.0], size=((G, F, S) if c else (F, S))) for c, G in enumerate(n_groups)]
    return feats, groups, weights, source, conc


@pytest.mark.parametrize("ft", ["16", "32", "64"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: f"N{s[0]}F{s[1]}S{s[2]}C{len(s[3])}")
def test_shape_sweep(shape, ft, monkeypatch):
    N, F, S, n_groups, na_rate = shape
    monkeypatch.setenv("SBE_FT", ft)
    rng = np.random.default_rng(hash((N, F, S, len(n_groups))) % (2 ** 32))
    feats, groups, weights, source, conc = random_case(rng, N, F, S, n_groups, na_rate)
    try:
        _run_case(feats, groups, weights, source, conc, n_groups, rng)
    except EngineError as exc:                     # a forced tile too wide for the tables is a valid refusal
        assert "too large for LDS" in str(exc) and (ft != "16" or "forced tile width" in str(exc)), exc


def test_default_tile_width_never_refuses():
    """Without SBE_FT the engine picks a tile width that fits for every shape of the sweep."""
    for shape in SHAPES:
        N, F, S, n_groups, na_rate = shape
        rng = np.random.default_rng(7)
        feats, groups, weights, source, conc = random_case(rng, N, F, S, n_groups, na_rate)
        _run_case(feats, groups, weights, source, conc, n_groups, rng, light=True)


def test_forced_direct_mode(monkeypatch):
    """SBE_DIRECT=1 forces the no-LDS-table form (tables gathered through L2) for every shape, not
    only the one whose tables do not fit."""
    monkeypatch.setenv("SBE_DIRECT", "1")
    for shape in SHAPES:
        N, F, S, n_groups, na_rate = shape
        rng = np.random.default_rng(11)
        feats, groups, weights, source, conc = random_case(rng, N, F, S, n_groups, na_rate)
        _run_case(feats, groups, weights, source, conc, n_groups, rng, light=True)


def _run_case(feats, groups, weights, source, conc, n_groups, rng, light=False):
    N = feats.shape[0]
    na = ~feats.any(-1)
    hc = orc.has_components(groups)
    covered = hc.any(axis=1)
    with Engine(feats, n_groups, n_slots=2) as eng:
        C = len(n_groups)
        for c in range(C):
            eng.set_concentration(c, conc[c])
            eng.set_groups(0, c, groups[c])
        eng.set_source(0, source)
        eng.recount(0)
        counts = orc.recalculate_feature_counts(feats, groups, source)
        for c in range(C):
            assert np.array_equal(eng.get_counts(0, c), counts[c])
            eng.update_probs(0, c)
            assert np.array_equal(eng.get_probs(0, c), orc.component_probs(counts[c], conc[c]))
        eng.set_weights(0, weights)
        lh = orc.likelihood_per_component(feats, na, groups, counts, conc)
        assert np.array_equal(eng.likelihood_per_component(0), lh)
        with np.errstate(invalid="ignore", divide="ignore"):
            w = orc.normalize_weights(weights, hc)
            obs = orc.mixture_observation_lh(w, lh)
            got_w = eng.weights_normalized(0)
            assert np.array_equal(got_w[covered], w[covered])
            assert np.array_equal(eng.observation_lh(0)[covered], obs[covered])
            assert np.array_equal(eng.likelihood_per_component_exact(0),
                                  orc.likelihood_per_component_exact(feats, na, groups, counts, conc, source))
            want = np.log(obs)[~na].sum()
        for kernel in (MIXTURE_PACKED, MIXTURE_ONEHOT, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS,
                       MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED_TUPLE_MFMA):
            for log_mode in (LOG_PER_OBS, LOG_PRODUCT):
                eng.set_option(kernel=kernel, log_mode=log_mode)
                try:
                    got = eng.mixture_loglik(0)
                except EngineError as exc:
                    if kernel in (MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA) and "not applicable" in str(exc):
                        continue                 # more than 64 distinct group tuples / table too large
                    raise
                if np.isfinite(want):
                    assert abs(got - want) <= 1e-10 * max(abs(want), 1e-300), (kernel, log_mode, got, want)
                else:
                    assert (np.isnan(got) and np.isnan(want)) or got == want, (kernel, log_mode, got, want)
        if light:
            return
        # collapsed likelihood and the stateless forms
        for c in range(C):
            pg, pf = eng.collapsed_loglik(0, c, per_feature=True)
            a = conc[c] if conc[c].ndim == 3 else np.broadcast_to(conc[c], counts[c].shape)
            want_pf = np.stack([orc.dirichlet_categorical_logpdf(counts[c][g], a[g]) for g in range(n_groups[c])])
            np.testing.assert_allclose(pf, want_pf, rtol=3e-6, atol=2e-6)
            np.testing.assert_allclose(pg, orc.collapsed_group_logliks(counts[c], conc[c]), rtol=2e-6, atol=1e-5)
            assert np.array_equal(eng.effect_counts(groups[c], source[..., c]), counts[c])
            sub = rng.choice(N, size=max(1, N // 3), replace=False)
            assert np.array_equal(eng.effect_counts(groups[c], source[..., c], sub),
                                  orc.compute_effect_counts(feats, groups[c], source[..., c], sub))
            assert np.array_equal(eng.normalize_tables(counts[c], conc[c]), orc.component_probs(counts[c], conc[c]))


@pytest.mark.parametrize("shape", [
    # N,   F,   S,  groups,     n_slots  (what it exercises in k_mixture_tuple64)
    (203,  72,  4,  [3, 1],     19),     # narrow last tile (8 features: sub-row mode), batch not a multiple of 8
    (203,  64,  4,  [3, 1],     9),      # no ragged tile
    (150,  100, 3,  [2, 1, 2],  8),      # last tile 36 features wide: full-tile mode with dead lanes; C = 3
    (97,   130, 20, [6, 1],     11),     # table beyond 64 KiB: 32-bit tuple-block offsets; 2-feature last tile
    (1100, 70,  3,  [4, 1],     3),      # fewer than 8 slots (slot-major order), short chunks
    (2203, 70,  3,  [4, 1],     128),    # long chunks: more than 64 quads per wave (two offset groups)
    (40,   72,  3,  [2, 1],     600),    # more slots than one generation of blocks per XCD holds (64): two generations
], ids=lambda s: f"N{s[0]}F{s[1]}S{s[2]}C{len(s[3])}B{s[4]}")
def test_tuple_kernel_batches(shape):
    """Batches through the scalar-unit group-tuple kernel: every slot holds a different state; the batched
    launch (XCD-dealt, generation-ordered blocks) must give each slot the value of its own single evaluation
    and of the oracle.  One slot has fewer group tuples than the launch's maximum (its unused table rows are skipped)."""
    N, F, S, n_groups, B = shape
    rng = np.random.default_rng(N * 1000 + F)
    feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, 0.05)
    na = ~feats.any(-1)
    with Engine(feats, n_groups, n_slots=B) as eng:
        C = len(n_groups)
        for c in range(C):
            eng.set_concentration(c, conc[c])
        want = []
        for b in range(B):
            a = rng.integers(0, 2 * n_groups[0], size=N)
            if b == 1:
                a[:] = 2 * n_groups[0] - 1          # nobody in a cluster: fewer group tuples than the other slots
            groups = [np.stack([a == k for k in range(n_groups[0])])] + groups0[1:]
            weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
            hc = orc.has_components(groups)
            src_idx = np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)      # a random available component
            source = np.eye(C, dtype=bool)[src_idx]
            source[na] = False
            eng.load_state(b, groups, weights, source=source)
            for c in range(C):
                eng.update_probs(b, c)
            counts = orc.recalculate_feature_counts(feats, groups, source)
            want.append(orc.mixture_loglik(feats, na, groups, counts, conc, weights))
        want = np.array(want)
        for kernel in (MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2):
            eng.set_option(kernel=kernel)
            try:
                got = eng.mixture_loglik_batch(0, B)
            except EngineError as exc:               # the matrix-pipe form holds at most 8 tuples
                assert kernel == MIXTURE_PACKED_TUPLE_MFMA and "not applicable" in str(exc), exc
                continue
            np.testing.assert_allclose(got, want, rtol=1e-10, err_msg=f"kernel {kernel}")
            singles = np.array([eng.mixture_loglik(b) for b in range(B)])
            np.testing.assert_allclose(singles, want, rtol=1e-10, err_msg=f"kernel {kernel} (single)")
        # sub-ranges of slots
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE)
        if B > 2:
            np.testing.assert_allclose(eng.mixture_loglik_batch(1, B - 2), want[1:B - 1], rtol=1e-10)


@pytest.mark.parametrize("shape", [
    # N,   F,   S,  groups,     n_slots   (what it exercises in k_mixture_tuple_mfma)
    (50,   30,  5,  [2, 1],     1),       # one slot of a 16-slot block; odd tuple count (padding tuple); 2 k-blocks padded to 4
    (203,  72,  4,  [3, 1],     19),      # batch not a multiple of 16; 9 column tiles, one split
    (97,   130, 20, [6, 1],     40),      # 7 tuples -> 4 M tiles (half-quad epilogue steps); 82 column tiles, several splits
    (1000, 37,  3,  [5, 1],     64),      # 32 k-blocks (the headline's depth), last column tile 15 columns wide
    (130,  40,  6,  [4],        33),      # C = 1 (every object in a group)
    (120,  36,  5,  [2, 1, 1],  17),      # C = 3
    (90,   25,  4,  [1, 1, 1, 1], 16),    # C = 4, exactly one block of slots
    (40,   72,  3,  [2, 1],     600),     # many blocks per column split
    (1301, 20,  3,  [2, 1],     16),      # 41 k-blocks padded to 44
    # wide forms (round 6): more than 8 group tuples -> 4 slots x <= 32 tuples, 2 slots x <= 64 tuples per block
    (300,  40,  4,  [3, 1, 4],  37),      # up to 4 x 5 = 20 tuples: 4 slots per block, 3 M tiles; batch not a multiple of 4
    (100,  36,  5,  [3, 1, 6],  600),     # the south_america layout (up to 28 tuples), many blocks
    (260,  24,  3,  [3, 1, 3, 3], 9),     # C = 4, up to 64 tuples: 2 slots per block, 4 M tiles; odd batch
    (700,  33,  6,  [5, 1, 7],  10),      # up to 48 tuples: 2 slots per block, 3 M tiles; 11 k-blocks padded to 12
    (150,  20,  2,  [8, 1],     21),      # 9 tuples: the smallest wide case (4 slots, 2 M tiles)
    (5200, 12,  3,  [5, 1],     10),      # many objects, few tuples: the A image of 16 slots (3 x 84 KB) does not fit LDS -> 4 slots per block
    (8900, 6,   2,  [2, 1],     5),       # 140 k-blocks: 4 slots x 1 M tile, the A image 140 KB -- the largest that fits a CU's LDS
], ids=lambda s: f"N{s[0]}F{s[1]}S{s[2]}C{len(s[3])}B{s[4]}")
def test_mfma_kernel_batches(shape):
    """Batches through the matrix-pipe group-tuple kernel (counts per (slot, tuple, feature, state) by i8 MFMA, one log per
    table entry): every slot holds a different state and must get the oracle's value and its own single-launch value.
    States include inapplicable feature states (probability exactly 0 where no observation falls) and one slot with
    fewer group tuples than the launch's maximum."""
    N, F, S, n_groups, B = shape
    rng = np.random.default_rng(N * 1000 + F + B)
    feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, 0.05)
    # inapplicable states: the last state of every third feature never occurs and has concentration 0 (p = 0 exactly)
    if S > 2:
        dead = np.arange(F) % 3 == 0
        moved = feats[:, dead, S - 1].copy()
        feats[:, dead, S - 1] = False
        feats[:, dead, 0] |= moved
        for c in range(len(conc)):
            conc[c] = conc[c].copy()
            conc[c][..., dead, S - 1] = 0.0
    na = ~feats.any(-1)
    C = len(n_groups)
    with Engine(feats, n_groups, n_slots=B) as eng:
        for c in range(C):
            eng.set_concentration(c, conc[c])
        want = []
        for b in range(B):
            if C == 1:
                groups = groups0
            else:
                a = rng.integers(0, 2 * n_groups[0], size=N)
                if b == 1:
                    a[:] = 2 * n_groups[0] - 1          # nobody in a cluster: fewer group tuples than the other slots
                groups = [np.stack([a == k for k in range(n_groups[0])])] + groups0[1:]
            weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
            hc = orc.has_components(groups)
            src_idx = np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)      # a random available component
            source = np.eye(C, dtype=bool)[src_idx]
            source[na] = False
            source[~hc.any(1)] = False
            eng.load_state(b, groups, weights, source=source)
            for c in range(C):
                eng.update_probs(b, c)
            counts = orc.recalculate_feature_counts(feats, groups, source)
            with np.errstate(divide="ignore", invalid="ignore"):
                want.append(orc.mixture_loglik(feats, na, groups, counts, conc, weights))
        want = np.array(want)
        assert np.all(np.isfinite(want))
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
        got = eng.mixture_loglik_batch(0, B)
        assert "k_mixture_tuple_mfma" in eng.last_mixture_kernel()
        if N >= 5000:
            assert "4 slots x M tiles 1" in eng.last_mixture_kernel(), eng.last_mixture_kernel()
        np.testing.assert_allclose(got, want, rtol=1e-10)
        assert np.array_equal(eng.mixture_loglik_batch(0, B), got)                     # fixed reduction order
        picks = sorted(set([0, B - 1, B // 2]))
        singles = np.array([eng.mixture_loglik(b) for b in picks])
        np.testing.assert_allclose(singles, want[picks], rtol=1e-10)
        if B > 2:
            np.testing.assert_allclose(eng.mixture_loglik_batch(1, B - 2), want[1:B - 1], rtol=1e-10)
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE)
        try:
            other = eng.mixture_loglik_batch(0, B)                                     # the vector-pipe form of the same table
        except EngineError as exc:                   # (its LDS table image [tuples][S + 1][tile] can exceed a CU's LDS where the
            assert "not applicable" in str(exc) and max(n_groups) > 4, exc             #  matrix-pipe form still runs: 48 tuples x 6 states)
        else:
            np.testing.assert_allclose(other, got, rtol=1e-12)


def test_mfma_kernel_default_choice_and_zero_probability():
    """SBE_MIXTURE_PACKED picks the matrix-pipe form from 320 slots per launch on (the vector-pipe form below), and a
    zero-probability OBSERVED state gives -inf like the reference's log(0), in that slot only."""
    rng = np.random.default_rng(5)
    N, F, S, B = 60, 12, 3, 320
    feats, groups0, _w, _s, _conc = random_case(rng, N, F, S, [2, 1], 0.0)
    with Engine(feats, [2, 1], n_slots=B) as eng:
        a = rng.integers(0, 4, size=N)
        groups = [np.stack([a == k for k in range(2)])] + groups0[1:]
        probs = [rng.dirichlet(np.ones(S), size=(2, F)).astype(np.float32), rng.dirichlet(np.ones(S), size=(1, F)).astype(np.float32)]
        weights = rng.dirichlet(np.ones(2), size=F).astype(np.float32)
        eng.load_state(0, groups, weights, probs=probs)
        for b in range(1, B):
            eng.copy_slot(b, 0)
        bad = [p.copy() for p in probs]
        bad[0][:, 3, :] = 0.0
        bad[1][:, 3, :] = 0.0                           # feature 3: every component gives probability 0
        eng.set_probs(7, 0, bad[0]); eng.set_probs(7, 1, bad[1])
        eng.set_option(kernel=MIXTURE_PACKED)
        got = eng.mixture_loglik_batch(0, B)
        assert "k_mixture_tuple_mfma" in eng.last_mixture_kernel()
        ref = eng.mixture_loglik_batch(0, B - 1)
        assert "k_mixture_tuple64" in eng.last_mixture_kernel() or "k_mixture_combo" in eng.last_mixture_kernel()
        assert got[7] == -np.inf and ref[7] == -np.inf
        keep = np.arange(B - 1) != 7
        np.testing.assert_allclose(got[:B - 1][keep], ref[keep], rtol=1e-12)
        assert np.all(np.isfinite(got[np.arange(B) != 7]))


@pytest.mark.parametrize("S", [1, 2])
def test_mfma_kernel_probabilities_next_to_one(S):
    """States whose every observation has probability ~1 (a single state per feature; or a second state of probability 1e-7
    that is never observed): the log-likelihood is ~0 -- the rounding of the float32 weights -- and the parts of a log
    (exponent, table value, series) must not cancel it away.  Tolerance of tools/fuzz_gpu.py: 1e-10 relative + 1e-16 per
    observation (found by the fuzzer: a first form of the kernel's own log summed k ln2 and log m apart, 2e-12 off)."""
    rng = np.random.default_rng(1503 + S)
    N, F, B = 143, 53, 20
    n_groups = [1, 1, 3]
    feats = np.zeros((N, F, S), dtype=bool)
    feats[..., 0] = rng.random((N, F)) > 0.03
    na = ~feats.any(-1)
    conf = rng.integers(0, 3, size=N)
    groups0 = [np.ones((1, N), dtype=bool), np.ones((1, N), dtype=bool), np.stack([conf == k for k in range(3)])]
    with Engine(feats, n_groups, n_slots=B) as eng:
        want = []
        for b in range(B):
            a = rng.random(N) < 0.5
            groups = [a[None, :]] + groups0[1:]
            probs = []
            for g in n_groups:
                p = np.zeros((g, F, S), dtype=np.float32)
                p[..., 0] = 1.0 if S == 1 else np.float32(1.0) - np.float32(1e-7) * rng.integers(0, 4, size=(g, F)).astype(np.float32)
                if S == 2:
                    p[..., 1] = np.float32(1.0) - p[..., 0]
                probs.append(p)
            weights = rng.dirichlet(np.ones(3), size=F).astype(np.float32)
            eng.load_state(b, groups, weights, probs=probs)
            lh = np.empty((N, F, 3))
            for c in range(3):                                   # (the reference's composition, SURVEY 8(d), from given tables)
                orc.compute_component_likelihood(feats, probs[c], groups[c], np.arange(n_groups[c]), lh[..., c])
            lh[na] = 1.0
            w = orc.normalize_weights(weights, orc.has_components(groups))
            want.append(np.log(orc.mixture_observation_lh(w, lh))[~na].sum())
        want = np.array(want)
        tol = 1e-10 * np.abs(want) + 1e-16 * N * F
        for kern in (MIXTURE_PACKED_TUPLE_MFMA, MIXTURE_PACKED_TUPLE):
            eng.set_option(kernel=kern)
            got = eng.mixture_loglik_batch(0, B)
            assert np.all(np.abs(got - want) <= tol), (kern, np.abs(got - want).max(), tol.min(), got[:3], want[:3])


@pytest.mark.parametrize("n_split", [1, 2, 3, 5])
def test_mfma_kernel_final_reduction_in_kernel(n_split, monkeypatch):
    """The matrix-pipe kernel finishes its own reduction: the LAST of a slot group's column-split blocks (tickets) adds the
    group's partial sums in split order.  Whichever block that is, the value is the same -- 60 launches return the same bits,
    synchronous and asynchronous form alike -- and equals the vector-pipe form's; a batch that is not a multiple of the 16
    slots of a block and a batch starting in the middle of the slots go through the same path."""
    monkeypatch.setenv("SBE_MFMA_SPLIT", str(n_split))
    rng = np.random.default_rng(77 + n_split)
    N, F, S, B = 150, 40, 4, 83
    feats, groups0, _w, _s, _conc = random_case(rng, N, F, S, [3, 1], 0.03)
    with Engine(feats, [3, 1], n_slots=B) as eng:
        for b in range(B):
            a = rng.integers(0, 5, size=N)
            groups = [np.stack([a == k for k in range(3)])] + groups0[1:]
            probs = [rng.dirichlet(np.ones(S), size=(3, F)).astype(np.float32), rng.dirichlet(np.ones(S), size=(1, F)).astype(np.float32)]
            eng.load_state(b, groups, rng.dirichlet(np.ones(2), size=F).astype(np.float32), probs=probs)
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE)
        ref = eng.mixture_loglik_batch(0, B)
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
        got = eng.mixture_loglik_batch(0, B)
        assert "k_mixture_tuple_mfma" in eng.last_mixture_kernel()
        np.testing.assert_allclose(got, ref, rtol=1e-12)
        for i in range(60):
            if i % 2:
                again = eng.mixture_loglik_batch(0, B)
            else:
                eng.mixture_loglik_batch_async(0, B)
                eng.mixture_loglik_batch_async(0, B)              # (two launches in flight share the tickets, one after the other)
                again = eng.fetch_results(0, B)
            assert np.array_equal(again, got), i
        assert np.array_equal(eng.mixture_loglik_batch(5, 40), got[5:45])
        assert eng.mixture_loglik(B - 1) == pytest.approx(got[B - 1], rel=1e-12)


def test_many_slots_one_launch():
    """More resident states than the former limit of 4096 (sbe_create takes up to 16384): one launch over 5000 small states through
    the matrix-pipe form -- slot groups beyond the 256th, a last group of 8 slots -- equals the vector-pipe form slot by slot."""
    rng = np.random.default_rng(50)
    N, F, S, B = 40, 9, 3, 5000
    feats, groups0, _w, _s, _conc = random_case(rng, N, F, S, [2, 1], 0.02)
    with Engine(feats, [2, 1], n_slots=B) as eng:
        a = rng.integers(0, 3, size=N)
        groups = [np.stack([a == k for k in range(2)])] + groups0[1:]
        for b in range(0, B, 250):                     # 20 distinct states, copied into the slots between
            probs = [rng.dirichlet(np.ones(S), size=(2, F)).astype(np.float32), rng.dirichlet(np.ones(S), size=(1, F)).astype(np.float32)]
            eng.load_state(b, groups, rng.dirichlet(np.ones(2), size=F).astype(np.float32), probs=probs)
            for k in range(1, 250):
                eng.copy_slot(b + k, b)
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
        got = eng.mixture_loglik_batch(0, B)
        assert "k_mixture_tuple_mfma" in eng.last_mixture_kernel()
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE)
        ref = eng.mixture_loglik_batch(0, B)
        np.testing.assert_allclose(got, ref, rtol=1e-12)
        assert np.array_equal(got.reshape(20, 250), np.repeat(got[::250, None], 250, axis=1))
    with pytest.raises(EngineError, match="n_slots"):
        Engine(feats, [2, 1], n_slots=16385)


@pytest.mark.parametrize("shape", [
    # N,   F,   S,  groups,        n_slots   (what it exercises in the pattern-sorted rows form)
    (1203, 70,  6,  [4, 1, 5, 3],  20),      # C = 4, 16 patterns possible, ragged last tile (70 = 2 * 32 + 6), runs of every length
    (777,  96,  4,  [3, 1, 2],     17),      # C = 3, three whole tiles
    (640,  33,  3,  [2, 1],        32),      # C = 2, a 1-feature last tile
    (515,  64,  5,  [6],           16),      # C = 1: a single pattern (one run)
], ids=lambda s: f"N{s[0]}F{s[1]}S{s[2]}C{len(s[3])}B{s[4]}")
def test_rows_kernel_pattern_sorted_objects(shape, monkeypatch):
    """k_mixture_rows with the slot's objects sorted by has_components pattern (weights in registers, state bytes gathered
    through the permutation; k_rowsort): every slot's value equals the oracle's and the unsorted form's to 1e-10 / 1e-12,
    the same call returns the same bits, and a slot whose groups change is re-sorted (its value follows)."""
    N, F, S, n_groups, B = shape
    monkeypatch.setenv("SBE_ROWS_SORTED", "2")
    monkeypatch.setenv("SBE_ROWS_FT", "32")
    rng = np.random.default_rng(N + F)
    feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, 0.04)
    na = ~feats.any(-1)
    C = len(n_groups)

    def random_state():
        if C == 1:
            groups = groups0
        else:
            a = rng.integers(0, 2 * n_groups[0], size=N)
            groups = [np.stack([a == k for k in range(n_groups[0])])] + groups0[1:]
        weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
        hc = orc.has_components(groups)
        src_idx = np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)
        source = np.eye(C, dtype=bool)[src_idx]
        source[na] = False
        source[~hc.any(1)] = False
        return groups, weights, source

    def oracle_ll(groups, weights, source):
        counts = orc.recalculate_feature_counts(feats, groups, source)
        with np.errstate(divide="ignore", invalid="ignore"):
            w = orc.normalize_weights(weights, orc.has_components(groups))
            obs = orc.mixture_observation_lh(w, orc.likelihood_per_component(feats, na, groups, counts, conc))
            return np.log(obs)[~na & orc.has_components(groups).any(1)[:, None]].sum()

    with Engine(feats, n_groups, n_slots=B) as eng:
        for c in range(C):
            eng.set_concentration(c, conc[c])
        want = []
        for b in range(B):
            groups, weights, source = random_state()
            if C > 1 and not orc.has_components(groups).any(1).all():
                groups[1][:] = True                                  # (every object keeps a component: the reference asserts it)
            eng.load_state(b, groups, weights, source=source)
            for c in range(C):
                eng.update_probs(b, c)
            want.append(oracle_ll(groups, weights, source))
        want = np.array(want)
        eng.set_option(kernel=MIXTURE_PACKED_GENERAL)
        got = eng.mixture_loglik_batch(0, B)
        assert "pattern-sorted objects" in eng.last_mixture_kernel(), eng.last_mixture_kernel()
        np.testing.assert_allclose(got, want, rtol=1e-10)
        assert np.array_equal(eng.mixture_loglik_batch(0, B), got)
        # a slot's groups change: its order is rebuilt, the others stay
        groups, weights, source = random_state()
        if C > 1:
            groups[1][:] = True
        eng.load_state(3, groups, weights, source=source)
        for c in range(C):
            eng.update_probs(3, c)
        got2 = eng.mixture_loglik_batch(0, B)
        assert abs(got2[3] - oracle_ll(groups, weights, source)) <= 1e-10 * abs(got2[3])
        keep = np.arange(B) != 3
        assert np.array_equal(got2[keep], got[keep])
        eng.set_option(kernel=MIXTURE_PACKED_V2)                         # the older general kernel: same values to rounding
        np.testing.assert_allclose(eng.mixture_loglik_batch(0, B), got2, rtol=1e-12)


def test_mfma_wave_specialised_form():
    """The opt-in wave-specialised form of the matrix-pipe kernel (SBE_MFMA_WS=1: producer waves count, consumer waves evaluate;
    sbe_mixture_mfma_ws.hip) gives the oracle's values.  The switch is read once per process, so this runs in a child."""
    import subprocess
    code = r'''
import sys
import numpy as np
sys.path.insert(0, %r)
from oracle import sbayes_oracle as orc
from sbayes_amd.engine import MIXTURE_PACKED_TUPLE_MFMA, Engine
from tests.test_gpu_shapes import random_case
for (N, F, S, n_groups, B) in [(203, 72, 4, [3, 1], 19), (1000, 37, 3, [5, 1], 64), (120, 36, 5, [2, 1, 1], 17), (97, 130, 20, [6, 1], 40), (130, 40, 6, [4], 33)]:
    rng = np.random.default_rng(N + B)
    feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, 0.05)
    na = ~feats.any(-1)
    C = len(n_groups)
    with Engine(feats, n_groups, n_slots=B) as eng:
        for c in range(C):
            eng.set_concentration(c, conc[c])
        want = []
        for b in range(B):
            if C == 1:
                groups = groups0
            else:
                a = rng.integers(0, 2 * n_groups[0], size=N)
                groups = [np.stack([a == k for k in range(n_groups[0])])] + groups0[1:]
            weights = rng.dirichlet(np.ones(C), size=F).astype(np.float32)
            hc = orc.has_components(groups)
            source = np.eye(C, dtype=bool)[np.argmax(rng.random((N, F, C)) * hc[:, None, :], axis=-1)]
            source[na] = False
            source[~hc.any(1)] = False
            eng.load_state(b, groups, weights, source=source)
            for c in range(C):
                eng.update_probs(b, c)
            counts = orc.recalculate_feature_counts(feats, groups, source)
            want.append(orc.mixture_loglik(feats, na, groups, counts, conc, weights))
        eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
        got = eng.mixture_loglik_batch(0, B)
        assert "k_mixture_tuple_mfma_ws" in eng.last_mixture_kernel(), eng.last_mixture_kernel()
        np.testing.assert_allclose(got, np.array(want), rtol=1e-10)
        assert np.array_equal(eng.mixture_loglik_batch(0, B), got)
print("ws ok")
''' % str(Path(__file__).resolve().parent.parent)
    env = dict(os.environ, SBE_MFMA_WS="1", SBE_MFMA_SMALL_SL4="0")      # (small launches would take four slots per block: no ws form there)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "ws ok" in res.stdout, res.stdout[-2000:] + res.stderr[-3000:]
