"""Round-3 delta / resident forms of the drop-in layer on the device (include/sbe_engine.h "round 3"): what the
unchanged reference sampler asks per MCMC step, as object lists and changed rows.  Every form is checked against the
oracle-backed double (tests/_fake_engine.py: the NumPy restatement of the reference expressions) on the same state:

  counts_delta                 update_feature_counts               sbayes/sampling/counts.py:55-95        bit-exact
  set_counts_rows              (bind cache delta upload)                                                  bit-exact
  given_unchanged_lh           component_likelihood_given_unchanged operators.py:863-928   bit-exact at T = 1,
                                                                                            2e-6 tempered (float32 powf)
  cluster_posterior_marginals  AlterCluster.compute_cluster_posterior operators.py:1035-1073              1e-9 (log space)
  jump_lh_resident             ClusterJump.get_jump_lh              operators.py:1679-1722                1e-9
"""
import numpy as np
import pytest

from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_state, make_workload
from tests._fake_engine import FakeEngine
from tests._fixtures import load_npz

pytestmark = pytest.mark.gpu


# shapes that reach the corners of the resident operator kernels (N, F, S, K, groups of the extra confounders, ragged states)
_SHAPES = {
    "wide": (260, 150, 33, 4, (3, 5), True),            # several 64-feature tiles, S = 33 (rows beyond the register form), C = 4
    "long": (90, 530, 7, 3, (4, 2), True),              # F > 256: the fused kernels' later passes over the features
    "five": (70, 40, 4, 2, (2, 3, 2), True),            # C = 5: more components than the fused kernels hold in registers
    "eight": (50, 24, 3, 2, (2, 2, 2, 2, 2, 2), True),  # C = 8: NumPy sums eight terms by its eight-accumulator tree
}


def _workload(name):
    if name in ("south_america", "cfg1_fixture"):
        fx = load_npz("south_america" if name == "south_america" else "cfg1")
        unif = fx.states_per_feature.astype(np.float64)
        return fx.features, fx.groups, fx.conc, fx.weights, fx.source, fx.counts, unif
    if name in _SHAPES:
        wl = make_workload(name, shape=_SHAPES[name])
    else:
        wl = make_workload(name)
    from oracle import sbayes_oracle as orc
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    return wl.features, wl.groups, wl.concentration, wl.weights, wl.source, counts, wl.states_per_feature.astype(np.float64)


def _pair(name):
    feats, groups, conc, weights, source, counts, unif = _workload(name)
    n_groups = [g.shape[0] for g in groups]
    eng, fake = Engine(feats, n_groups, n_slots=3), FakeEngine(feats, n_groups)
    for e in (eng, fake):
        for c in range(len(groups)):
            e.set_concentration(c, conc[c])
            e.set_groups(0, c, groups[c])
            e.set_counts(0, c, counts[c])
        e.set_source(0, source)
        e.set_weights(0, weights)
        e.set_uniform_counts(unif)
    for c in range(len(groups)):
        eng.update_probs(0, c)
    return eng, fake, groups, source, counts


def _ids(groups, objs, off):
    sub = groups[:, objs]
    return np.where(sub.any(axis=0), sub.argmax(axis=0) + off, -1).astype(np.int32)


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide"])
def test_counts_delta_and_row_uploads(name):
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(7)
        N, F, C = source.shape
        K = groups[0].shape[0]
        off = eng.group_offsets
        for n in (1, 2, 30, min(N, 700)):
            objs = np.sort(rng.choice(N, size=min(n, N), replace=False))
            new_clusters = groups[0].copy()
            new_clusters[:, objs] = False
            move = rng.integers(0, K + 1, size=objs.size)                  # K = leaves every cluster
            new_clusters[move[move < K], objs[move < K]] = True
            new_source = source.copy()
            pick = rng.integers(0, C + 1, size=(objs.size, F))             # C = no source (an NA-like row)
            new_source[objs] = pick[..., None] == np.arange(C)
            gid_old = np.stack([_ids(groups[c], objs, off[c]) for c in range(C)])
            gid_new = gid_old.copy()
            gid_new[0] = _ids(new_clusters, objs, 0)
            so = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
            sn = np.where(new_source[objs].any(-1), new_source[objs].argmax(-1), 255).astype(np.uint8)
            touched, diff = eng.counts_delta(objs, gid_old, gid_new, so, sn)
            want_t, want_d = fake.counts_delta(objs, gid_old, gid_new, so, sn)
            assert np.array_equal(touched, want_t)
            assert diff.dtype == np.float32 and np.array_equal(diff, want_d), (name, n)
            # ... and it is the reference's new_counts - old_counts, component by component
            from oracle import sbayes_oracle as orc
            all_groups_new = [new_clusters] + list(groups[1:])
            for c in range(C):
                ref = (orc.compute_effect_counts(fake.features, all_groups_new[c], new_source[..., c], objs)
                       - orc.compute_effect_counts(fake.features, groups[c], source[..., c], objs))
                full = np.zeros_like(ref)
                mine = (touched >= off[c]) & (touched < off[c + 1])
                full[touched[mine] - off[c]] = diff[mine]
                assert np.array_equal(full, ref), (name, n, c)
            # ... and a slot that holds the old counts FOLLOWS (sbe_counts_delta_apply): counts += difference, the probability
            # rows of the touched groups rebuilt -- what set_counts_rows(update_probs=True) with the new rows leaves, bit for
            # bit; inside the tile kernel (n <= 256) and behind the general one
            eng.copy_slot(1, 0)
            eng.copy_slot(2, 0)
            t2, d2 = eng.counts_delta(objs, gid_old, gid_new, so, sn, follow_slot=1, update_probs=True)
            assert np.array_equal(t2, touched) and np.array_equal(d2, diff), (name, n, "difference with a following slot")
            comp_of = np.searchsorted(off, touched, side="right") - 1
            new_rows = np.stack([counts[c][g - off[c]] for g, c in zip(touched, comp_of)]) + diff
            assert (new_rows >= 0).all()
            eng.set_counts_rows(2, touched, new_rows, update_probs=True)
            for c in range(C):
                assert np.array_equal(eng.get_counts(1, c), eng.get_counts(2, c)), (name, n, c, "following counts")
                assert np.array_equal(eng.get_probs(1, c), eng.get_probs(2, c)), (name, n, c, "following probability rows")
            assert eng.mixture_loglik(1) == eng.mixture_loglik(2), (name, n)            # (the tile-transposed copies too)
            eng.copy_slot(1, 0)
            eng.counts_delta(objs, gid_old, gid_new, so, sn, follow_slot=1)              # counts only: the tables stay
            for c in range(C):
                assert np.array_equal(eng.get_counts(1, c), eng.get_counts(2, c)), (name, n, c, "following counts, no tables")
                assert np.array_equal(eng.get_probs(1, c), eng.get_probs(0, c)), (name, n, c)
            assert np.array_equal(eng.get_source_rows(1, objs), source[objs])              # (and so does the source)
            # ... update_source: the slot's source rows of the subset become the new rows (what set_source_rows leaves)
            eng.copy_slot(1, 0)
            eng.counts_delta(objs, gid_old, gid_new, so, sn, follow_slot=1, update_probs=True, update_source=True)
            eng.set_source_rows(2, objs, new_source[objs])
            assert np.array_equal(eng.get_source_rows(1, objs), new_source[objs]), (name, n, "following source rows")
            with np.errstate(divide="ignore"):
                assert np.array_equal(eng.source_prior(1), eng.source_prior(2), equal_nan=True), (name, n)
        # delta upload of count rows: only the listed groups change
        for c in range(C):
            g = int(rng.integers(0, groups[c].shape[0]))
            row = rng.integers(0, 50, size=counts[c][g].shape).astype(np.float32)
            eng.set_counts_rows(0, [off[c] + g], row[None])
            want = counts[c].copy()
            want[g] = row
            assert np.array_equal(eng.get_counts(0, c), want)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide", "long", "five", "eight"])
def test_resident_operator_forms(name):
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(11)
        N = source.shape[0]
        K = groups[0].shape[0]
        for temp, ptemp in ((1.0, 1.0), (1.3, 1.5)):
            for k in range(min(K, 3)):
                # component_likelihood_given_unchanged: subsets of 1, a few, and many objects
                for n in (1, 9, min(N, 300)):
                    objs = np.sort(rng.choice(N, size=min(n, N), replace=False))
                    got = eng.given_unchanged_lh(0, k, objs, temp, ptemp)
                    want = fake.given_unchanged_lh(0, k, objs, temp, ptemp)
                    assert got.dtype == np.float32 and got.shape == want.shape
                    if temp == 1.0:
                        assert np.array_equal(got, want), (name, k, n)
                    else:
                        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
                # GibbsSampleSource's posterior rows of a subset (operators.py:554-574), any component count
                objs = np.sort(rng.choice(N, size=min(N, 17), replace=False))
                got = eng.source_posterior(0, objs, temp, ptemp)
                want = fake.source_posterior(0, objs, temp, ptemp)
                if temp == 1.0 and ptemp == 1.0:
                    assert np.array_equal(got, want), (name, k, "source_posterior")
                else:
                    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-7)
                # cluster-membership marginals of the available objects with the candidate table built on the device
                available = np.flatnonzero(~groups[0].any(axis=0) | groups[0][k])
                got = eng.cluster_posterior_marginals(0, k, available, temp, ptemp)
                underflows = source.shape[1] > 300       # the reference's linear-space products are 0 / denormal there:
                if not underflows:                       # the device's sums of logs are checked against its own explicit-table form below
                    want = fake.cluster_posterior_marginals(0, k, available, temp, ptemp)
                    np.testing.assert_allclose(got, want, rtol=1e-9 if temp == 1.0 else 2e-6, atol=1e-9)
                assert np.isfinite(got).all()
                # the explicit-table form gives the same numbers (same kernel, table from the stateless a10 call)
                table = eng.normalize_tables(counts[0][[k]], fake.conc[0], temperature=temp, prior_temperature=ptemp,
                                             unif_counts=fake.unif)
                assert np.array_equal(eng.cluster_marginals(0, table, available, ptemp), got)
                # ClusterJump: members of cluster k staying / jumping to the next cluster
                if K > 1 and groups[0][k].any():
                    members = np.flatnonzero(groups[0][k])
                    got = eng.jump_lh_resident(0, k, (k + 1) % K, members, temp, ptemp)
                    if not underflows:
                        want = fake.jump_lh_resident(0, k, (k + 1) % K, members, temp, ptemp)
                        np.testing.assert_allclose(got, want, rtol=1e-9 if ptemp == 1.0 else 2e-6, atol=1e-9)
                    assert np.isfinite(got).all()
        # per-group collapsed values and the per-object source prior from the resident state
        for c in range(len(groups)):
            np.testing.assert_allclose(eng.collapsed_loglik(0, c), fake.collapsed_loglik(0, c), rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(eng.source_prior(0), fake.source_prior(0), rtol=2e-6, atol=1e-6)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide", "long", "five", "eight"])
def test_fused_tables_same_bits(name):
    """SBE_OPT_FUSE_TABLES (round 4): the one-launch forms -- table entries built inside the consuming
    kernel -- return the bits of the table-kernel-in-front forms, for every resident operator call that has both."""
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(23)
        N, F, C = source.shape
        K = groups[0].shape[0]
        has = np.stack([g.any(axis=0) for g in groups], axis=1)                       # [N, C]

        def both(call):
            eng.set_option(fuse_tables=True)
            a = call()
            eng.set_option(fuse_tables=False)
            b = call()
            eng.set_option(fuse_tables=True)
            return a, b

        for temp, ptemp in ((1.0, 1.0), (1.3, 1.5)):
            for k in range(min(K, 3)):
                available = np.flatnonzero(~groups[0].any(axis=0) | groups[0][k])
                a, b = both(lambda: eng.cluster_posterior_marginals(0, k, available, temp, ptemp))
                assert np.array_equal(a, b), (name, "marginals", k, temp)
                if K > 1 and groups[0][k].any():
                    members = np.flatnonzero(groups[0][k])
                    a, b = both(lambda: eng.jump_lh_resident(0, k, (k + 1) % K, members, temp, ptemp))
                    assert np.array_equal(a, b), (name, "jump", k, temp)
                for n in (1, 9, min(N, 300)):
                    objs = np.sort(rng.choice(N, size=min(n, N), replace=False))
                    a, b = both(lambda: eng.given_unchanged_lh(0, k, objs, temp, ptemp))
                    assert np.array_equal(a, b), (name, "given_unchanged_lh", k, n, temp)
                    hc_new = has[objs].copy()
                    hc_new[:, 0] = rng.random(objs.size) < 0.5
                    hc_new[~hc_new.any(axis=1), 1 if C > 1 else 0] = True
                    hc_old = has[objs].copy()
                    hc_old[~hc_old.any(axis=1), 1 if C > 1 else 0] = True
                    src_old = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
                    z = rng.random((objs.size, F))
                    for from_prior in (False, True):
                        try:
                            a = b = None
                            a, b = both(lambda: eng.given_unchanged_gibbs(0, k, objs, hc_new, hc_old, src_old, z, temp, ptemp, from_prior))
                        except Exception as exc:                     # (a random pattern may leave an observation no component:
                            assert "normalize" in str(exc) or "sum" in str(exc), exc           # both forms raise it alike)
                            eng.set_option(fuse_tables=True)
                            continue
                        for u, v in zip(a, b):
                            assert np.array_equal(u, v), (name, "given_unchanged_gibbs", k, n, temp, from_prior)
                        if temp == 1.0 and ptemp == 1.0:             # ... and both are the oracle-backed double's bits
                            want = fake.given_unchanged_gibbs(0, k, objs, hc_new, hc_old, src_old, z, temp, ptemp, from_prior)
                            for u, v in zip(a, want):
                                assert np.array_equal(u, v), (name, "given_unchanged_gibbs vs oracle", k, n, from_prior)
        # a count row that sums to nothing with a zero concentration: normalize's assert fires in both forms
        if name == "cfg1_fixture":
            conc0 = np.array(fake.conc[0], dtype=np.float64, copy=True)
            conc0[..., 0, :] = 0.0
            zero = np.zeros_like(counts[0])
            eng.set_concentration(0, conc0)
            eng.set_counts(0, 0, zero)
            for fuse in (True, False):
                eng.set_option(fuse_tables=fuse)
                with pytest.raises(Exception, match="(?i)normali"):
                    eng.cluster_posterior_marginals(0, 0, np.arange(5), 1.0, 1.0)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide", "long", "eight"])
def test_count_rows_with_their_probability_rows(name):
    """set_counts_rows(update_probs=True) = set_counts_rows + update_probs of the touched components: the same counts,
    the same probability tables bit for bit (and so the same likelihoods), one launch; refused while the tables of a
    touched component do not exist."""
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(31)
        C = len(groups)
        off = eng.group_offsets
        eng.copy_slot(1, 0)
        idx, rows = [], []
        for c in range(C):
            for g in rng.choice(groups[c].shape[0], size=min(2, groups[c].shape[0]), replace=False):
                idx.append(off[c] + int(g))
                rows.append(rng.integers(0, 40, size=counts[c][g].shape).astype(np.float32))
        idx, rows = np.asarray(idx, dtype=np.int32), np.stack(rows)
        eng.set_counts_rows(0, idx, rows, update_probs=True)
        eng.set_counts_rows(1, idx, rows)
        eng.update_probs(1, range(C))
        for c in range(C):
            assert np.array_equal(eng.get_counts(0, c), eng.get_counts(1, c))
            assert np.array_equal(eng.get_probs(0, c), eng.get_probs(1, c)), (name, c)
        assert eng.mixture_loglik(0) == eng.mixture_loglik(1)                    # (the tile-transposed copies agree too)
        assert np.array_equal(eng.likelihood_per_component(0), eng.likelihood_per_component(1))
        # a row that normalises to nothing is reported like update_probs reports it
        if name == "cfg1_fixture":
            conc0 = np.array(fake.conc[0], dtype=np.float64, copy=True)
            conc0[..., 0, :] = 0.0
            eng.set_concentration(0, conc0)
            eng.update_probs(0, 0)
            with pytest.raises(Exception, match="(?i)normali"):
                eng.set_counts_rows(0, [0], np.zeros((1,) + counts[0].shape[1:], dtype=np.float32), update_probs=True)
                eng.sync()
    finally:
        eng.close()
    # tables that were never built cannot be patched row-wise
    feats, groups, conc, weights, source, counts, unif = _workload("cfg1_fixture")
    with Engine(feats, [g.shape[0] for g in groups], n_slots=1) as fresh:
        for c in range(len(groups)):
            fresh.set_concentration(c, conc[c])
            fresh.set_counts(0, c, counts[c])
        with pytest.raises(Exception, match="sbe_update_probs first"):
            fresh.set_counts_rows(0, [0], counts[0][:1], update_probs=True)
        fresh.set_counts_rows(0, [0], counts[0][:1])                               # (the plain patch needs no tables)


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide", "five"])
def test_gibbs_propose_in_one_call(name):
    """sbe_gibbs_propose (GibbsSampleSource._propose, operators.py:495-552, in one engine call) against the oracle-backed
    double, which composes it from the call-by-call pieces: drawn ids, touched groups and count rows always bit for bit,
    the selected probabilities bit for bit at temperature 1; and the candidate slot it leaves behind -- source rows, counts,
    tables -- is the double's."""
    eng, fake, groups, source, counts = _pair(name)
    try:
        if not eng.gibbs_propose_supported():
            pytest.skip("tables beyond the fused table kernel of the chain")
        rng = np.random.default_rng(41)
        N, F, C = source.shape
        everyone = np.arange(N, dtype=np.int32)
        for temp, ptemp, from_prior in ((1.0, 1.0, False), (1.0, 1.0, True), (1.4, 1.2, False)):
            for n in (1, 7, min(N, 200)):
                objs = np.sort(rng.choice(N, size=min(n, N), replace=False)).astype(np.int32)
                z = rng.random((objs.size, F))
                want = fake.gibbs_propose(0, 1, objs, z, temp, ptemp, from_prior)
                tags = ("ids", "sel", "sel_back", "touched", "rows")
                results = []
                for tile_form in (True, False):              # one kernel per feature tile / the chain that builds the candidate slot
                    eng.set_option(fuse_tables=tile_form)
                    got = eng.gibbs_propose(0, 1, objs, z, temp, ptemp, from_prior)
                    results.append(got)
                    for tag, g, w in zip(tags, got, want):
                        assert g.shape == w.shape and g.dtype == w.dtype, (name, tag, g.dtype, w.dtype)
                        if tag in ("ids", "touched", "rows") or (temp == 1.0 and ptemp == 1.0):
                            assert np.array_equal(g, w), (name, tag, n, temp, from_prior, tile_form)
                        else:
                            np.testing.assert_allclose(g, w, rtol=2e-6, atol=1e-7)
                    # the current slot is untouched
                    assert np.array_equal(eng.get_source_rows(0, everyone), source)
                for a, b in zip(*results):
                    assert np.array_equal(a, b), (name, n, temp, from_prior, "tile form != chain form")
                # the chain form (run last) leaves the candidate in slot 1: source rows, counts, tables are the double's
                assert np.array_equal(eng.get_source_rows(1, everyone), fake.get_source_rows(1, everyone))
                for c in range(C):
                    assert np.array_equal(eng.get_counts(1, c), fake._slot(1)["counts"][c]), (name, c)
                assert np.array_equal(eng.likelihood_per_component(1), fake._state(1)[2])      # (the candidate's tables)
                # ... and with follow=True the CURRENT slot takes the proposal (sbe_gibbs_propose_apply), in both forms: the same
                # five arrays, and the slot ends as the candidate -- source rows, counts, tables (and their tile-transposed copy)
                for tile_form in (True, False):
                    eng.set_option(fuse_tables=tile_form)
                    eng.copy_slot(2, 0)
                    got_f = eng.gibbs_propose(2, 1, objs, z, temp, ptemp, from_prior, follow=True)
                    for a, b in zip(got_f, results[0]):
                        assert np.array_equal(a, b), (name, n, temp, from_prior, tile_form, "with the slot following")
                    if got_f[3].size:
                        assert np.array_equal(eng.get_source_rows(2, everyone), fake.get_source_rows(1, everyone)), (name, n, tile_form)
                        for c in range(C):
                            assert np.array_equal(eng.get_counts(2, c), fake._slot(1)["counts"][c]), (name, c, tile_form)
                        assert np.array_equal(eng.likelihood_per_component(2), fake._state(1)[2]), (name, n, tile_form)
                        eng.set_option(fuse_tables=False)
                        eng.gibbs_propose(0, 1, objs, z, temp, ptemp, from_prior)               # (the chain form's candidate in slot 1)
                        assert eng.mixture_loglik(2) == eng.mixture_loglik(1), (name, n, tile_form)
        eng.set_option(fuse_tables=True)
        with pytest.raises(Exception, match="differ"):
            eng.gibbs_propose(0, 0, [0], np.zeros((1, F)))
        with pytest.raises(Exception, match="out of range"):
            eng.gibbs_propose(0, 1, [N], np.zeros((1, F)))
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide", "long", "five"])
def test_cluster_gibbs_with_its_count_delta(name):
    """sbe_given_unchanged_gibbs_counts: the cluster operators' source resampling AND the count delta of the proposal
    (update_feature_counts, counts.py:55-95) from one launch -- the draw's three arrays are the plain call's, touched groups
    and count rows are counts_delta's for the ids that were drawn (device and oracle-backed double), with the subset's
    cluster membership changed between the two samples."""
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(53)
        N, F, C = source.shape
        K = groups[0].shape[0]
        off = eng.group_offsets
        has = np.stack([g.any(axis=0) for g in groups], axis=1)
        for fuse in (True, False):
            eng.set_option(fuse_tables=fuse)
            for n in (1, 11, min(N, 120)):
                objs = np.sort(rng.choice(N, size=min(n, N), replace=False)).astype(np.int32)
                k = int(rng.integers(0, K))
                old_clusters = groups[0].copy()                           # the OLD sample: the subset's membership differs
                old_clusters[:, objs] = False
                move = rng.integers(0, K + 1, size=objs.size)
                old_clusters[move[move < K], objs[move < K]] = True
                gid_new = np.stack([_ids(groups[c], objs, off[c]) for c in range(C)])
                gid_old = gid_new.copy()
                gid_old[0] = _ids(old_clusters, objs, 0)
                hc_new = has[objs].copy()
                hc_old = hc_new.copy()
                hc_old[:, 0] = old_clusters[:, objs].any(axis=0)
                if C > 1:
                    hc_new[:, 1] = hc_old[:, 1] = True
                src_old = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
                src_old[(src_old == 0) & ~hc_old[:, [0]].repeat(F, 1)] = 1 if C > 1 else 255       # (no cluster source without a cluster)
                z = rng.random((objs.size, F))
                try:
                    plain = eng.given_unchanged_gibbs(0, k, objs, hc_new, hc_old, src_old, z)
                except Exception as exc:
                    assert "normalize" in str(exc), exc
                    continue
                got = eng.given_unchanged_gibbs(0, k, objs, hc_new, hc_old, src_old, z, gid_old=gid_old, gid_new=gid_new)
                assert len(got) == 5
                for a, b in zip(got[:3], plain):
                    assert np.array_equal(a, b), (name, fuse, n, "draw differs with the count delta asked for")
                t_want, r_want = eng.counts_delta(objs, gid_old, gid_new, src_old, got[0])
                assert np.array_equal(got[3], t_want) and got[4].dtype == np.float32 and np.array_equal(got[4], r_want), (name, fuse, n)
                want = fake.given_unchanged_gibbs(0, k, objs, hc_new, hc_old, src_old, z, gid_old=gid_old, gid_new=gid_new)
                for a, b, what in zip(got, want, ("ids", "sel", "back", "touched", "rows")):
                    assert np.array_equal(a, b), (name, fuse, n, what)
                # ... and the slot FOLLOWS the proposal (sbe_given_unchanged_gibbs_apply): the same five arrays, and behind them the
                # slot's counts, probability rows and source rows are what the explicit patches of the same data leave
                eng.copy_slot(1, 0)
                eng.copy_slot(2, 0)
                comp_of = np.searchsorted(off, got[3], side="right") - 1
                new_rows = (np.stack([counts[c][g - off[c]] for g, c in zip(got[3], comp_of)]) + got[4]) if got[3].size else None
                if new_rows is not None and (new_rows < 0).any():
                    continue                                          # (this test's made-up old state took counts that are not there)
                fol = eng.given_unchanged_gibbs(1, k, objs, hc_new, hc_old, src_old, z, gid_old=gid_old, gid_new=gid_new, follow=True,
                                                update_probs=True)
                for a, b, what in zip(fol, got, ("ids", "sel", "back", "touched", "rows")):
                    assert np.array_equal(a, b), (name, fuse, n, what, "with the slot following")
                if got[3].size:
                    eng.set_counts_rows(2, got[3], new_rows, update_probs=True)
                    eng.set_source_rows(2, objs, got[0][..., None] == np.arange(C, dtype=np.uint8))
                for c in range(C):
                    assert np.array_equal(eng.get_counts(1, c), eng.get_counts(2, c)), (name, fuse, n, c, "following counts")
                    assert np.array_equal(eng.get_probs(1, c), eng.get_probs(2, c)), (name, fuse, n, c, "following probability rows")
                assert np.array_equal(eng.get_source_rows(1, objs), eng.get_source_rows(2, objs)), (name, fuse, n, "following source rows")
    finally:
        eng.close()


def test_argument_checks():
    eng, fake, groups, source, counts = _pair("cfg1_fixture")
    try:
        with pytest.raises(Exception, match="out of range"):
            eng.given_unchanged_lh(0, 99, [0])
        with pytest.raises(Exception, match="out of range"):
            eng.cluster_posterior_marginals(0, 0, [10 ** 6])
        with pytest.raises(Exception, match="out of range"):
            eng.set_counts_rows(0, [10 ** 6], np.zeros((1,) + counts[0].shape[1:], dtype=np.float32))
        touched, diff = eng.counts_delta(np.zeros(0, dtype=np.int32), np.zeros((len(groups), 0)), np.zeros((len(groups), 0)),
                                         np.zeros((0, source.shape[1])), np.zeros((0, source.shape[1])))
        assert touched.size == 0 and diff.shape[0] == 0
        # a following slot must hold the tables the difference is added to (slot 2 was never set)
        C, F = len(groups), source.shape[1]
        ids = np.full((C, 1), -1, dtype=np.int32)
        ids[0, 0] = 0
        src = np.zeros((1, F), dtype=np.uint8)
        with pytest.raises(Exception, match="counts of component 0 not set"):
            eng.counts_delta([0], ids, ids, src, src, follow_slot=2)
        eng.set_counts(2, 0, counts[0])                                   # (counts, but no probability tables yet)
        with pytest.raises(Exception, match="sbe_update_probs first"):
            eng.counts_delta([0], ids, ids, src, src, follow_slot=2, update_probs=True)
        eng.counts_delta([0], ids, ids, src, src, follow_slot=2)
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["cfg1_fixture", "south_america", "headline", "wide"])
@pytest.mark.parametrize("from_prior", [False, True], ids=["posterior", "from_prior"])
def test_given_unchanged_gibbs_against_the_double(name, from_prior):
    """sbe_given_unchanged_gibbs (ClusterOperator.gibbs_sample_source, operators.py:796-851) against the oracle-backed double's
    restatement of the reference expressions: at T = T_prior = 1 the drawn components, p[drawn] and p_back[old source] are
    bit-identical; tempered, the draws may differ only where a uniform falls within float32 powf distance of a cdf step
    (none at these seeds) and the probabilities agree to 2e-6."""
    eng, fake, groups, source, counts = _pair(name)
    try:
        rng = np.random.default_rng(17)
        N, F, C = source.shape
        K = groups[0].shape[0]
        for trial in range(6):
            i_cluster = int(rng.integers(0, K))
            n = int(rng.integers(1, min(N, 40)))
            objs = np.sort(rng.choice(N, size=n, replace=False)).astype(np.int32)
            hc_new = np.stack([g[:, objs].any(axis=0) for g in groups], axis=1)
            hc_old = hc_new.copy()
            flip = rng.random(n) < 0.5                                   # the objects the proposal moved: their cluster bit differs
            hc_old[flip, 0] = ~hc_old[flip, 0]
            src_old = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
            z = rng.random((n, F))
            for t, tp in ((1.0, 1.0), (2.5, 1.7)):
                got = eng.given_unchanged_gibbs(0, i_cluster, objs, hc_new, hc_old, src_old, z, t, tp, from_prior)
                want = fake.given_unchanged_gibbs(0, i_cluster, objs, hc_new, hc_old, src_old, z, t, tp, from_prior)
                assert got[0].dtype == np.uint8 and got[1].dtype == np.float32 and got[2].dtype == np.float32
                assert np.array_equal(got[0], want[0]), (name, trial, t)
                if t == 1.0 and tp == 1.0:
                    assert np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2]), (name, trial)
                else:
                    np.testing.assert_allclose(got[1], want[1], rtol=2e-6, atol=1e-7)
                    np.testing.assert_allclose(got[2], want[2], rtol=2e-6, atol=1e-7)
        with pytest.raises(Exception, match="out of range"):
            eng.given_unchanged_gibbs(0, 99, [0], hc_new[:1], hc_old[:1], src_old[:1], z[:1])
        bad = src_old[:1].copy()
        bad[0, 0] = C                                                    # neither a component nor 255
        with pytest.raises(Exception, match="old source component"):
            eng.given_unchanged_gibbs(0, 0, objs[:1], hc_new[:1], hc_old[:1], bad, z[:1])
    finally:
        eng.close()

