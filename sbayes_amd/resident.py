"""Resident MCMC-step flow: the sample state of a chain lives in engine
slots; a step ships only the proposed delta (changed cluster rows, changed source rows, weights)
across PCIe, and counts (a9), probability tables (a4), the collapsed likelihood (a7/a8) and the
mixture log-likelihood (8(d)) are all recomputed on the device from resident data.

    chain = ResidentChain(model, sample)            # uploads the state once (slot `cur`)
    cand = chain.propose(clusters=new_clusters, source_rows=(objects, rows), weights=None)
    ll = cand.collapsed_loglik()                    # Likelihood.__call__ value, device-resident
    mix = cand.mixture_loglik()                     # fused kernel
    chain.accept()  /  chain.reject()               # slot swap, no copy on reject

The accept/reject decision, proposal logic and RNG stay with the (reference) sampler."""
from __future__ import annotations

from contextlib import contextmanager

import numpy as np

from .conditionals import _engine


class _SlotView:
    def __init__(self, chain, slot):
        self.chain, self.slot = chain, slot

    def collapsed_loglik(self) -> float:
        """Likelihood.__call__(sample, caching=False) value from the slot's resident counts."""
        eng = self.chain.eng
        per_group = eng.collapsed_loglik_all(self.slot)
        off = np.concatenate([[0], np.cumsum(eng.n_groups)])
        return float(sum(per_group[off[c]:off[c + 1]].sum() for c in range(eng.n_components)))

    def collapsed_group_logliks(self):
        eng = self.chain.eng
        return [eng.collapsed_loglik(self.slot, c) for c in range(eng.n_components)]

    def mixture_loglik(self) -> float:
        eng = self.chain.eng
        for c in self.chain._probs_dirty[self.slot]:
            eng.update_probs(self.slot, c)
        self.chain._probs_dirty[self.slot] = set()
        return eng.mixture_loglik(self.slot)

    def counts(self, component):
        return self.chain.eng.get_counts(self.slot, component)


class ResidentChain:
    """Two slots per chain: `cur` (accepted state) and `cand` (proposal under evaluation)."""

    def __init__(self, model, sample, slots=None):
        self.model = model
        self.eng = _engine(model)
        if slots is None:
            # the engine is shared through the registry with the drop-in operator forms, which bind samples into
            # slot 0 / slots (0, 1) by default: the chain's resident state lives in the TOP two slots
            if self.eng.n_slots < 4:
                raise ValueError(f"engine has {self.eng.n_slots} slots; a ResidentChain next to the drop-in forms needs 4")
            slots = (self.eng.n_slots - 2, self.eng.n_slots - 1)
        if max(slots) >= self.eng.n_slots or min(slots) < 0 or slots[0] == slots[1]:
            raise ValueError(f"engine has {self.eng.n_slots} slots, need two distinct slots, got {slots}")
        self.cur, self.cand = slots
        self.names = list(sample.component_names)
        with self._deferred_checks():
            self._upload(model, sample)

    @contextmanager
    def _deferred_checks(self):
        """Inside: state-setting calls never stall the stream (their data checks are queued).  On exit the option is
        restored, which synchronises and raises a queued check HERE -- in the chain call that caused it, not in some
        later call of another user of the shared engine."""
        previous = getattr(self.eng, "_deferred", False)       # the engine is shared through the registry: another user
        self.eng.set_option(deferred_checks=True)               # (bench, an outer chain call) may have turned it on
        try:
            yield
        finally:
            if not previous:
                self.eng.set_option(deferred_checks=False)

    def _upload(self, model, sample):
        eng = self.eng
        conc = [np.asarray(model.prior.prior_cluster_effect.concentration_array)] + [
            np.asarray(model.prior.prior_confounding_effects[k].concentration_array(sample)) for k in self.names[1:]]
        groups = [sample.clusters.value] + [c.group_assignment for c in sample.confounders.values()]
        for c in range(eng.n_components):
            eng.set_concentration(c, conc[c])
            eng.set_groups(self.cur, c, groups[c])
        eng.set_source(self.cur, sample.source.value)
        eng.recount(self.cur)
        eng.set_weights(self.cur, sample.weights.value)
        self._clusters = sample.clusters.value.copy()
        self._cand_clusters = None
        self._probs_dirty = {self.cur: set(range(eng.n_components)), self.cand: set()}
        self.changed_groups = None
        self._pending = False

    @property
    def current(self):
        return _SlotView(self, self.cur)

    def propose(self, clusters=None, source_rows=None, weights=None):
        """Build the candidate state in the `cand` slot from the current one plus a delta.
        clusters: full bool [K, N] of the candidate (only N*2 bytes of ids cross PCIe);
        source_rows: (object indices, bool rows [n, F, C]) of the objects whose source changed;
        weights: float32 [F, C] or None.  Counts are delta-updated on the device over the union of
        the objects whose cluster membership or source changed."""
        with self._deferred_checks():
            return self._propose(clusters, source_rows, weights)

    def _propose(self, clusters, source_rows, weights):
        eng = self.eng
        eng.copy_slot(self.cand, self.cur)
        self._probs_dirty[self.cand] = set(self._probs_dirty[self.cur])
        moved = np.zeros(eng.n_objects, dtype=bool)
        self._cand_clusters = self._clusters
        if clusters is not None:
            clusters = np.asarray(clusters, dtype=bool)
            moved |= (clusters != self._clusters).any(axis=0)
            eng.set_groups(self.cand, 0, clusters)
            self._cand_clusters = clusters.copy()
        if source_rows is not None:
            objects, rows = source_rows
            objects = np.asarray(objects)
            if objects.size:
                eng.set_source_rows(self.cand, objects, rows)
                moved[objects] = True
        if weights is not None:
            eng.set_weights(self.cand, weights)
        subset = np.flatnonzero(moved)
        self.changed_groups = eng.update_counts(self.cand, self.cur, subset) if subset.size else \
            np.zeros(eng.n_groups_total, dtype=bool)
        off = eng.group_offsets
        for c in range(eng.n_components):
            if self.changed_groups[off[c]:off[c + 1]].any():
                self._probs_dirty[self.cand].add(c)
        self._pending = True
        return _SlotView(self, self.cand)

    def propose_gibbs_source(self, objects, temperature=1.0, prior_temperature=1.0, sample_from_prior=False,
                             z=None, device_rng=False):
        """GibbsSampleSource._propose (operators.py:495-552) on the resident state: the source assignments of the
        listed objects are redrawn from their posterior ON THE DEVICE into the `cand` slot, counts and tables follow
        there, and both transition log-probabilities come back -- only the object ids (and, unless
        device_rng, the uniforms) cross PCIe; nothing of the sample state is re-uploaded (the drop-in operator
        form, operators.gibbs_sample_source, rebinds the whole sample on every call).
        z: uniforms [n, F] (default: np.random.random([n, F, 1]) drawn exactly where the reference draws them);
        device_rng=True: the engine's Philox stream instead.  Returns (candidate view, log_q, log_q_back);
        follow with accept() or reject()."""
        with self._deferred_checks():
            return self._propose_gibbs_source(objects, temperature, prior_temperature, sample_from_prior, z, device_rng)

    def _propose_gibbs_source(self, objects, temperature, prior_temperature, sample_from_prior, z, device_rng):
        eng = self.eng
        objects = np.asarray(objects)
        if objects.dtype == np.bool_:
            objects = np.flatnonzero(objects)
        objects = np.ascontiguousarray(objects, dtype=np.int32)
        for c in self._probs_dirty[self.cur]:               # the posterior needs the current tables
            eng.update_probs(self.cur, c)
        self._probs_dirty[self.cur] = set()
        if device_rng:
            z = None
        elif z is None:
            z = np.random.random([objects.size, eng.n_features, 1])
        eng.copy_slot(self.cand, self.cur)
        log_q = eng.sample_source(self.cur, self.cand, objects, z, temperature, prior_temperature, sample_from_prior)
        self.changed_groups = eng.update_counts(self.cand, self.cur, objects) if objects.size else \
            np.zeros(eng.n_groups_total, dtype=bool)
        off = eng.group_offsets
        for c in range(eng.n_components):
            if self.changed_groups[off[c]:off[c + 1]].any():
                eng.update_probs(self.cand, c)               # log_q_back is evaluated against the NEW tables
        self._probs_dirty[self.cand] = set()
        log_q_back = eng.source_logprob(self.cand, self.cur, objects, temperature, prior_temperature, sample_from_prior)
        self._cand_clusters = self._clusters
        self._pending = True
        return _SlotView(self, self.cand), log_q, log_q_back

    def gibbs_step(self, objects, temperature=1.0, prior_temperature=1.0, sample_from_prior=False, z=None,
                   device_rng=False):
        """propose_gibbs_source + evaluation of the candidate in ONE engine call (sbe_gibbs_step: five launches, one
        synchronisation).  Returns (log_q, log_q_back, collapsed log-likelihood, per-group values, mixture
        log-likelihood).  Follow with accept() or reject()."""
        eng = self.eng
        objects = np.asarray(objects)
        if objects.dtype == np.bool_:
            objects = np.flatnonzero(objects)
        for c in self._probs_dirty[self.cur]:
            eng.update_probs(self.cur, c)
        self._probs_dirty[self.cur] = set()
        if device_rng:
            z = None
        elif z is None:
            z = np.random.random([objects.size, eng.n_features, 1])
        log_q, log_q_back, glh, mix, changed = eng.gibbs_step(self.cur, self.cand, objects, z, temperature,
                                                              prior_temperature, sample_from_prior)
        self.changed_groups = changed
        self._probs_dirty[self.cand] = set()
        self._cand_clusters = self._clusters
        self._pending = True
        return log_q, log_q_back, float(glh.sum()), glh, mix

    def accept(self):
        if not self._pending:
            raise RuntimeError("no pending proposal")
        self.cur, self.cand = self.cand, self.cur
        if self._cand_clusters is not None:
            self._clusters = self._cand_clusters
        elif getattr(self, "_cand_delta", None) is not None:        # step_delta: the accepted moves into the host mirror
            mo, mc = self._cand_delta
            self._clusters = self._clusters.copy()
            self._clusters[:, mo] = False
            self._clusters[mc[mc >= 0], mo[mc >= 0]] = True
        self._cand_delta = None
        self._pending = False

    def reject(self):
        self._pending = False


    def step(self, clusters=None, source_rows=None, weights=None):
        """propose + evaluate in ONE engine call (sbe_step): returns (collapsed log-likelihood, per-group
        values, mixture log-likelihood).  Follow with accept() or reject()."""
        objs, rows = (None, None) if source_rows is None else source_rows
        glh, mix, changed = self.eng.step(self.cur, self.cand, clusters=clusters, changed_objects=objs,
                                          source_rows=rows, weights=weights)
        self._cand_clusters = np.asarray(clusters, dtype=bool).copy() if clusters is not None else self._clusters
        self.changed_groups = changed
        self._probs_dirty[self.cand] = set()
        self._pending = True
        return float(glh.sum()), glh, mix


    def step_delta(self, moved_objects=None, moved_cluster=None, source_rows=None, weights=None):
        """step() with the proposal in delta form (sbe_step_delta): the objects that change cluster with their new
        cluster index (-1: none) instead of the full cluster matrix.  Follow with accept() or reject()."""
        objs, rows = (None, None) if source_rows is None else source_rows
        glh, mix, changed = self.eng.step_delta(self.cur, self.cand, moved_objects, moved_cluster, objs, rows, weights)
        self._cand_clusters = None                   # (built from the delta on accept: nothing [K, N]-sized per step)
        self._cand_delta = None if moved_objects is None or not len(moved_objects) else \
            (np.array(moved_objects, dtype=np.int64), np.array(moved_cluster, dtype=np.int64))
        self.changed_groups = changed
        self._probs_dirty[self.cand] = set()
        self._pending = True
        return float(glh.sum()), glh, mix


class ResidentChainBatch:
    """B independent chains resident on one engine, stepped together (sbe_step_batch): the reference's
    `for c in chains: self.step(...)` (MCMC.generate_samples, sbayes/sampling/mcmc.py:237-241) as ONE engine call per
    sweep -- one candidate-building launch, one fused mixture launch over the B candidates, one reduction launch,
    one synchronisation.  Chain i owns the slots (2i, 2i+1): current and candidate, swapped on accept.

        batch = ResidentChainBatch(model, [sample_0, ..., sample_{B-1}])
        ll, group_lh, mix = batch.step(clusters=[...], source_rows=[...], weights=[...])   # per chain, None = unchanged
        batch.accept(mask)                                     # bool [B]: accepted proposals swap their slots

    Proposal logic, RNG and the accept / reject decisions stay with the sampler."""

    def __init__(self, model, samples, device=None):
        from .engine import Engine
        from .registry import default_device
        self.model = model
        self.n = len(samples)
        first = samples[0]
        self.names = list(first.component_names)
        feats = model.data.features.values
        n_groups = [model.shapes.n_clusters] + [c.n_groups for c in model.data.confounders.values()]
        self.eng = eng = Engine(feats, n_groups, n_slots=2 * self.n, device=default_device() if device is None else device)
        conc = [np.asarray(model.prior.prior_cluster_effect.concentration_array)] + [
            np.asarray(model.prior.prior_confounding_effects[k].concentration_array(first)) for k in self.names[1:]]
        for c in range(eng.n_components):
            eng.set_concentration(c, conc[c])
        eng.set_option(deferred_checks=True)
        self.cur = np.arange(0, 2 * self.n, 2, dtype=np.int32)
        self.cand = self.cur + 1
        clusters0 = []
        for i, sample in enumerate(samples):
            slot = int(self.cur[i])
            groups = [sample.clusters.value] + [c.group_assignment for c in sample.confounders.values()]
            for c in range(eng.n_components):
                eng.set_groups(slot, c, groups[c])
            eng.set_source(slot, sample.source.value)
            eng.recount(slot)
            for c in range(eng.n_components):
                eng.update_probs(slot, c)
            eng.set_weights(slot, sample.weights.value)
            clusters0.append(np.array(sample.clusters.value, dtype=bool))
        eng.set_option(deferred_checks=False)                   # (synchronises: a queued data check is raised here)
        # host mirror of the chains' cluster matrices, stacked [B, K, N]; the candidates are kept as a reference to the
        # caller's stacked array (+ mask) until accept() -- no per-chain Python work on the step path
        self._clusters = np.stack(clusters0) if clusters0 else np.zeros((0, n_groups[0], feats.shape[0]), dtype=bool)
        self._cand = None                                        # private copy of the last step's candidate clusters
        self.changed_groups = None
        self._pending = False

    def close(self):
        self.eng.close()

    def step(self, clusters=None, source_rows=None, weights=None):
        """Per-chain deltas as lists of length B (entries None = unchanged): clusters bool [K, N]; source_rows
        (object indices, bool rows [n, F, C]); weights float32 [F, C].  Returns (collapsed log-likelihood [B],
        per-group values [B, G_total], mixture log-likelihood [B]) of the B candidates."""
        eng, n = self.eng, self.n
        cl = cm = w = wm = None
        if clusters is not None and any(c is not None for c in clusters):
            cm = np.array([c is not None for c in clusters])
            cl = np.stack([np.asarray(c, dtype=bool) if c is not None else self._clusters[i] for i, c in enumerate(clusters)])
        if weights is not None and any(x is not None for x in weights):
            wm = np.array([x is not None for x in weights])
            zero = np.zeros((eng.n_features, eng.n_components), dtype=np.float32)
            w = np.stack([np.asarray(x, dtype=np.float32) if x is not None else zero for x in weights])
        ptr = objs = rows = None
        if source_rows is not None and any(r is not None and len(r[0]) for r in source_rows):
            counts = [0 if r is None else len(r[0]) for r in source_rows]
            ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
            objs = np.concatenate([np.asarray(r[0], dtype=np.int32) for r in source_rows if r is not None and len(r[0])])
            rows = np.concatenate([np.asarray(r[1], dtype=bool) for r in source_rows if r is not None and len(r[0])])
        return self.step_arrays(cl, cm, ptr, objs, rows, w, wm)

    def step_arrays(self, clusters=None, clusters_mask=None, rows_ptr=None, changed_objects=None, source_rows=None,
                    weights=None, weights_mask=None):
        """The same with the deltas already stacked (Engine.step_batch's arguments): no per-chain Python work."""
        self._cand, self.changed_groups, self._pending = None, None, False     # nothing of an earlier step survives a
        glh, mix, changed = self.eng.step_batch(self.cur, self.cand, clusters, clusters_mask, rows_ptr, changed_objects,   # failed call
                                                source_rows, weights, weights_mask)
        if clusters is not None:
            # the candidates' cluster rows are COPIED here (masked chains only): the caller may reuse its stacked array
            # in place before accept()
            cm = None if clusters_mask is None else np.array(clusters_mask, dtype=bool)
            cl = np.asarray(clusters)
            self._cand = (np.array(cl, dtype=bool) if cm is None else (np.flatnonzero(cm), np.array(cl[cm], dtype=bool)), cm)
        self.changed_groups = changed
        self._pending = True
        return glh.sum(axis=1), glh, mix

    def step_delta(self, moved_ptr, moved_objects, moved_cluster, rows_ptr=None, changed_objects=None, source_rows=None,
                   weights=None, weights_mask=None):
        """One step per chain with the proposals in DELTA form (sbe_step_batch_delta): per chain the objects that change
        cluster and their new cluster (-1: none), CSR by moved_ptr, and the objects whose source rows change (each once).
        What an operator produces goes in as it is: no [K, N] cluster matrices are built, stacked, scanned or sent, and
        the library patches the candidates in O(delta).  Same return values as step_arrays; follow with accept(mask)."""
        self._cand, self.changed_groups, self._pending = None, None, False
        glh, mix, changed = self.eng.step_batch_delta(self.cur, self.cand, moved_ptr, moved_objects, moved_cluster, rows_ptr,
                                                      changed_objects, source_rows, weights, weights_mask)
        self._cand = ("delta", np.array(moved_ptr, dtype=np.int64), np.array(moved_objects, dtype=np.int64),
                      np.array(moved_cluster, dtype=np.int64))
        self.changed_groups = changed
        self._pending = True
        return glh.sum(axis=1), glh, mix

    def accept(self, mask=None):
        """Swap the slots of the accepted chains (all of them by default); rejected chains keep their current slot."""
        if not self._pending:
            raise RuntimeError("no pending step: accept() follows exactly one step() / step_arrays()")
        mask = np.ones(self.n, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        cur, cand = self.cur.copy(), self.cand.copy()
        self.cur = np.where(mask, cand, cur).astype(np.int32)
        self.cand = np.where(mask, cur, cand).astype(np.int32)
        if self._cand is not None and isinstance(self._cand[0], str):      # ("delta", ...): the accepted chains' moves into the host mirror
            _tag, mp, mo, mc = self._cand
            if mo.size:                                                    # (vectorised over every chain's moves)
                chain = np.repeat(np.arange(self.n), np.diff(mp))
                keep = mask[chain]
                self._clusters[chain[keep], :, mo[keep]] = False
                inside = keep & (mc >= 0)
                self._clusters[chain[inside], mc[inside], mo[inside]] = True
            self._cand = None
        elif self._cand is not None:                             # the accepted chains' candidate clusters become current
            cl, cm = self._cand
            if cm is None:
                self._clusters[mask] = cl[mask]
            else:
                idx, rows = cl
                keep = mask[idx]
                self._clusters[idx[keep]] = rows[keep]
            self._cand = None
        self._pending = False

    def reject(self):
        """Drop the pending step of every chain (same as accept(all False))."""
        self._cand, self._pending = None, False

    def counts(self, chain, component):
        return self.eng.get_counts(int(self.cur[chain]), component)
