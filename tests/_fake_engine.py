"""Test double for sbayes_amd.engine.Engine built on the CPU oracle -- TEST INFRASTRUCTURE.

Lets the `-m "not gpu"` suite exercise the HOST LOGIC of the drop-in layer (cache nodes,
`what_changed`, partial updates, copy-on-write samples, pickling) without a GPU.  It lives under
tests/ and is injected by monkeypatching `registry.get_engine`; the product never sees it."""
import numpy as np

from oracle import sbayes_oracle as orc


class FakeEngine:
    def __init__(self, features, n_groups=None, n_slots=4, device=0):
        self.features = np.asarray(features, dtype=bool)
        self.n_objects, self.n_features, self.n_states = self.features.shape
        self.n_groups = list(n_groups) if n_groups is not None else [1]
        self.n_components = len(self.n_groups)
        self.calls = []
        self.slots = {}                 # slot -> {"groups": {c: bool [G, N]}, "counts": {c: ...}, "weights": ...}
        self.conc = {}
        self._bound = {}                # the real Engine's bind cache protocol (binding._bind_slot): every
        self._bound_conc = {}           # slot-changing method drops the slot's entry
        self._bound_unif = None
        self._mirror = {}
        self.unif = None

    @property
    def group_offsets(self):
        return np.concatenate([[0], np.cumsum(self.n_groups)]).astype(int)

    @property
    def n_groups_total(self):
        return int(sum(self.n_groups))

    def _touch(self, slot):
        self._bound.pop(slot, None)
        self._mirror.pop(slot, None)

    def close(self):
        pass

    def na_values(self):
        return ~self.features.any(axis=-1)

    def component_lh(self, probs, groups, changed_groups, out, na_value=0.0):
        self.calls.append(("component_lh", tuple(int(c) for c in changed_groups)))
        groups = np.asarray(groups, dtype=bool)
        changed_groups = np.asarray(changed_groups, dtype=np.int64)
        orc.compute_component_likelihood(self.features, np.asarray(probs), groups, changed_groups, out)
        if na_value != 0.0:           # rows this call wrote: members of changed groups and no-group objects
            written = ~groups.any(axis=0)
            for i in changed_groups:
                written |= groups[i]
            rows = out[written]
            rows[self.na_values()[written]] = na_value
            out[written] = rows
        return out

    def normalize_tables(self, counts, concentration, temperature=None, prior_temperature=None, unif_counts=None):
        self.calls.append(("normalize_tables", np.shape(counts)[0]))
        return orc.conditional_effect_mean(np.asarray(concentration, dtype=np.float64), np.asarray(counts),
                                           unif_counts=unif_counts, prior_temperature=prior_temperature,
                                           temperature=temperature)

    def dirichlet_logpdf(self, counts, concentration, per_group=False):
        counts = np.asarray(counts, dtype=np.float32)
        conc = np.asarray(concentration, dtype=np.float64)
        self.calls.append(("dirichlet_logpdf", counts.shape[0]))
        a = conc if conc.ndim == 3 else np.broadcast_to(conc, counts.shape)
        pf = np.stack([orc.dirichlet_categorical_logpdf(counts[g], a[g]) for g in range(counts.shape[0])])
        if not per_group:
            return pf
        return pf, np.array([float(row.sum()) for row in pf])

    def effect_counts(self, group_assignment, source_is_component, object_subset=None):
        subset = slice(None) if object_subset is None else np.asarray(object_subset)
        return orc.compute_effect_counts(self.features, np.asarray(group_assignment, dtype=bool),
                                         np.asarray(source_is_component, dtype=bool), subset)

    # ---- resident-slot surface used by the device forms of the operators (sbayes_amd/operators.py) ----
    def _slot(self, slot):
        return self.slots.setdefault(slot, {"groups": {}, "counts": {}, "weights": None})

    def set_groups(self, slot, component, groups):
        self._touch(slot)
        self.calls.append(("set_groups", component))
        g = np.asarray(groups, dtype=bool)
        # the real engine's contract (sbe_set_groups, round 6): an object in several groups keeps the LAST one as its id -- what
        # an uncached likelihood evaluation ends up with (likelihood.py:126-130; the oracle's a1 does the same on the matrix) --
        # and the slot is marked: calls that would derive COUNTS from one id per object refuse it (_reject_overlap)
        multi = np.flatnonzero(g.sum(axis=0) > 1)
        marks = self._slot(slot).setdefault("overlap", {})
        if multi.size:
            n = int(multi[0])
            g1, g2 = (int(v) for v in np.flatnonzero(g[:, n])[:2])
            marks[component] = f"object {n} is in groups {g1} and {g2} of component {component}"
        else:
            marks.pop(component, None)
        self._slot(slot)["groups"][component] = g.copy()

    def _reject_overlap(self, slot):
        marks = self._slot(slot).get("overlap")
        if marks:
            from sbayes_amd.engine import GroupOverlapError
            raise GroupOverlapError(4, next(iter(marks.values())))

    def set_concentration(self, component, concentration):
        self._bound.clear()
        self._bound_conc.pop(component, None)
        self.conc[component] = np.asarray(concentration, dtype=np.float64).copy()

    def set_counts(self, slot, component, counts):
        self._touch(slot)
        self.calls.append(("set_counts", component))
        self._slot(slot)["counts"][component] = np.asarray(counts, dtype=np.float32).copy()

    def set_source(self, slot, source):
        self._touch(slot)
        self.calls.append(("set_source",))
        self._slot(slot)["source"] = np.asarray(source, dtype=bool).copy()

    def observation_lh_exact(self, slot):
        self.calls.append(("observation_lh_exact",))
        s = self.slots[slot]
        C = len(s["groups"])
        groups = [s["groups"][c] for c in range(C)]
        counts = [s["counts"][c] for c in range(C)]
        conc = [self.conc[c] for c in range(C)]
        lh = orc.likelihood_per_component_exact(self.features, self.na_values(), groups, counts, conc, s["source"])
        w = orc.normalize_weights(s["weights"], orc.has_components(groups))
        return orc.logger_row(w, lh).reshape(self.n_objects, self.n_features)

    def set_weights(self, slot, weights):
        self._touch(slot)
        self.calls.append(("set_weights",))
        self._slot(slot)["weights"] = np.asarray(weights, dtype=np.float32).copy()

    def update_probs(self, slot, component):
        pass                            # (one index or several) tables are derived on demand below

    def _state(self, slot):
        s = self.slots[slot]
        C = len(s["groups"])
        groups = [s["groups"][c] for c in range(C)]
        counts = [s["counts"][c] for c in range(C)]
        conc = [self.conc[c] for c in range(C)]
        lh = orc.likelihood_per_component(self.features, self.na_values(), groups, counts, conc)
        return groups, s["weights"], lh

    def cluster_marginals(self, slot, table, objects, prior_temperature=1.0):
        self.calls.append(("cluster_marginals", len(objects)))
        groups, weights, lh = self._state(slot)
        available = np.zeros(self.n_objects, dtype=bool)
        available[np.asarray(objects)] = True
        wz = orc.feature_weights_with_and_without(weights, orc.has_components(groups), available, prior_temperature)
        with np.errstate(divide="ignore"):
            return np.log(orc.cluster_marginals(self.features, self.na_values(), lh, np.asarray(table), available, wz))

    def source_posterior(self, slot, objects, temperature=1.0, prior_temperature=1.0):
        self.calls.append(("source_posterior", len(objects)))
        groups, weights, lh = self._state(slot)
        w = orc.normalize_weights(weights, orc.has_components(groups))
        return orc.source_posterior(lh, w, np.asarray(objects), temperature, prior_temperature)

    def subset_lh(self, objects, tables, group_idx, temperature=1.0):
        """float32 [n, F, C]: tables[c][group_idx[c][i]][f][x(objects[i], f)], -1 -> 0, NA -> 1, then ** (1/T)."""
        self.calls.append(("subset_lh", len(objects)))
        objects = np.asarray(objects)
        feats = self.features[objects]
        out = np.zeros((objects.size, self.n_features, len(tables)), dtype=np.float32)
        for c, tab in enumerate(tables):
            tab = np.asarray(tab, dtype=np.float32)
            gi = np.asarray(group_idx[c])
            has = gi >= 0
            lh = np.einsum("ijk,ijk->ij", feats[has], tab[gi[has]])
            out[has, :, c] = lh
        out[self.na_values()[objects]] = 1.0
        return out ** np.float32(1 / temperature) if temperature != 1.0 else out

    def jump_lh(self, slot, pconf, p_source, p_target, objects, prior_temperature=1.0):
        """float64 [2, n]: sums of logs of the reference's float32 per-feature stay / jump likelihoods."""
        self.calls.append(("jump_lh", len(objects)))
        s = self.slots[slot]
        C = len(s["groups"])
        groups = [s["groups"][c] for c in range(C)]
        objects = np.asarray(objects)
        wh = orc.weights_heated(s["weights"], orc.has_components(groups), prior_temperature)[objects]
        feats = self.features[objects]
        pconf = np.asarray(pconf, dtype=np.float32).reshape((-1,) + self.features.shape[1:])
        expected = np.zeros(feats.shape, dtype=np.float32)
        off = 0
        for c in range(1, C):
            for i_g, g in enumerate(groups[c]):
                m = g[objects]
                expected[m] += wh[m][:, :, c][..., None] * pconf[off + i_g][None]
            off += groups[c].shape[0]
        ps = np.asarray(p_source, dtype=np.float32).reshape(self.features.shape[1:])
        pt = np.asarray(p_target, dtype=np.float32).reshape(self.features.shape[1:])
        stay = np.sum(feats * (expected + wh[:, :, 0][..., None] * ps[None]), axis=-1)
        jump = np.sum(feats * (expected + wh[:, :, 0][..., None] * pt[None]), axis=-1)
        valid = ~self.na_values()[objects]
        with np.errstate(divide="ignore"):
            return np.stack([np.where(valid, np.log(stay.astype(np.float64)), 0.0).sum(axis=-1),
                             np.where(valid, np.log(jump.astype(np.float64)), 0.0).sum(axis=-1)])

    def source_lh_by_feature(self, slot):
        self.calls.append(("source_lh_by_feature",))
        s = self.slots[slot]
        groups = [s["groups"][c] for c in range(len(s["groups"]))]
        w = orc.normalize_weights(s["weights"], orc.has_components(groups))
        return orc.source_lh_by_feature(s["source"], w, self.na_values())

    def normalize_weights(self, weights, has_components):
        return orc.normalize_weights(np.asarray(weights, dtype=np.float32), np.asarray(has_components, dtype=bool))

    def recount(self, slot, component=-1):
        self._reject_overlap(slot)
        self._touch(slot)
        self.calls.append(("recount",))
        s = self._slot(slot)
        groups = [s["groups"][c] for c in range(len(s["groups"]))]
        for c, table in enumerate(orc.recalculate_feature_counts(self.features, groups, s["source"])):
            s["counts"][c] = table

    def mixture_loglik(self, slot=0):
        self.calls.append(("mixture_loglik",))
        groups, counts, conc, s = self._full_state(slot)
        return float(orc.mixture_loglik(self.features, self.na_values(), groups, counts, conc, s["weights"]))

    def get_counts(self, slot, component):
        return self._slot(slot)["counts"][component].copy()

    def get_counts_all(self, slot):
        self.calls.append(("get_counts_all",))
        return tuple(self._slot(slot)["counts"][c].copy() for c in range(len(self.n_groups)))

    # ---- Gibbs source proposal on slot state (operators.gibbs_sample_source; sbe_sample_source / sbe_source_logprob) ----
    def copy_slot(self, dst, src):
        import copy
        self._touch(dst)
        self.calls.append(("copy_slot",))
        self.slots[dst] = copy.deepcopy(self._slot(src))

    def _source_probs(self, slot, objects, temperature, prior_temperature, from_prior):
        groups, weights, lh = self._state(slot)
        w = orc.normalize_weights(weights, orc.has_components(groups))
        if from_prior:                                            # operators.py:520-522
            return orc.normalize(w[objects] ** (1 / float(prior_temperature)), axis=-1)
        return orc.source_posterior(lh, w, objects, float(temperature), float(prior_temperature))

    def sample_source(self, slot, dst_slot, objects, z, temperature=1.0, prior_temperature=1.0, from_prior=False,
                      return_selected=False):
        self._touch(dst_slot)
        self.calls.append(("sample_source", len(objects)))
        objects = np.asarray(objects)
        p = self._source_probs(slot, objects, temperature, prior_temperature, from_prior)
        idx = orc.sample_categorical(p, np.asarray(z, dtype=np.float64).reshape(objects.size, self.n_features))
        x = np.eye(p.shape[-1], dtype=bool)[idx]
        na = self.na_values()[objects]
        x[na] = False                                             # operators.py:527
        self._slot(dst_slot)["source"][objects] = x
        sel = np.take_along_axis(p, idx[..., None], axis=-1)[..., 0].astype(np.float32)
        sel[na] = 1.0                                             # (the engine's convention: an NA observation selects nothing)
        with np.errstate(divide="ignore"):
            log_q = float(np.log(sel[~na]).sum())
        return (log_q, sel) if return_selected else log_q

    def source_logprob(self, slot, src_slot, objects, temperature=1.0, prior_temperature=1.0, from_prior=False,
                       return_selected=False):
        self.calls.append(("source_logprob", len(objects)))
        objects = np.asarray(objects)
        p = self._source_probs(slot, objects, temperature, prior_temperature, from_prior)
        src = self._slot(src_slot)["source"][objects]
        sel = np.where(src.any(axis=-1), np.take_along_axis(p, src.argmax(axis=-1)[..., None], axis=-1)[..., 0], 1.0).astype(np.float32)
        na = self.na_values()[objects]
        sel[na] = 1.0
        with np.errstate(divide="ignore"):
            log_q = float(np.log(sel[~na]).sum())
        return (log_q, sel) if return_selected else log_q

    def update_counts(self, slot_new, slot_old, objects):
        self._touch(slot_new)
        self.calls.append(("update_counts", len(objects)))
        new, old = self._slot(slot_new), self._slot(slot_old)
        C = len(new["groups"])
        mask = np.zeros(self.n_objects, dtype=bool)
        mask[np.asarray(objects)] = True
        counts, changed = orc.update_feature_counts([old["counts"][c] for c in range(C)], self.features,
                                                    [old["groups"][c] for c in range(C)], [new["groups"][c] for c in range(C)],
                                                    old["source"], new["source"], mask)
        for c in range(C):
            new["counts"][c] = counts[c]
        return np.concatenate(changed)

    def get_source_rows(self, slot, objects):
        return self._slot(slot)["source"][np.asarray(objects)].copy()

    # ---- round 3: delta / resident forms ------------------------------------------------------------------------
    def set_uniform_counts(self, unif_counts):
        self.unif = np.asarray(unif_counts, dtype=np.float64).copy()

    def set_counts_rows(self, slot, group_idx, rows, update_probs=False):
        self._touch(slot)                               # (update_probs: tables are derived on demand here)
        self.calls.append(("set_counts_rows", len(group_idx)))
        off = self.group_offsets
        for gg, row in zip(np.asarray(group_idx), np.asarray(rows, dtype=np.float32)):
            c = int(np.searchsorted(off, gg, side="right") - 1)
            self._slot(slot)["counts"][c][gg - off[c]] = row

    def set_slot_delta(self, slot, groups_component=0, groups=None, count_idx=None, count_rows=None, update_probs=False,
                       source_objects=None, source_rows=None):
        """set_groups, set_counts_rows, set_source_rows of one bind as ONE call (class-qualified: one logged call)."""
        n_calls = len(self.calls)
        if groups is not None:
            FakeEngine.set_groups(self, slot, groups_component, groups)
        if count_idx is not None:
            FakeEngine.set_counts_rows(self, slot, count_idx, count_rows, update_probs=update_probs)
        if source_objects is not None:
            FakeEngine.set_source_rows(self, slot, source_objects, source_rows)
        del self.calls[n_calls:]
        self.calls.append(("set_slot_delta", groups is not None, count_idx is not None, source_objects is not None))

    def set_source_rows(self, slot, objects, rows):
        self._touch(slot)
        self.calls.append(("set_source_rows", len(objects)))
        self._slot(slot)["source"][np.asarray(objects)] = np.asarray(rows, dtype=bool)

    def counts_delta(self, objects, gid_old, gid_new, src_old, src_new, follow_slot=None, update_probs=False, update_source=False):
        """(touched, diff rows) of update_feature_counts for the listed objects: the oracle's a9 on one-object
        group matrices rebuilt from the ids.  follow_slot: that slot's counts take the difference (tables are derived on
        demand here, so update_probs has nothing to do)."""
        self.calls.append(("counts_delta", len(objects)))
        if follow_slot is not None:
            self._touch(follow_slot)
        objects = np.asarray(objects)
        gid_old, gid_new = np.asarray(gid_old), np.asarray(gid_new)
        src_old, src_new = np.asarray(src_old), np.asarray(src_new)
        touched = np.union1d(gid_old[gid_old >= 0], gid_new[gid_new >= 0]).astype(np.int32)
        off = self.group_offsets
        diff = np.zeros((touched.size, self.n_features, self.n_states), dtype=np.float32)
        feats = self.features[objects]
        for t, gg in enumerate(touched):
            c = int(np.searchsorted(off, gg, side="right") - 1)
            for sign, gid, src in ((1.0, gid_new, src_new), (-1.0, gid_old, src_old)):
                members = gid[c] == gg
                if members.any():
                    diff[t] += sign * np.count_nonzero((src[members] == c)[:, :, None] & feats[members], axis=0)
            if follow_slot is not None:
                self._slot(follow_slot)["counts"][c][gg - off[c]] += diff[t]
        if follow_slot is not None and update_source and objects.size:
            n_comp = self._slot(follow_slot)["source"].shape[2]
            self._slot(follow_slot)["source"][objects] = src_new[..., None] == np.arange(n_comp)
        return touched, diff

    def _full_state(self, slot):
        s = self.slots[slot]
        C = len(s["groups"])
        return ([s["groups"][c] for c in range(C)], [s["counts"][c] for c in range(C)], [self.conc[c] for c in range(C)], s)

    def collapsed_loglik(self, slot, component, per_feature=False):
        self.calls.append(("collapsed_loglik", component))
        _, counts, conc, _ = self._full_state(slot)
        return orc.collapsed_group_logliks(counts[component], conc[component])

    def collapsed_loglik_all(self, slot):
        self.calls.append(("collapsed_loglik_all",))
        _, counts, conc, _ = self._full_state(slot)
        parts = [orc.collapsed_group_logliks(counts[c], conc[c]) if self.n_groups[c] else np.zeros(0)
                 for c in range(len(counts))]
        return np.concatenate(parts).astype(np.float64)

    def source_prior(self, slot):
        self.calls.append(("source_prior",))
        groups, _, _, s = self._full_state(slot)
        w = orc.normalize_weights(s["weights"], orc.has_components(groups))
        return orc.source_prior_per_object(w, s["source"], self.na_values())

    def collapsed_and_source_prior(self, slot):
        self.calls.append(("collapsed_and_source_prior",))
        groups, counts, conc, s = self._full_state(slot)
        parts = [orc.collapsed_group_logliks(counts[c], conc[c]) if self.n_groups[c] else np.zeros(0)
                 for c in range(len(counts))]
        w = orc.normalize_weights(s["weights"], orc.has_components(groups))
        return np.concatenate(parts).astype(np.float64), orc.source_prior_per_object(w, s["source"], self.na_values())

    def given_unchanged_lh(self, slot, i_cluster, objects, temperature=1.0, prior_temperature=1.0):
        self.calls.append(("given_unchanged_lh", len(objects)))
        groups, counts, conc, s = self._full_state(slot)
        subset = np.zeros(self.n_objects, dtype=bool)
        subset[np.asarray(objects)] = True
        return orc.component_likelihood_given_unchanged(
            self.features, self.na_values(), groups, counts, conc, s["source"], subset, i_cluster, self.unif,
            [self.unif] * (len(groups) - 1), temperature=temperature, prior_temperature=prior_temperature)

    def given_unchanged_gibbs(self, slot, i_cluster, objects, hc_new, hc_old, src_old, z, temperature=1.0, prior_temperature=1.0,
                              from_prior=False, gid_old=None, gid_new=None, follow=False, update_probs=False):
        """ClusterOperator.gibbs_sample_source (operators.py:808-847) restated with the oracle's pieces: the expressions of
        the reference, in its dtypes, on the subset.  follow: the slot takes the proposal when it touches any group (counts
        += delta, the subset's source rows = the drawn ids; tables are derived on demand here)."""
        objects = np.asarray(objects)
        # (class-qualified: a recording / memoising subclass must not log this inner step as a call of its own -- the real
        #  engine receives ONE call)
        lh = FakeEngine.given_unchanged_lh(self, slot, i_cluster, objects, temperature, prior_temperature)     # float32 [n, F, C]
        self.calls[-1] = ("given_unchanged_gibbs", len(objects))
        weights = self._slot(slot)["weights"]
        inv_tp = 1 / float(prior_temperature)
        na = self.na_values()[objects]
        z = np.asarray(z, dtype=np.float64).reshape(objects.size, self.n_features)
        probs = []
        for hc in (hc_new, hc_old):
            w = orc.normalize_weights(weights, np.asarray(hc, dtype=bool)) ** inv_tp
            probs.append(w if from_prior else orc.normalize(w * lh, axis=-1))
        p, p_back = probs
        idx = orc.sample_categorical(p, z)
        ids = idx.astype(np.uint8)
        ids[na] = 255
        sel = np.take_along_axis(p, idx[..., None], axis=-1)[..., 0].astype(np.float32)
        sel[na] = 1.0
        so = np.asarray(src_old)
        back = np.where(so != 255, np.take_along_axis(p_back, np.minimum(so, p.shape[-1] - 1).astype(np.int64)[..., None], axis=-1)[..., 0],
                        1.0).astype(np.float32)
        if gid_old is not None:                 # + the proposal's count delta (the double's own counts_delta, not logged separately)
            touched, rows = FakeEngine.counts_delta(self, objects, gid_old, gid_new, so, ids,
                                                    follow_slot=slot if follow else None, update_source=bool(follow))
            self.calls.pop()
            if follow and touched.size == 0:          # (the engine's rule: nothing touched, nothing follows)
                pass
            return ids, sel, back, touched, rows
        return ids, sel, back

    def gibbs_propose_supported(self):
        return True

    def gibbs_propose(self, cur_slot, cand_slot, objects, z, temperature=1.0, prior_temperature=1.0, from_prior=False, follow=False):
        """GibbsSampleSource._propose in one call, composed of the double's own pieces (class-qualified: ONE logged call).
        follow: when any group is touched, the current slot takes the proposal (it becomes the candidate's state)."""
        objects = np.asarray(objects)
        n_calls = len(self.calls)
        FakeEngine.copy_slot(self, cand_slot, cur_slot)
        _, sel = FakeEngine.sample_source(self, cur_slot, cand_slot, objects, z, temperature, prior_temperature, from_prior,
                                          return_selected=True)
        FakeEngine.update_counts(self, cand_slot, cur_slot, objects)
        _, back = FakeEngine.source_logprob(self, cand_slot, cur_slot, objects, temperature, prior_temperature, from_prior,
                                            return_selected=True)
        rows_new = FakeEngine.get_source_rows(self, cand_slot, objects)
        del self.calls[n_calls:]
        self.calls.append(("gibbs_propose", len(objects)))
        ids = np.where(rows_new.any(-1), rows_new.argmax(-1), 255).astype(np.uint8)
        cur, cand = self._slot(cur_slot), self._slot(cand_slot)
        off = self.group_offsets
        seen = np.zeros(int(off[-1]), dtype=bool)
        for c in range(len(cur["groups"])):
            sub = np.asarray(cur["groups"][c])[:, objects]
            seen[off[c] + np.flatnonzero(sub.any(axis=1))] = True
        touched = np.flatnonzero(seen).astype(np.int32)
        rows = np.zeros((touched.size, self.n_features, self.n_states), dtype=np.float32)
        for j, gg in enumerate(touched):
            c = int(np.searchsorted(off, gg, side="right") - 1)
            rows[j] = cand["counts"][c][gg - off[c]] - cur["counts"][c][gg - off[c]]
        if follow and touched.size:
            import copy
            self._touch(cur_slot)
            self.slots[cur_slot] = copy.deepcopy(cand)
        return ids, np.asarray(sel, dtype=np.float32), np.asarray(back, dtype=np.float32), touched, rows

    def cluster_posterior_marginals(self, slot, i_cluster, objects, temperature=1.0, prior_temperature=1.0):
        _, counts, conc, _ = self._full_state(slot)
        table = orc.conditional_effect_mean(conc[0], counts[0][[i_cluster]], unif_counts=self.unif,
                                            prior_temperature=prior_temperature, temperature=temperature)
        return FakeEngine.cluster_marginals(self, slot, table, objects, prior_temperature)     # (not the recording subclass's: ONE engine call)

    def jump_lh_resident(self, slot, i_source, i_target, objects, temperature=1.0, prior_temperature=1.0):
        self.calls.append(("jump_lh_resident", len(objects)))
        groups, counts, conc, s = self._full_state(slot)
        assert np.array_equal(np.asarray(objects), np.flatnonzero(groups[0][i_source]))
        return orc.jump_log_lh(self.features, self.na_values(), groups, counts, conc, self.unif, s["weights"], i_source,
                               i_target, temperature, prior_temperature)


def make_get_engine(engines, cls=None):
    """`registry.get_engine` for the tests: one double per feature block, kept in the dict `engines`.  Like the real
    registry (which re-creates an engine whose component layout differs), a double first made by a stateless call
    (n_groups unknown -> [1]) adopts the layout once a caller names it."""
    cls = cls or FakeEngine

    def get_engine(features, n_groups=None, n_slots=4, device=None):
        key = (np.asarray(features).ctypes.data, np.asarray(features).shape)
        if key not in engines:
            engines[key] = cls(features, n_groups)
        elif n_groups is not None and list(n_groups) != engines[key].n_groups and engines[key].n_groups == [1]:
            engines[key].n_groups = [int(g) for g in n_groups]
            engines[key].n_components = len(n_groups)
        return engines[key]
    return get_engine


def make_engine_for_observations(engines):
    """`registry.engine_for_observations` for the tests: the double whose feature block has the NA mask handed over."""
    def engine_for_observations(na_features, n_components, n_groups=None):
        na = np.asarray(na_features)
        for e in engines.values():
            if n_groups is not None and [int(g) for g in n_groups] != list(e.n_groups):
                continue
            if (e.n_objects, e.n_features) == na.shape and e.n_components == n_components and np.array_equal(e.na_values(), na):
                return e
        return None
    return engine_for_observations
