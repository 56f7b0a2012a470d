#!/usr/bin/env python3
"""GPU box: soak of the one-launch operator forms (wave-specialised marginals / jump kernels, fused given_unchanged forms with
their count delta, gibbs_propose, count rows with their probability rows): every call repeated REPS times on the same
inputs -- two argument sets alternated call by call -- every result compared bit for bit with the first of its set: a race
between builder and object waves, a stale LDS table or a completion flag that overtakes a result would show as a
differing repetition (a result read too early is the OTHER set's).
   python tools/soak_operator_forms.py [reps] [shape ...]"""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from sbayes_amd.engine import Engine                                    # noqa: E402
from tests.test_gpu_delta_forms import _ids, _workload                   # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    shapes = sys.argv[2:] or ["cfg1_fixture", "south_america", "headline", "long"]
    for name in shapes:
        feats, groups, conc, weights, source, counts, unif = _workload(name)
        ng = [g.shape[0] for g in groups]
        eng = Engine(feats, ng, n_slots=3)
        eng.set_option(deferred_checks=True)
        for c in range(len(groups)):
            eng.set_concentration(c, conc[c]); eng.set_groups(0, c, groups[c]); eng.set_counts(0, c, counts[c])
        eng.set_source(0, source); eng.set_weights(0, weights); eng.set_uniform_counts(unif); eng.update_probs(0, range(len(groups)))
        rng = np.random.default_rng(7)
        N, F, C = source.shape
        K = ng[0]
        off = eng.group_offsets
        has = np.stack([g.any(axis=0) for g in groups], axis=1)
        available = np.flatnonzero(~groups[0].any(axis=0) | groups[0][0]).astype(np.int32)
        members = np.flatnonzero(groups[0][0]).astype(np.int32)
        # TWO argument sets per form, alternated call by call: a result read before it has landed would be the OTHER set's
        # (the same buffers serve both), not a repeat of the right one
        variants = []
        for v in range(2):
            objs = np.sort(rng.choice(N, size=min(N, 23), replace=False)).astype(np.int32)
            hc = has[objs].copy()
            if C > 1:
                hc[:, 1] = True
            so = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
            z = rng.random((objs.size, F))
            gid = np.stack([_ids(groups[c], objs, off[c]) for c in range(C)])
            rows_val = (counts[0][:1] + v).astype(np.float32)
            variants.append((objs, hc, so, z, gid, rows_val, available[v::2].copy(), members[v::2].copy(), v % K))
        rows_idx = np.array([0], dtype=np.int32)
        eng.copy_slot(2, 0)                               # the FOLLOWING slot of the *_apply forms (third session)

        def follow_pair(objs, gid, so):
            """A slot follows a difference and its reverse (sbe_counts_delta_apply: counts, probability rows and source rows
            behind the completion flag): what is read back right after each call is complete, and the slot returns to its base."""
            sn = np.where(so == 255, 255, (so + 1) % C).astype(np.uint8)
            _, d1 = eng.counts_delta(objs, gid, gid, so, sn, follow_slot=2, update_probs=True, update_source=True)
            c1, p1, s1 = eng.get_counts(2, 0), eng.get_probs(2, 0), eng.get_source_rows(2, objs)
            _, d2 = eng.counts_delta(objs, gid, gid, sn, so, follow_slot=2, update_probs=True, update_source=True)
            return d1, c1, p1, s1, d2, eng.get_counts(2, 0), eng.get_probs(2, 0), eng.get_source_rows(2, objs)

        def forms(v):
            objs, hc, so, z, gid, rows_val, avail, memb, k = variants[v]
            return {
                "cluster_posterior_marginals": lambda: eng.cluster_posterior_marginals(0, k, avail, 1.0, 1.0),
                "jump_lh_resident": (lambda: eng.jump_lh_resident(0, 0, 1 % K, memb, 1.0, 1.0)) if memb.size and K > 1 else None,
                "given_unchanged_lh": lambda: eng.given_unchanged_lh(0, k, objs, 1.0, 1.0),
                "given_unchanged_gibbs+counts": lambda: eng.given_unchanged_gibbs(0, k, objs, hc, hc, so, z, 1.0, 1.0, False, gid_old=gid, gid_new=gid),
                "gibbs_propose": (lambda: eng.gibbs_propose(0, 1, objs, z)) if eng.gibbs_propose_supported() else None,
                "set_counts_rows(update_probs) + mixture": lambda: (eng.set_counts_rows(0, rows_idx, rows_val, update_probs=True), eng.mixture_loglik(0))[1],
                "counts_delta": lambda: eng.counts_delta(objs, gid, gid, so, np.where(so == 255, 255, (so + 1) % C).astype(np.uint8)),
                "counts_delta, a slot following + back": (lambda: follow_pair(objs, gid, so)) if C > 1 else None,
            }
        f0, f1 = forms(0), forms(1)
        for label in f0:
            fns = (f0[label], f1[label])
            if fns[0] is None or fns[1] is None:
                continue
            try:
                firsts = []
                for fn in fns:
                    first = fn()
                    first = first if isinstance(first, tuple) else (first,)
                    firsts.append(tuple(np.array(a, copy=True) for a in first))
            except Exception as exc:
                print(f"{name:14s} {label:40s} skipped: {exc}", flush=True)
                continue
            assert any(not np.array_equal(a, b) for a, b in zip(*firsts) if a.shape == b.shape) or any(a.shape != b.shape for a, b in zip(*firsts)), \
                (name, label, "the two argument sets give the same result: nothing to tell apart")
            t0 = time.time()
            for r in range(reps):
                got = fns[r & 1]()
                got = got if isinstance(got, tuple) else (got,)
                for a, b in zip(got, firsts[r & 1]):
                    if not np.array_equal(np.asarray(a), b):
                        raise SystemExit(f"{name} {label}: repetition {r} differs from the first call with these arguments")
            print(f"{name:14s} {label:40s} {reps} alternating repetitions identical ({(time.time() - t0) / reps * 1e6:.1f} us each with the compare)", flush=True)
        eng.close()
    print("[soak] done, no differing repetition", flush=True)


if __name__ == "__main__":
    main()
