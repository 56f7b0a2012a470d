#!/usr/bin/env python3
"""GPU box: what a host-synchronous engine call costs at the least -- wall time per call of the one-launch calls of the
drop-in path at a given shape, through the Python wrapper and straight through ctypes (the wrapper's share), next to an
idle-stream synchronisation.  Kernel times come from a rocprofv3 --kernel-trace run of the same script.
   python tools/call_floor.py [headline|south_america|cfg1] [reps]"""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from sbayes_amd.engine import Engine, _ptr        # noqa: E402
from tests.test_gpu_delta_forms import _workload   # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "headline"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    feats, groups, conc, weights, source, counts, unif = _workload(name)
    n_groups = [g.shape[0] for g in groups]
    eng = Engine(feats, n_groups, n_slots=2)
    eng.set_option(deferred_checks=True)
    for c in range(len(groups)):
        eng.set_concentration(c, conc[c])
        eng.set_groups(0, c, groups[c])
        eng.set_counts(0, c, counts[c])
    eng.set_source(0, source)
    eng.set_weights(0, weights)
    eng.set_uniform_counts(unif)
    eng.update_probs(0, range(len(groups)))
    N, F = feats.shape[0], feats.shape[1]
    K = n_groups[0]
    available = np.flatnonzero(~groups[0].any(axis=0) | groups[0][0]).astype(np.int32)
    members = np.flatnonzero(groups[0][0]).astype(np.int32)
    some = np.sort(np.random.default_rng(0).choice(N, size=min(N, 12), replace=False)).astype(np.int32)
    lib, h = eng._lib, eng._h
    out_sp = np.empty(N)
    out_cl = np.empty(eng.n_groups_total)
    out_cm = np.empty((2, available.size))
    calls = {
        "sync (idle stream)": (lambda: eng.sync(), lambda: lib.sbe_sync(h)),
        "roundtrip: empty kernel, 1 block": (None, lambda: lib.sbe_test_roundtrip(h, 1, 0)),
        "roundtrip: + read a mapped word": (None, lambda: lib.sbe_test_roundtrip(h, 1, 1)),
        "roundtrip: + store a mapped double": (None, lambda: lib.sbe_test_roundtrip(h, 1, 3)),
        "roundtrip: 64 blocks, read + store": (None, lambda: lib.sbe_test_roundtrip(h, 64, 3)),
        "roundtrip: 1000 blocks, read + store": (None, lambda: lib.sbe_test_roundtrip(h, 1000, 3)),
        "source_prior": (lambda: eng.source_prior(0), lambda: lib.sbe_source_prior(h, 0, _ptr(out_sp))),
        "collapsed_loglik_all": (lambda: eng.collapsed_loglik_all(0), lambda: lib.sbe_collapsed_loglik_all(h, 0, _ptr(out_cl))),
        "collapsed_and_source_prior": (lambda: eng.collapsed_and_source_prior(0), None),
        "mixture_loglik": (lambda: eng.mixture_loglik(0), None),
        "source_lh_by_feature": (lambda: eng.source_lh_by_feature(0), None),
        f"cluster_posterior_marginals[{available.size}]": (
            lambda: eng.cluster_posterior_marginals(0, 0, available, 1.0, 1.0),
            lambda: lib.sbe_cluster_posterior_marginals(h, 0, 0, 1.0, 1.0, _ptr(available), available.size, _ptr(out_cm))),
        "cluster_posterior_marginals[8]": (lambda: eng.cluster_posterior_marginals(0, 0, available[:8], 1.0, 1.0), None),
        "cluster_posterior_marginals[64]": (lambda: eng.cluster_posterior_marginals(0, 0, available[:64], 1.0, 1.0), None),
        "cluster_posterior_marginals[256]": (lambda: eng.cluster_posterior_marginals(0, 0, available[:256], 1.0, 1.0), None),
        f"jump_lh_resident[{members.size}]": (lambda: eng.jump_lh_resident(0, 0, 1 % K, members, 1.0, 1.0), None),
        f"given_unchanged_lh[{some.size}]": (lambda: eng.given_unchanged_lh(0, 0, some, 1.0, 1.0), None),
        f"source_posterior[{some.size}]": (lambda: eng.source_posterior(0, some, 1.0, 1.0), None),
        "update_probs + sync": (lambda: (eng.update_probs(0, 0), eng.sync()), None),
    }
    only = sys.argv[3].split(",") if len(sys.argv) > 3 else None
    for label, (py, raw) in calls.items():
        if only and not any(o in label for o in only):
            continue
        for fn, kind in ((py, "python"), (raw, "ctypes")):
            if fn is None:
                continue
            if kind == "ctypes" and py is not None and only is None and "roundtrip" not in label and "[587]" not in label and label not in ("source_prior",):
                continue
            for _ in range(200):
                fn()
            best = None
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                dt = (time.perf_counter() - t0) / reps * 1e6
                best = dt if best is None else min(best, dt)
            print(f"{name:14s} {label:42s} {kind:7s} {best:7.2f} us/call", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
