#!/usr/bin/env python3
"""GPU box: the WIDE matrix-pipe forms (more than 8 group tuples: 4 / 2 slots per block) against the vector-pipe group-tuple form
on the same resident states -- kernel time per launch by HIP event pairs, both oracle-checked on a few slots.
    python tools/ab_wide.py [B ...]"""
import json
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import sbayes_oracle as orc                                       # noqa: E402  (checker only)
from sbayes_amd.engine import MIXTURE_PACKED, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA, Engine, EngineError   # noqa: E402
from tests.test_gpu_shapes import random_case                                  # noqa: E402

SHAPES = [  # name, N, F, S, n_groups
    ("south_america-like 100x36x5 [3,1,6]", 100, 36, 5, [3, 1, 6]),
    ("400x60x5 [3,1,6]", 400, 60, 5, [3, 1, 6]),
    ("1000x100x6 [4,1,3]", 1000, 100, 6, [4, 1, 3]),
    ("2000x100x6 [4,1,3]", 2000, 100, 6, [4, 1, 3]),
    ("1000x200x10 [5,1,4]", 1000, 200, 10, [5, 1, 4]),
    ("1500x80x4 [3,1,3,3]", 1500, 80, 4, [3, 1, 3, 3]),
    ("3000x120x8 [7,1,5]", 3000, 120, 8, [7, 1, 5]),
]


def main():
    batches = [int(a) for a in sys.argv[1:]] or [512, 2048]
    for name, N, F, S, n_groups in SHAPES:
        rng = np.random.default_rng(N + F)
        feats, groups0, _w, _s, conc = random_case(rng, N, F, S, n_groups, 0.03)
        na = ~feats.any(-1)
        C = len(n_groups)
        B = max(batches)
        with Engine(feats, n_groups, n_slots=B) as eng:
            for c in range(C):
                eng.set_concentration(c, conc[c])
            eng.set_option(deferred_checks=True)
            want = {}
            n_distinct = 16
            for b in range(n_distinct):
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
                if b < 3:
                    counts = orc.recalculate_feature_counts(feats, groups, source)
                    want[b] = float(orc.mixture_loglik(feats, na, groups, counts, conc, weights))
            for b in range(n_distinct, B):
                eng.copy_slot(b, b % n_distinct)
            eng.set_option(deferred_checks=False)
            row = {"shape": name}
            for nb in batches:
                for label, kernel in (("vector", MIXTURE_PACKED_TUPLE), ("matrix", MIXTURE_PACKED_TUPLE_MFMA)):
                    eng.set_option(kernel=kernel)
                    try:
                        got = eng.mixture_loglik_batch(0, nb)
                    except EngineError as exc:
                        row[f"{label}_b{nb}"] = "n/a: " + str(exc)[-60:]
                        continue
                    for b, w in want.items():
                        assert abs(got[b] - w) <= 1e-10 * abs(w), (name, label, b, got[b], w)
                    _t, k = eng.profile_mixture(0, nb, 30)
                    _t, k = eng.profile_mixture(0, nb, 50)
                    row[f"{label}_b{nb}"] = round(k * 1e3, 2)
                    row[f"{label}_kernel"] = eng.last_mixture_kernel().split("<")[0] + (" " + eng.last_mixture_kernel().split("matrix pipe ")[1].split(",")[1] if "matrix pipe" in eng.last_mixture_kernel() else "")
                eng.set_option(kernel=MIXTURE_PACKED)
                eng.mixture_loglik_batch(0, nb)
                _t, k = eng.profile_mixture(0, nb, 30)
                _t, k = eng.profile_mixture(0, nb, 50)
                row[f"default_b{nb}"] = [eng.last_mixture_kernel().split("<")[0], round(k * 1e3, 2)]
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
