#!/usr/bin/env python3
"""Kernel-trace target for the literal a1 / a3 surfaces at the headline shape: 200 calls of each (run under
`rocprofv3 --kernel-trace --stats -- python3 tools/prof_surfaces.py`; the streamed kernels' durations are the PCIe transfer)."""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from sbayes_amd.engine import Engine                 # noqa: E402
from sbayes_amd.synthetic import make_workload       # noqa: E402

wl = make_workload("headline")
eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=1)
for c in range(wl.n_components):
    eng.set_concentration(c, wl.concentration[c])
eng.load_state(0, wl.groups, wl.weights, source=wl.source)
for c in range(wl.n_components):
    eng.update_probs(0, c)
N, F, _ = wl.shape
buf = np.empty((N, F, wl.n_components))
probs0 = eng.get_probs(0, 0)
allg = np.arange(wl.groups[0].shape[0])
for name, fn in (("a1", lambda: eng.component_lh(probs0, wl.groups[0], allg, buf[..., 0])),
                 ("a3", lambda: eng.likelihood_per_component(0, buf))):
    for _ in range(20):
        fn()
    t0 = time.perf_counter()
    for _ in range(200):
        fn()
    print(f"{name}: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call", flush=True)
eng.close()
