#!/usr/bin/env python3
"""Same-box A/B of the literal call surfaces: a1 / a3 / single eval / PCIe-inclusive eval
with the pipelined copy-out of large results on the host pool (round 4) against the calling thread alone and against one
plain copy (SBE_D2H_PIECES=1: round-1 behaviour).  Each setting runs in a fresh child process (the switches are read once
per process), alternating, three times."""
import json
import os
import subprocess
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
CHILD = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, %r)
from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload
wl = make_workload("headline")
eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=1)
for c in range(wl.n_components):
    eng.set_concentration(c, wl.concentration[c])
eng.load_state(0, wl.groups, wl.weights, source=wl.source)
for c in range(wl.n_components):
    eng.update_probs(0, c)
def rate(fn, t=0.4):
    fn(); n = 0; t0 = time.perf_counter()
    while time.perf_counter() - t0 < t:
        fn(); n += 1
    return n / (time.perf_counter() - t0)
N, F, _ = wl.shape
buf = np.empty((N, F, wl.n_components))
probs0 = eng.get_probs(0, 0)
allg = np.arange(wl.groups[0].shape[0])
counts = [eng.get_counts(0, c) for c in range(wl.n_components)]
def pcie():
    for c in range(wl.n_components):
        eng.set_groups(0, c, wl.groups[c]); eng.set_counts(0, c, counts[c]); eng.update_probs(0, c)
    eng.set_weights(0, wl.weights)
    return eng.mixture_loglik(0)
out = {"a1": rate(lambda: eng.component_lh(probs0, wl.groups[0], allg, buf[..., 0])),
       "a3": rate(lambda: eng.likelihood_per_component(0, buf)),
       "single_eval": rate(lambda: eng.mixture_loglik(0)),
       "pcie_inclusive": rate(pcie)}
print(json.dumps({k: round(v, 1) for k, v in out.items()}))
''' % str(REPO)

# round 4: results streamed by the kernel into host-mapped staging and copied out chunk by chunk on the host pool (default)
# against the calling thread alone (SBE_D2H_THREADS=1), against the copy-engine path (SBE_STREAM_RESULTS=0: one
# hipMemcpyAsync, then the copy out of staging on the pool / on one thread = the round-1 form); pool sizes: SBE_STEP_THREADS
SETTINGS = [("default (ordered, 8 chunks)", {}), ("one thread", {"SBE_D2H_THREADS": "1"}),
            ("ordered, 16 chunks", {"SBE_STREAM_CHUNKS": "16"}), ("ordered, 4 chunks", {"SBE_STREAM_CHUNKS": "4"}),
            ("unordered chunks", {"SBE_STREAM_ORDERED": "0", "SBE_STREAM_CHUNKS": "16"}), ("not streamed", {"SBE_STREAM_RESULTS": "0"})]
for rep in range(3):
    for name, extra in SETTINGS:
        env = dict(os.environ, **extra)
        res = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, timeout=300)
        line = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-300:]
        print(f"{name:28s}: {line}", flush=True)
