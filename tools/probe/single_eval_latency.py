#!/usr/bin/env python3
"""Host-synchronous single evals (sbe_mixture_loglik) and small async batches per shape: microseconds per call, median of
5 x 2000 calls.  One process per setting of the SBE_* experiment variables.   python tools/probe/single_eval_latency.py"""
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
import bench                                                      # noqa: E402

out = []
for name in ("cfg1", "south_america", "headline", "stress"):
    wl = bench.load_workload(name)
    eng = bench.setup_engine(wl, 8, 0)
    for _ in range(200):
        eng.mixture_loglik(0)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(2000):
            eng.mixture_loglik(0)
        ts.append((time.perf_counter() - t0) / 2000 * 1e6)
    tb = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(500):
            eng.mixture_loglik_batch(0, 8)
        tb.append((time.perf_counter() - t0) / 500 * 1e6)
    out.append(f"{name}: single {np.median(ts):.2f} us, batch of 8 (sync) {np.median(tb):.2f} us")
    eng.close()
env = {k: v for k, v in os.environ.items() if k.startswith("SBE_")}
print(env, "; ".join(out), flush=True)
