#!/usr/bin/env python3
"""Kernel time of the matrix-pipe form at the headline shape (HIP event pairs, median of 5 x 20 launches) for the library
named by SBAYES_AMD_LIB (A/B of two builds on one box) and whatever SBE_* experiment variables are set; one process per
build.    python tools/probe/mfma_kernel_time.py [B]"""
import hashlib
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
from sbayes_amd.engine import MIXTURE_PACKED_TUPLE_MFMA          # noqa: E402
from sbayes_amd.synthetic import make_workload                    # noqa: E402
import bench                                                      # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
wl = make_workload("headline")
eng = bench.setup_engine(wl, B, 0)
eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
first = eng.mixture_loglik_batch(0, B)
ts = []
for _ in range(5):
    eng.kernel_timing_start()
    for _ in range(20):
        eng.mixture_loglik_batch_async(0, B)
    got = eng.fetch_results(0, B)
    n, ms = eng.kernel_timing_stop()
    ts.append(ms * 1e3)
    assert np.array_equal(got, first), "results changed between launches"
env = {k: v for k, v in os.environ.items() if k.startswith("SBE_") or k == "SBAYES_AMD_LIB"}
print(f"{env} B={B}: {np.median(ts):.2f} us (min {min(ts):.2f})  digest {hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16]}", flush=True)
eng.close()
