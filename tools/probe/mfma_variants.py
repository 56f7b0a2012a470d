#!/usr/bin/env python3
"""Kernel time of the matrix-pipe form at the headline shape, B = 2048, for the variant named by the environment
(SBE_MFMA_VARIANT / SBE_MFMA_FUSED / SBE_MFMA_WAVES); one process per variant.  Prints the median of 5 x 20 launches."""
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
eng.mixture_loglik_batch(0, B)
ts = []
for _ in range(5):
    eng.kernel_timing_start()
    for _ in range(20):
        eng.mixture_loglik_batch_async(0, B)
    got = eng.fetch_results(0, B)
    n, ms = eng.kernel_timing_stop()
    ts.append(ms * 1e3)
import hashlib
print(f"variant={os.environ.get('SBE_MFMA_VARIANT', '-')} fused={os.environ.get('SBE_MFMA_FUSED', '-')} waves={os.environ.get('SBE_MFMA_WAVES', '-')} "
      f"B={B}: {np.median(ts):.2f} us (min {min(ts):.2f})  digest {hashlib.sha1(np.ascontiguousarray(got).tobytes()).hexdigest()[:16]}", flush=True)
eng.close()
