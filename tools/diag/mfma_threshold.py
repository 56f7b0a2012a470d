#!/usr/bin/env python3
"""GPU box: where the matrix-pipe form overtakes k_mixture_tuple64 at the headline shape (the default switched at 512 slots per launch before round 6, at 320 since: per
launch, SBE_MFMA_MIN_BATCH): kernel time per launch of both forms at small batches."""
import sys
sys.path.insert(0, ".")
import bench
from sbayes_amd.engine import MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA

wl = bench.load_workload("headline")
eng = bench.setup_engine(wl, 1024, 0)
for b in (int(x) for x in (sys.argv[1:] or "32 64 128 192 256 384 512 768 1024".split())):
    row = [b]
    for k in (MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA):
        eng.set_option(kernel=k)
        eng.profile_mixture(0, b, 30)
        tot, kern = eng.profile_mixture(0, b, 100)
        row += [round(kern * 1e3, 2), round(tot / 100 * 1e3, 2)]
    print("B=%4d  tuple64 kernel %7.2f us (loop %7.2f)   mfma kernel %7.2f us (loop %7.2f)" % tuple(row), flush=True)
eng.close()
