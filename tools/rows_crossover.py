"""Where does k_mixture_rows (1024-thread blocks, long object ranges) beat k_mixture_v2 (256-thread blocks)?
Kernel time per launch (sbe_profile_mixture, HIP events) of both general forms over N, at two component layouts.
usage (GPU box): python tools/rows_crossover.py  ->  table on stdout"""
import sys

import numpy as np

sys.path.insert(0, ".")
from bench import setup_engine                                    # noqa: E402
from sbayes_amd.engine import MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2      # noqa: E402
from sbayes_amd.synthetic import make_workload                    # noqa: E402

for extra, label in (((), "C=2"), ((20, 20), "C=4")):
    for n_obj in (500, 1000, 2000, 3000, 5000):
        for n_feat, n_states in ((200, 10),):
            wl = make_workload("x", shape=(n_obj, n_feat, n_states, 10, extra, False))
            for batch in (8, 64, 256):
                eng = setup_engine(wl, batch, 0, kernel="packed_general")
                row = []
                for kern in (MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_V2):
                    eng.set_option(kernel=kern)
                    eng.mixture_loglik_batch(0, batch)
                    _, k_ms = eng.profile_mixture(0, batch, 20)
                    row.append((k_ms * 1e3, eng.last_mixture_kernel()))
                print(f"{label} N={n_obj:5d} F={n_feat} S={n_states} B={batch:4d}  rows {row[0][0]:8.1f} us   v2 {row[1][0]:8.1f} us   "
                      f"ratio {row[1][0] / row[0][0]:.2f}   [{row[0][1]} | {row[1][1]}]", flush=True)
                eng.close()
