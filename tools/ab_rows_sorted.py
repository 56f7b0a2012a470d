#!/usr/bin/env python3
"""Same-box A/B of the rows kernel at the stress shape: objects in their own order (weights read per observation) against
pattern-sorted objects (weights in registers, state bytes gathered).  Two engines in one process (the option is read at
creation), alternating runs, kernel time from HIP event pairs.   python tools/ab_rows_sorted.py [--batch 64] [--reps 3]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from sbayes_amd.synthetic import make_workload        # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--workload", default="stress")
    args = ap.parse_args()
    import bench
    wl = make_workload(args.workload)
    engines = {}
    for name, flag in (("plain", "0"), ("sorted", "2")):
        os.environ["SBE_ROWS_SORTED"] = flag
        engines[name] = bench.setup_engine(wl, args.batch, 0, kernel="packed_general")
    ref = None
    try:
        for rep in range(args.reps):
            for name, eng in engines.items():
                eng.mixture_loglik_batch(0, args.batch)
                eng.kernel_timing_start()
                for _ in range(20):
                    eng.mixture_loglik_batch_async(0, args.batch)
                got = eng.fetch_results(0, args.batch)
                n, ms = eng.kernel_timing_stop()
                if name == "plain":
                    ref = got
                else:
                    err = float(np.max(np.abs(got - ref) / np.abs(ref)))
                    assert err <= 1e-10, err
                print(f"rep {rep} {name:7s} {ms * 1e3:8.2f} us per launch of {args.batch} evals ({n} launches)  {eng.last_mixture_kernel()}", flush=True)
    finally:
        for eng in engines.values():
            eng.close()


if __name__ == "__main__":
    main()
