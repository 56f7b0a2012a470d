#!/usr/bin/env python3
"""Same-box A/B of the batched group-tuple forms at the headline shape: k_mixture_tuple64 (vector-pipe gather) against
k_mixture_tuple_mfma (counts on the matrix pipe), kernel time from HIP event pairs around the launch, alternating runs.
    python tools/ab_mfma.py [--batches 256,512,1024,2048] [--reps 5]"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from sbayes_amd.engine import MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA           # noqa: E402
from sbayes_amd.synthetic import make_workload                                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="256,512,1024,2048")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--workload", default="headline")
    args = ap.parse_args()
    wl = make_workload(args.workload)
    batches = [int(b) for b in args.batches.split(",")]
    B = max(batches)
    import bench
    eng = bench.setup_engine(wl, B, 0)
    print(f"{B} states resident", flush=True)
    ref = {}
    try:
        for rep in range(args.reps):
            for b in batches:
                for name, kern in (("tuple64", MIXTURE_PACKED_TUPLE), ("mfma", MIXTURE_PACKED_TUPLE_MFMA)):
                    eng.set_option(kernel=kern)
                    eng.mixture_loglik_batch(0, b)                    # warm
                    eng.kernel_timing_start()
                    for _ in range(20):
                        eng.mixture_loglik_batch_async(0, b)
                    got = eng.fetch_results(0, b)
                    n, ms = eng.kernel_timing_stop()
                    if name == "tuple64":
                        ref[b] = got
                    else:
                        err = float(np.max(np.abs(got - ref[b]) / np.abs(ref[b])))
                        assert err <= 1e-10, err
                    print(f"B={b:5d} rep {rep} {name:8s} {ms * 1e3:8.2f} us per launch ({n} launches)  {eng.last_mixture_kernel()}", flush=True)
    finally:
        eng.close()


if __name__ == "__main__":
    main()
