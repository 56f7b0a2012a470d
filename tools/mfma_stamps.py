#!/usr/bin/env python3
"""In-kernel phase times of k_mixture_tuple_mfma (diagnostic build: build.sh -DSBE_MFMA_STAMPS): shader-clock stamps per
wave at block start (0), phase 0 done (1), past the barrier (2), first pass: counts done (3), epilogue done (4), second pass
(5, 6), wave end (7).  python tools/mfma_stamps.py [--batch 2048]"""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
OUT = REPO / "gpurun_out" / "mfma_stamps.bin"
os.environ["SBE_MFMA_STAMPS_FILE"] = str(OUT)
from sbayes_amd.engine import MIXTURE_PACKED_TUPLE_MFMA           # noqa: E402
from sbayes_amd.synthetic import make_workload                     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    args = ap.parse_args()
    import bench
    wl = make_workload("headline")
    eng = bench.setup_engine(wl, args.batch, 0, kernel="packed_tuple_mfma")
    for _ in range(5):
        eng.mixture_loglik_batch(0, args.batch)
    eng.close()
    st = np.fromfile(OUT, dtype=np.uint64).reshape(1024, 8, 16).astype(np.int64)
    n_blocks = int((st[:, 0, 0] != 0).sum())
    st = st[:n_blocks]
    t0 = st[:, :, 0].min(axis=1, keepdims=True)
    names = ["start", "phase0 done", "past barrier", "pass1 counts", "pass1 epilogue", "pass2 counts", "pass2 epilogue", "end"]
    print(f"{n_blocks} blocks; cycles since the block's first wave started (median over blocks and waves / p90):")
    names += ["p0: X asked", "p0: meta", "p0: logtab"]
    for k in (0, 8, 9, 10, 1, 2, 3, 4, 5, 6, 7):
        d = (st[:, :, k] - t0)
        ok = st[:, :, k] != 0
        if ok.any():
            lo, hi = d[:, :4][ok[:, :4]], d[:, 4:][ok[:, 4:]]
            print(f"  {names[k]:16s} {np.median(d[ok]):10.0f} {np.percentile(d[ok], 90):10.0f}   waves 0-3: {np.median(lo):8.0f}   waves 4-7: {np.median(hi):8.0f}")
    span = st[:, :, 7].max() - st[:, :, 0].min()
    print(f"first start to last end: {span} cycles")


if __name__ == "__main__":
    main()
