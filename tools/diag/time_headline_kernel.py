#!/usr/bin/env python3
"""Timing only (no parity gate: experiment builds may compute nothing): the headline launch's kernel time by HIP event pairs.
    SBAYES_AMD_LIB=... python tools/diag/time_headline_kernel.py [B] [iters]"""
import os
import sys
sys.path.insert(0, ".")
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
wl = bench.load_workload("headline")
eng = bench.setup_engine(wl, B, 0)
for _ in range(3):
    tot, k = eng.profile_mixture(0, B, iters)
print(os.environ.get("SBAYES_AMD_LIB", "default").split("/")[-1], {k: v for k, v in os.environ.items() if k.startswith("SBE_")},
      eng.last_mixture_kernel().split("<")[0], f"B={B} kernel {k * 1e3:.2f} us  loop {tot / iters * 1e3:.2f} us per launch")
eng.close()
