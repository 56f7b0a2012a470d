import sys, time, os
sys.path.insert(0, '.')
import numpy as np
import bench
wl = bench.load_workload("headline")
for B in (512, 4096):
    eng = bench.setup_engine(wl, B, 0, n_slots=2 * B)
    for thr in (None,):
        r = bench.changing_tables_leg(eng, wl, B, sweeps=6)
        print(B, r["evals_per_s"], r["ms_per_step"], r["kernel"], flush=True)
    eng.close()
