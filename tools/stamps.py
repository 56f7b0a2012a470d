"""Diagnostic: summarise the in-kernel cycle stamps of k_mixture_tuple64 (build with -DSBE_STAMPS, run any
mixture launch with SBE_STAMPS_FILE set).  Stamps per wave: start, staged, sync1, built, sync2, gathered, end;
[7] = 1 for sub-row (light) blocks; [8], [9] = 100 MHz wall clock at start / end."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 12)
a = a[a[:, 0, 0] != 0]
names = ["stage", "sync1", "build", "sync2", "gather", "reduce"]
rt = a[:, :, 8:10].astype(np.int64)
cyc = (a[:, :, 6].astype(np.int64) - a[:, :, 0].astype(np.int64))
ns = (rt[:, :, 1] - rt[:, :, 0]) * 10.0
print(f"blocks {len(a)};  cycle counter ~ {np.median(cyc / np.maximum(ns, 1)):.3f} GHz;  kernel span (wall clock) "
      f"{(rt[:, :, 1].max() - rt[:, :, 0].min()) * 0.01:.2f} us; block start spread {(rt[:, :, 0].max() - rt[:, :, 0].min()) * 0.01:.2f} us")
for kind, sel in (("full-tile", a[:, 0, 7] == 0), ("sub-row", a[:, 0, 7] == 1)):
    b = a[sel]
    if not len(b):
        continue
    d = np.diff(b[:, :, :7].astype(np.int64), axis=2)
    print(f"-- {kind} blocks: {len(b)};  block duration {np.median((b[:, :, 9].astype(np.int64) - b[:, :, 8].astype(np.int64))) * 0.01:.2f} us median")
    for i, n in enumerate(names):
        x = d[:, :, i].ravel()
        print(f"{n:8s} mean {x.mean():9.0f} med {np.median(x):9.0f} min {x.min():9.0f} max {x.max():9.0f} cycles")
