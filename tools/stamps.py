"""Diagnostic: summarise the in-kernel cycle stamps of k_mixture_tuple64 (build with -DSBE_STAMPS, run any
mixture launch with SBE_STAMPS_FILE set).  Stamps per wave: start, staged, sync1, built, sync2, gathered, end."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 8)
a = a[a[:, 0, 0] != 0]
names = ["stage", "sync1", "build", "sync2", "gather", "reduce"]
d = np.diff(a[:, :, :7].astype(np.int64), axis=2)
print("blocks", len(a))
for i, n in enumerate(names):
    x = d[:, :, i].ravel()
    print(f"{n:8s} mean {x.mean():9.0f} med {np.median(x):9.0f} min {x.min():9.0f} max {x.max():9.0f} cycles")
t0 = a[:, :, 0].min()
print("first start -> last end:", int(a[:, :, 6].max() - t0), "cycles; start spread:", int(a[:, :, 0].max() - t0))
