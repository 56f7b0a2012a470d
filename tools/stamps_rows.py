"""Diagnostic: in-kernel wall-clock stamps of k_mixture_rows (build with -DSBE_STAMPS, run a mixture launch with
SBE_STAMPS_FILE set): per block start / image staged / loop done / end on the 100 MHz clock."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 48)[:, :4].astype(np.int64)
a = a[a[:, 0] != 0]
t0 = a[:, 0].min()
print(f"blocks {len(a)}; kernel span {(a[:, 3].max() - t0) * 0.01:.2f} us; block start spread {(a[:, 0].max() - t0) * 0.01:.2f} us")
for name, d in (("stage", a[:, 1] - a[:, 0]), ("loop", a[:, 2] - a[:, 1]), ("tail", a[:, 3] - a[:, 2]), ("block", a[:, 3] - a[:, 0])):
    print(f"{name:6s} mean {d.mean() * 0.01:8.2f} med {np.median(d) * 0.01:8.2f} min {d.min() * 0.01:8.2f} max {d.max() * 0.01:8.2f} us")
