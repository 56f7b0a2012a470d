"""Diagnostic: in-kernel wall-clock stamps of k_mixture_rows (build with -DSBE_STAMPS, run a mixture launch with
SBE_STAMPS_FILE set): per block start / image staged / loop done / end on the 100 MHz clock; per wave: loop end and
the ballot of its "bad product" flags."""
import sys
import numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 48)
raw = raw[raw[:, 0] != 0]
a = raw[:, :4].astype(np.int64)
t0 = a[:, 0].min()
span = (a[:, 3].max() - t0) * 0.01
print(f"blocks {len(a)}; kernel span {span:.2f} us; block start spread {(a[:, 0].max() - t0) * 0.01:.2f} us")
for name, d in (("stage", a[:, 1] - a[:, 0]), ("loop", a[:, 2] - a[:, 1]), ("tail", a[:, 3] - a[:, 2]), ("block", a[:, 3] - a[:, 0])):
    print(f"{name:6s} mean {d.mean() * 0.01:8.2f} med {np.median(d) * 0.01:8.2f} min {d.min() * 0.01:8.2f} max {d.max() * 0.01:8.2f} us")
w = raw[:, 4:20].astype(np.int64)
if w.any():
    skew = (w.max(axis=1) - w.min(axis=1)) * 0.01
    print(f"per-wave loop end: skew inside a block mean {skew.mean():.2f} max {skew.max():.2f} us; "
          f"last wave after wave 0: mean {((w.max(axis=1) - w[:, 0]) * 0.01).mean():.2f} us")
    bad = raw[:, 20:36]
    print(f"waves with a bad-product lane: {(bad != 0).sum()} of {bad.size}")
busy = (a[:, 3] - a[:, 0]).sum() * 0.01
print(f"sum of block lifetimes / (256 CUs x span) = {busy / (256 * span):.3f};  staging share of block lifetime {((a[:, 1] - a[:, 0]).sum() / (a[:, 3] - a[:, 0]).sum()):.3f}")
