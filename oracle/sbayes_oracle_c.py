"""ctypes face of oracle/sbayes_oracle_c.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as sbayes_oracle.py).

`mixture_loglik(...)` has the signature of `sbayes_oracle.mixture_loglik`; `build()` compiles the C file with gcc -O3 into
oracle/_build/ (git-ignored; `__graft_entry__.build()` calls it, so the built library travels to the GPU box with the snapshot)."""
from __future__ import annotations

import ctypes as ct
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
SRC = HERE / "sbayes_oracle_c.c"
OUT = HERE / "_build" / "libsbayes_oracle_c.so"
_LIB = None


def build(force=False) -> Path:
    if force or not OUT.exists() or OUT.stat().st_mtime < SRC.stat().st_mtime:
        OUT.parent.mkdir(parents=True, exist_ok=True)
        cmd = [os.environ.get("CC", "gcc"), "-O3", "-fPIC", "-shared", "-Wall", str(SRC), "-o", str(OUT), "-lm"]
        subprocess.run(cmd, check=True)
    return OUT


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = ct.CDLL(str(build()))
        _LIB.sbo_mixture_loglik.restype = ct.c_int
    return _LIB


def mixture_loglik(features, na_values, groups_by_component, counts_by_component, concentration_by_component, weights):
    """One uncached eval by the compiled C restatement; `na_values` is implied by the feature block."""
    feats = np.ascontiguousarray(features, dtype=np.bool_)
    n_obj, n_feat, n_states = feats.shape
    C = len(groups_by_component)
    groups = [np.ascontiguousarray(g, dtype=np.bool_) for g in groups_by_component]
    counts = [np.ascontiguousarray(c, dtype=np.float32) for c in counts_by_component]
    conc = [np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), counts[c].shape)) for c, a in enumerate(concentration_by_component)]
    w = np.ascontiguousarray(weights, dtype=np.float32)
    n_groups = np.array([g.shape[0] for g in groups], dtype=np.int32)
    ptrs = lambda arrs: (ct.c_void_p * C)(*[a.ctypes.data for a in arrs])       # noqa: E731
    out = ct.c_double(0.0)
    rc = _lib().sbo_mixture_loglik(ct.c_void_p(feats.ctypes.data), n_obj, n_feat, n_states, C, ct.c_void_p(n_groups.ctypes.data),
                                   ptrs(groups), ptrs(counts), ptrs(conc), ct.c_void_p(w.ctypes.data), ct.byref(out))
    if rc:
        raise RuntimeError(f"sbo_mixture_loglik failed ({rc})")
    return out.value
