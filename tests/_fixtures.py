"""Helpers to load the golden fixtures (tests/golden/*.npz|json) -- test infrastructure."""
from __future__ import annotations

import hashlib
import json
import zlib
from pathlib import Path
from types import SimpleNamespace

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def sha(a) -> str:
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def load_npz(name):
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    n_comp = len(meta["groups"])
    fx = SimpleNamespace(
        name=name, meta=meta, z=z,
        features=z["features"], states_per_feature=z["states_per_feature"],
        weights=z["weights"], source=z["source"],
        groups=[z[f"groups_{i}"] for i in range(n_comp)],
        conc=[z[f"conc_{i}"] for i in range(n_comp)],
        counts=[z[f"counts_{i}"] for i in range(n_comp)],
        group_lh=[z[f"group_lh_{i}"] for i in range(n_comp)],
        probs=[z[f"probs_{i}"] for i in range(n_comp)],
        dcl=[z[f"dcl_{i}"] for i in range(n_comp)],
        n_comp=n_comp,
    )
    fx.na_values = ~fx.features.any(axis=-1)
    return fx


def load_json(name):
    with open(GOLDEN / f"{name}.json") as fh:
        return json.load(fh)


def load_trace(name):
    z = np.load(GOLDEN / f"{name}_trace.npz", allow_pickle=False)
    k, n = (int(v) for v in z["clusters_shape"])
    sshape = tuple(int(v) for v in z["source_shape"])
    n_steps = z["clusters"].shape[0]

    def clusters(i):
        return np.unpackbits(z["clusters"][i])[: k * n].reshape(k, n).astype(bool)

    def source(i):
        return np.unpackbits(z["source"][i])[: int(np.prod(sshape))].reshape(sshape).astype(bool)

    return SimpleNamespace(z=z, n_steps=n_steps, clusters=clusters, source=source,
                           weights=z["weights"], last_lh=z["last_lh"], mixture_ll=z["mixture_ll"],
                           lh_sha=z["lh_sha"], group_lh=z["group_lh"], operator=z["operator"])
