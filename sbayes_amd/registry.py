"""Per-process registry of engines keyed by the resident feature block.

The reference's free functions receive the `features` array on every call
(compute_component_likelihood(features, ...), compute_effect_counts(features, ...)); the
engine that already holds that block in HBM is found by the array's buffer address and
shape.  Engines are created lazily in the process that uses them and are never pickled."""
from __future__ import annotations

import os
import weakref

import numpy as np

from .engine import Engine

_ENGINES: dict = {}


def _key(features: np.ndarray):
    return (features.ctypes.data, features.shape, features.strides)


def get_engine(features, n_groups=None, n_slots=4, device=None) -> Engine:
    """Engine holding `features`; created on first use.  `n_groups` (groups per mixture
    component) sizes the slot state; stateless calls work without it."""
    features = np.asarray(features)
    key = _key(features)
    entry = _ENGINES.get(key)
    if entry is not None:
        eng, ref = entry
        alive = ref() is not None if ref is not None else True
        if alive and (n_groups is None or list(n_groups) == eng.n_groups):
            return eng
        eng.close()
        del _ENGINES[key]
    if device is None:
        device = int(os.environ.get("SBAYES_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    eng = Engine(features, list(n_groups) if n_groups is not None else [1], n_slots=n_slots, device=device)
    try:
        ref = weakref.ref(features)
    except TypeError:
        ref = None
    _ENGINES[key] = (eng, ref)
    return eng


_KNOWN = {}      # (n_objects, n_features) -> (weakref to a feature block, n_groups) noted by Likelihood(...)


def note_features(features, n_groups=None):
    """Remember a feature block (weakly) so that shape-only calls can find / create its engine."""
    features = np.asarray(features)
    try:
        _KNOWN[(features.shape[0], features.shape[1])] = (weakref.ref(features), list(n_groups) if n_groups else None)
    except TypeError:
        pass


def engine_for_shape(n_objects: int, n_features: int) -> Engine:
    """Engine for calls that do not carry the feature block (normalize_weights(weights, has_components),
    likelihood.py:171-190): an existing engine of that shape, else the engine of a block noted by
    Likelihood(...), else a featureless stand-in of that shape (the call only needs N and F)."""
    for eng, ref in _ENGINES.values():
        if eng.n_objects == n_objects and eng.n_features == n_features and (ref is None or ref() is not None):
            return eng
    known = _KNOWN.get((n_objects, n_features))
    if known is not None and known[0]() is not None:
        return get_engine(known[0](), known[1])
    key = ("shape", n_objects, n_features)
    if key not in _ENGINES:
        _ENGINES[key] = (Engine(np.zeros((n_objects, n_features, 1), dtype=bool), [1], n_slots=1,
                                device=int(os.environ.get("SBAYES_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))), None)
    return _ENGINES[key][0]


def release_all():
    for eng, _ in list(_ENGINES.values()):
        eng.close()
    _ENGINES.clear()
    _KNOWN.clear()
