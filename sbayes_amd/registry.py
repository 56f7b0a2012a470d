"""Per-process registry of engines keyed by the resident feature block.

The reference's free functions receive the `features` array on every call
(compute_component_likelihood(features, ...), compute_effect_counts(features, ...)); the
engine that already holds that block in HBM is found by the array's buffer address and
shape.  Engines are created lazily in the process that uses them and are never pickled."""
from __future__ import annotations

import os
import warnings
import weakref

import numpy as np

from . import _proc
from .engine import Engine

_ENGINES: dict = {}          # per process: emptied in a fork()ed child (_forget_inherited), never shared
_LAST = [None]               # (feature array object, its engine) of the last get_engine call: identity fast path


def _key(features: np.ndarray):
    return (features.ctypes.data, features.shape, features.strides)


def _extent(key):
    """[lo, hi) byte range the elements of the (bool) array behind a registry key lie in."""
    lo = hi = key[0]
    for n, st in zip(key[1], key[2]):
        if n > 1:
            if st > 0:
                hi += (n - 1) * st
            else:
                lo += (n - 1) * st
    return lo, hi + 1


def get_engine(features, n_groups=None, n_slots=4, device=None, deferred_checks=True) -> Engine:
    """Engine holding `features`; created on first use.  `n_groups` (groups per mixture
    component) sizes the slot state; stateless calls work without it.  `deferred_checks` (applied when the engine is
    CREATED here): data checks of state-setting calls -- normalize's positive-sum assert, a source row that is not
    one-hot -- are reported by the next call that waits for the device instead of stalling the stream at the setter; the
    message names the call that supplied the data ("[deferred data check: raised by sbe_update_probs ...]").  The
    drop-in layer relies on it (a bind is several setters); pass False for the reference's own timing of the assert."""
    # the same array OBJECT as in the last call (the drop-in functions pass data.features.values every time): its
    # engine, unless that was closed or the caller names another component layout -- no key is built, nothing is hashed
    last = _LAST[0]
    if last is not None and last[0]() is features:        # (weakly held: the registry never keeps a feature block alive)
        eng = last[1]
        if getattr(eng, "_h", True) and (n_groups is None or list(n_groups) == eng.n_groups):
            return eng
    features = np.asarray(features)
    key = _key(features)
    entry = _ENGINES.get(key)
    if entry is not None:
        eng, ref = entry
        alive = ref() is not None if ref is not None else True
        if alive and (n_groups is None or list(n_groups) == eng.n_groups):
            if ref is not None:
                _LAST[0] = (ref, eng)
            return eng
        eng.close()
        del _ENGINES[key]
    if device is None:
        device = default_device()
    lo, hi = _extent(key)
    for okey, (_, oref) in _ENGINES.items():
        if not isinstance(okey[0], int):             # the featureless stand-ins of engine_for_features: no array behind them
            continue
        olo, ohi = _extent(okey)
        if (oref is None or oref() is not None) and olo < hi and lo < ohi:
            warnings.warn("sbayes_amd.registry: a second engine is created for another view of a feature block that "
                          f"already has one (shape {okey[1]} at {okey[0]:#x}, now {features.shape} at {key[0]:#x}); "
                          "pass the same array object (e.g. data.features.values once, not a fresh slice per call) to "
                          "share the resident copy", RuntimeWarning, stacklevel=2)
            break
    eng = Engine(features, list(n_groups) if n_groups is not None else [1], n_slots=n_slots, device=device)
    # the drop-in layer's state-setting calls (bind cache: changed rows, ids, weights, table rebuilds) never stall the
    # stream: a data check they raise (normalize's positive-sum assert, a source row that is not one-hot) is reported
    # by the next call that fetches a result -- in the same host function, a few lines later
    if deferred_checks and hasattr(eng, "set_option"):          # (test doubles without options)
        eng.set_option(deferred_checks=True)
    try:
        ref = weakref.ref(features)
    except TypeError:
        ref = None
    _ENGINES[key] = (eng, ref)
    if ref is not None:
        _LAST[0] = (ref, eng)
    return eng


_KNOWN = {}      # n_features -> (weakref to a feature block, n_groups) noted by Likelihood(...)


def default_device() -> int:
    """Device of this process: SBAYES_AMD_DEVICE, else LOCAL_RANK folded onto the visible devices (chains.device_for:
    ranks may outnumber GPUs in rehearsals on a one-GPU box), else 0."""
    if "SBAYES_AMD_DEVICE" in os.environ:
        return int(os.environ["SBAYES_AMD_DEVICE"])
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if local_rank == 0:
        return 0
    from .chains import device_for
    from .engine import device_count
    return device_for(local_rank, device_count())


def note_features(features, n_groups=None):
    """Remember a feature block (weakly) so that calls which do not carry it can find / create its engine."""
    features = np.asarray(features)
    try:
        _KNOWN[features.shape[1]] = (weakref.ref(features), list(n_groups) if n_groups else None)
    except TypeError:
        pass


def engine_for_features(n_features: int) -> Engine:
    """Engine for calls that carry neither the feature block nor a fixed row count
    (normalize_weights(weights, has_components), likelihood.py:171-190: has_components has N rows in update_weights
    but n_available rows in AlterCluster.compute_feature_weights_with_and_without, operators.py:1086).  The call needs
    nothing of the engine but F, its stream and its scratch memory: any live engine with that many features serves,
    else the engine of a block noted by Likelihood(...), else ONE featureless stand-in per F (never one per row
    count: a run would otherwise accumulate an engine, with its pinned arenas, for every value n_available takes)."""
    for eng, ref in _ENGINES.values():
        if eng.n_features == n_features and (ref is None or ref() is not None):
            return eng
    known = _KNOWN.get(n_features)
    if known is not None and known[0]() is not None:
        return get_engine(known[0](), known[1])
    key = ("features", n_features)
    if key not in _ENGINES:
        _ENGINES[key] = (Engine(np.zeros((1, n_features, 1), dtype=bool), [1], n_slots=1, device=default_device()), None)
    return _ENGINES[key][0]


def engine_for_observations(na_features, n_components, n_groups=None):
    """The live engine whose resident feature block has exactly the NA mask `na_features` (bool [N, F]) and
    `n_components` mixture components -- and, when `n_groups` is given, exactly that group layout (two datasets of one
    shape and one mask, e.g. no missing values, differ in their confounders' group counts: ADVICE r4) -- or None.  For calls that receive the NA mask but neither the feature block nor
    the model (GibbsSampleWeights.source_lh_by_feature(source, weights, na_features), operators.py:677-685): the shape
    alone does not identify a dataset -- two datasets of one shape differ in which observations count -- so the mask
    is compared with the engine's own (fetched once per engine; an array object verified once is recognised by
    identity afterwards, it is kept alive by the entry)."""
    na = np.asarray(na_features)
    if na.ndim != 2:
        return None
    for eng, ref in _ENGINES.values():
        if (eng.n_objects, eng.n_features) != na.shape or eng.n_components != n_components:
            continue
        if n_groups is not None and [int(g) for g in n_groups] != list(eng.n_groups):
            continue
        if ref is not None and ref() is None:
            continue
        seen = getattr(eng, "_na_verified", None)
        if seen is not None and seen is na and not na.flags.writeable:
            return eng                      # (a read-only array verified once cannot have changed: identity is enough)
        mask = getattr(eng, "_na_host", None)
        if mask is None:
            mask = eng._na_host = eng.na_values()
        if na.dtype == np.bool_ and np.array_equal(mask, na):
            eng._na_verified = na
            return eng
    return None


@_proc.on_fork_clear
def _forget_inherited():
    """In a fork()ed child: the engines in these tables are the parent's (their handles were nulled by
    _proc._after_fork_in_child, nothing is destroyed); the child starts with an empty registry."""
    _ENGINES.clear()
    _KNOWN.clear()
    _LAST[0] = None


def release_all():
    for eng, _ in list(_ENGINES.values()):
        eng.close()
    _ENGINES.clear()
    _KNOWN.clear()
    _LAST[0] = None
