"""Drop-in for sbayes/model/likelihood.py on the MI355X engine.

Same public surface as the reference module:
  class Likelihood(data, shapes, prior)       likelihood.py:22-101
  compute_component_likelihood(...)           likelihood.py:104-133
  compute_component_likelihood_exact(...)     likelihood.py:136-150
  update_weights(sample, caching=True)        likelihood.py:153-168
  normalize_weights(weights, has_components)  likelihood.py:171-190
Control flow (what is cached, what `changed_groups` is) is host Python exactly as in the
reference; every array operation runs on the GPU through the C ABI.  There is no CPU
fallback: without the HIP library / a GPU these functions raise.
"""
from __future__ import annotations

import collections
import weakref
from types import SimpleNamespace

import numpy as np

from .counts import forget_jump_state, recalculate_feature_counts
from . import _fast, registry
from .binding import _bind_slot
from .engine import GroupOverlapError
from .registry import get_engine


class LazyBlock:
    """Stand-in for a cache array that was allocated but never computed.

    Every reference sample carries `cache.component_likelihoods`, a float64 [n_objects, n_features, n_components] block that
    ModelCache.__init__ allocates with np.empty (sbayes/sampling/state.py:398-400) and Sample.copy() copies with every proposal
    (`assign_from`: state.py:317-321, :475) -- 3.2 MB, 250 us per MCMC step at the 1000 x 200 x 2 shape -- whether or not anything
    ever computes it.  Under patch.install(operators=True) nothing in the sampling loop does (the operators that read
    likelihood_per_component run their device forms), so a block that is NOT CURRENT for its sample is replaced by this object: shape and
    dtype of the array it stands for, immutable, `copy()` of it is itself (what CacheNode.assign_from does to it), pickles as two
    small tuples.  likelihood_per_component materialises it (np.empty, exactly what it replaced) before the first write; anything
    else that asks it for values gets a RuntimeError, not garbage."""

    __slots__ = ("shape", "dtype")

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), np.dtype(dtype)

    ndim = property(lambda self: len(self.shape))

    def materialize(self):
        return np.empty(self.shape, dtype=self.dtype)

    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self

    def __reduce__(self):
        return LazyBlock, (self.shape, self.dtype.str)

    # values are not there: reading or writing elements fails loudly (a write into a temporary would be lost silently -- e.g. the
    # reference's own likelihood_per_component after patch.uninstall() on a sample that still carries the stand-in)
    def _no_values(self, *args, **kwargs):
        raise RuntimeError("this cache block was dropped by sbayes_amd's lean samples (likelihood.LazyBlock): it holds no values; "
                           "sbayes_amd.conditionals.likelihood_per_component(model, sample) rebuilds it, "
                           "SBAYES_AMD_LEAN_SAMPLES=0 keeps the arrays")

    __array__ = __getitem__ = __setitem__ = _no_values

    def __repr__(self):
        return f"LazyBlock(shape={self.shape}, dtype={self.dtype})"


# set by patch.install(operators=True): Likelihood.__call__ replaces a sample's uncomputed component_likelihoods block (LazyBlock)
LEAN_SAMPLES = False
_LEAN_MIN_BYTES = 1 << 16


def _lean_sample(sample):
    node = getattr(sample.cache, "component_likelihoods", None)
    if node is None:
        return
    value = node._value
    if type(value) is np.ndarray and value.nbytes >= _LEAN_MIN_BYTES and _fast.node_outdated(node):
        # never computed, or computed for an earlier state (the initialiser's samples) and not current for this one: nothing in the
        # loop will bring it up to date.  clear() (state.py:304-308) marks every group as changed, so whoever does ask later
        # (likelihood_per_component) recomputes the whole block instead of patching rows of the array dropped here
        node.clear()
        node._value = LazyBlock(value.shape, value.dtype)


class Likelihood:
    """Collapsed Dirichlet-categorical likelihood of the sBayes mixture model."""

    def __init__(self, data, shapes, prior):
        self.features = data.features.values
        self.confounders = data.confounders
        self.shapes = shapes
        self.prior = prior
        self.source_index = {"clusters": 0}
        for i, conf in enumerate(self.confounders, start=1):
            self.source_index[conf] = i
        self._na_features = None
        self._engine = None
        self._bind_model = SimpleNamespace(prior=prior)      # what binding._bind_slot reads of a model
        self._freeze_static_inputs()
        registry.note_features(self.features, self.n_groups)
        # patch.install(operators=True) serves SourcePrior.__call__ (prior.py:573-611) from the device; that method has no
        # way back to the model, so the model's likelihood leaves a note on the prior's SourcePrior instance
        source_prior = getattr(prior, "source_prior", None)
        if source_prior is not None:
            try:
                source_prior._sbayes_amd_owner = SimpleNamespace(prior=prior, likelihood=self)
            except AttributeError:
                pass

    def _freeze_static_inputs(self):
        """Inputs of the path that nothing in the sampler ever writes -- the confounders' group matrices
        (load_data.py:139-184) and the cluster-effect prior's concentration tables (prior.py:440-455) -- are marked
        read-only, the way the reference itself freezes its static confounding-effect tables (prior.py:322-323).  The bind
        cache then recognises them by identity (binding._same) instead of comparing their content on every call, and
        an in-place write -- which would silently desynchronise the resident copy -- raises instead."""
        arrays = [getattr(conf, "group_assignment", None) for conf in self.confounders.values()]
        cluster_prior = getattr(self.prior, "prior_cluster_effect", None)
        arrays += [getattr(cluster_prior, "concentration_array", None), getattr(cluster_prior, "uniform_concentration_array", None)]
        for arr in arrays:
            if type(arr) is np.ndarray and arr.flags.owndata and arr.flags.writeable:
                arr.setflags(write=False)

    # device handles are per process: never pickled, re-created lazily (mcmc_setup.py:299, model.py:53)
    def __getstate__(self):
        from . import patch
        state = dict(self.__dict__)
        state["_na_features"] = None
        state["_engine"] = None
        state["_call_memo"] = None
        state["_fast_call"] = None
        state["_dynamic_priors_memo"] = None
        state["_sbayes_amd_patch"] = patch.installed()      # how the pickling process had sBayes patched, if at all
        return state

    def __setstate__(self, state):
        """Unpickled in another process (an MC3 worker, a Pool worker): under the spawn / forkserver start methods that
        process is a fresh interpreter whose sBayes is not patched yet, and the model is the first thing it receives
        (mcmc_setup.py:299, :554) -- re-install the patch the sender ran under, before the worker builds its chain."""
        how = state.pop("_sbayes_amd_patch", None)
        self.__dict__.update(state)
        self._engine = None
        self._freeze_static_inputs()                        # (flags do not travel through pickle)
        if how is not None:
            from . import patch
            cur = patch.installed()
            if cur is None or (how["operators"] and not cur["operators"]) or (how.get("gibbs_source") and not cur.get("gibbs_source")):
                patch.install(operators=how["operators"], gibbs_source=bool(how.get("gibbs_source")))
        registry.note_features(self.features, self.n_groups)

    @property
    def n_groups(self):
        return [self.shapes.n_clusters] + [conf.n_groups for conf in self.confounders.values()]

    @property
    def engine(self):
        # the registry lookup (key of the feature block, weak reference, layout check) once per engine: a live handle
        # created by this process is reused; a closed one (registry.release_all, fork: _proc) is looked up again
        eng = self.__dict__.get("_engine")
        if eng is None or not getattr(eng, "_h", True):
            eng = self._engine = get_engine(self.features, self.n_groups)
        return eng

    @property
    def na_features(self):
        """bool [n_objects, n_features]: observations with no state set (likelihood.py:40)."""
        if self._na_features is None:
            self._na_features = self.engine.na_values()
        return self._na_features

    def __call__(self, sample, caching=True) -> float:
        if LEAN_SAMPLES:
            _lean_sample(sample)
        if caching and _fast._h is not None:
            # the whole caching path in one native call (csrc/sbe_pyhost.c: likelihood_call -- the cache-node protocol of
            # compute_lh_clusters / compute_lh_confounder below, component after component, same order, same sums); the values
            # come from _all_group_logliks.  NotImplemented: a node form it does not serve; None: overlapping groups -- the
            # Python form below serves the call (nothing it would do differently has been done)
            fast = self.__dict__.get("_fast_call")
            if fast is None:
                fast = self._fast_call = (["clusters", *self.confounders], self._all_group_logliks)
            forget_jump_state()
            res = _fast._h.likelihood_call(self, sample, fast[0], self.engine.group_offsets, fast[1])
            if res is not NotImplemented and res is not None:
                return res
        if not caching:
            recalculate_feature_counts(self.features, sample)
        # one bind and one fetch per call: the components are asked one after another (likelihood.py:58-63) for the SAME
        # sample, so the per-group values of all of them are taken once (_group_logliks) and kept for the rest of the call
        self._call_memo = [sample, None, caching]
        forget_jump_state()                  # (counts.py: a hint that lives for one MCMC step)
        try:
            log_lh = 0.0
            log_lh += self.compute_lh_clusters(sample, caching=caching)
            for conf in self.confounders:
                log_lh += self.compute_lh_confounder(sample, conf, caching=caching)
            return log_lh
        finally:
            self._call_memo = None

    def _all_group_logliks(self, sample, slot=0):
        """float64 [sum of G_c]: the collapsed log-likelihood of every group of every component of `sample`, inside a caching
        Likelihood.__call__ (native form) -- one bind, one device call, the source prior the prior is about to ask for taken along
        (see _group_logliks, which this is the first-request part of); None when the groups overlap (no resident form)."""
        eng = self.engine
        sp_cache = conditionals.source_prior_wanted(self.prior, sample)
        try:
            _bind_slot(eng, self._bind_model, sample, slot, with_source=sp_cache is not None)
        except GroupOverlapError:
            return None
        bound = getattr(eng, "_bound", None)
        entry = bound.get(slot) if bound is not None else None
        lh_all = entry.get("lh_all") if entry is not None else None
        if lh_all is None:
            if sp_cache is not None:
                lh_all, per_object = eng.collapsed_and_source_prior(slot)
                conditionals.store_source_prior_ahead(sp_cache, sample, per_object)
            else:
                lh_all = eng.collapsed_loglik_all(slot)
            if entry is not None:
                entry["lh_all"] = lh_all
        return lh_all

    def _group_logliks(self, sample, component, groups, slot=0):
        """float64 per listed group of one component: float32-summed Dirichlet-categorical log-pdf (a7/a8) from the
        RESIDENT counts and concentration tables of the slot `sample` is bound to -- the bind sends the count rows of
        the groups that changed since the slot was last bound (those are the `groups` asked for here), nothing else
        goes up, G_c doubles come back."""
        eng = self.engine
        memo = self.__dict__.get("_call_memo")
        if memo is not None and memo[0] is sample and memo[1] is not None:     # (later components of the same __call__)
            return memo[1][int(eng.group_offsets[component]) + np.asarray(groups)]
        # inside Model.__call__ (likelihood, then prior, of the same sample: sbayes/model/model.py:47-51) the source prior the
        # prior is about to ask for comes back with the collapsed values: one launch instead of two -- and ONE bind, which then
        # takes the sample's source rows along
        sp_cache = None
        if memo is not None and memo[0] is sample and memo[2]:
            sp_cache = conditionals.source_prior_wanted(self.prior, sample)
        try:
            _bind_slot(eng, self._bind_model, sample, slot, with_source=sp_cache is not None)
        except GroupOverlapError:
            # groups that overlap have no resident form; the collapsed likelihood needs none -- it is a function of the
            # count and concentration tables alone (likelihood.py:65-101): the stateless device call
            name = sample.component_names[component]
            counts = np.asarray(sample.feature_counts[name].value)[np.asarray(groups)]
            if component == 0:
                conc = np.asarray(self.prior.prior_cluster_effect.concentration_array, dtype=np.float64)
            else:
                conc = np.asarray(self.prior.prior_confounding_effects[name].concentration_array(sample),
                                  dtype=np.float64)[np.asarray(groups)]
            return eng.dirichlet_logpdf(counts, conc, per_group=True)[1]
        # Likelihood.__call__ asks component after component (likelihood.py:58-63): every component's groups are
        # evaluated by the first request (one launch, one synchronisation) and kept on the slot's bind entry until a
        # count row or a concentration table changes
        entry = eng._bound.get(slot) if getattr(eng, "_bound", None) is not None else None
        lh_all = entry.get("lh_all") if entry is not None else None
        if lh_all is None:
            if sp_cache is not None:
                lh_all, per_object = eng.collapsed_and_source_prior(slot)
                conditionals.store_source_prior_ahead(sp_cache, sample, per_object)
            else:
                lh_all = eng.collapsed_loglik_all(slot)
            if entry is not None:
                entry["lh_all"] = lh_all
        if memo is not None and memo[0] is sample:
            memo[1] = lh_all
        return lh_all[int(eng.group_offsets[component]) + np.asarray(groups)]

    def compute_lh_clusters(self, sample, caching=True) -> float:
        cache = sample.cache.group_likelihoods["clusters"]
        if caching and not cache.is_outdated():
            # every input still has the version the node was computed from: what_changed() would list nothing and edit()
            # would re-record the same versions (state.py:232-270) -- the cached values are the answer
            return cache.value.sum()
        with cache.edit() as lh:
            changed = cache.what_changed("counts", caching=caching)
            if len(changed) > 0:
                lh[changed] = self._group_logliks(sample, 0, changed)
        return cache.value.sum()

    def compute_lh_confounder(self, sample, conf, caching=True) -> float:
        cache = sample.cache.group_likelihoods[conf]
        if caching and not cache.is_outdated():             # (as above; a changed hyperprior input makes the node outdated too)
            return cache.value.sum()
        conf_prior = self.prior.prior_confounding_effects[conf]
        with cache.edit() as lh:
            hyperprior_has_changed = conf_prior.any_dynamic_priors and cache.ahead_of("universal_counts")
            changed = cache.what_changed("counts", caching=caching and not hyperprior_has_changed)
            if len(changed) > 0:
                lh[changed] = self._group_logliks(sample, self.source_index[conf], changed)
        return cache.value.sum()


def compute_component_likelihood(features, probs, groups, changed_groups, out, na_value=0.0):
    """In-place partial update of `out` [n_objects, n_features] (float64, any strides) with the
    likelihood of every observation under one mixture component; returns `out`.  `na_value` (not in
    the reference signature, default = the reference's result 0) lets likelihood_per_component have
    the device write its NA value 1 directly."""
    return get_engine(features).component_lh(probs, groups, changed_groups, out, na_value)


def compute_component_likelihood_exact(features, probs, groups, changed_groups, out):
    """Leave-one-out form: probs[i] is a per-member table [n_in_group, F, S].  Each member's row
    is one a1 gather with its own table; served by the same device kernel, one group at a time."""
    eng = get_engine(features)
    groups = np.asarray(groups)
    out[~groups.any(axis=0), :] = 0.0
    n_obj = groups.shape[1]
    for i in changed_groups:
        members = np.flatnonzero(groups[i])
        if members.size == 0:
            continue
        tables = np.asarray(probs[i])
        # pseudo-groups: one per member (its own table) + one unchanged group holding everyone
        # else, so that the kernel's "no group -> 0" rule leaves the other rows untouched
        pseudo = np.zeros((members.size + 1, n_obj), dtype=bool)
        pseudo[np.arange(members.size), members] = True
        pseudo[-1] = ~groups[i]
        tables = np.concatenate([tables, np.zeros((1,) + tables.shape[1:], dtype=tables.dtype)])
        eng.component_lh(tables, pseudo, np.arange(members.size), out)
    return out


# normalize_weights for a FEW rows (the operators' update_weights(sample)[object_subset], operators.py:520, 756-757, 819-842):
# a row depends on the weights and on its own has_components pattern only, so each pattern's row is computed on the
# device ONCE per weights content and gathered from this memo afterwards -- a cluster step asks for the same objects'
# rows before and after the move, and the weights only change at AlterWeights steps.  (Device results, kept: not a host
# computation -- a pattern the memo does not hold is a device call.)
_ROW_MEMO = collections.OrderedDict()      # (weights shape, weights bytes) -> {pattern bytes: float32 [F, C] device result}
_ROW_MEMO_WEIGHTS = 4                      # weight arrays kept (MC3 chains alternate between theirs)
_ROW_MEMO_ROWS = 64                        # row requests up to this size go through the memo


def _memo_for(w):
    """The pattern memo of one weights content (float32, C-contiguous [F, C])."""
    key = (w.shape, w.tobytes())
    memo = _ROW_MEMO.get(key)
    if memo is None:
        memo = _ROW_MEMO[key] = {}
        if len(_ROW_MEMO) > _ROW_MEMO_WEIGHTS:
            _ROW_MEMO.popitem(last=False)
    else:
        _ROW_MEMO.move_to_end(key)
    return memo


def _rows_from_memo(memo, w, hc, eng_of):
    """float32 [n, F, C]: the rows of normalize_weights(w, hc) from `memo` (pattern bytes -> device result [F, C]); patterns it
    does not hold are ONE device call (`eng_of()` -> the engine), with every other non-empty pattern riding along."""
    n = hc.shape[0]
    if n == 1:
        k = hc.tobytes()
        row = memo.get(k)
        if row is not None:
            return row[None].copy()
        keys = [k]
    else:
        keys = [row.tobytes() for row in hc]
    missing = [k for k in dict.fromkeys(keys) if k not in memo]
    if missing:
        n_comp = w.shape[1]
        want = list(missing)
        if n_comp <= 4:                    # every other non-empty pattern rides along: the step's next request needs no call
            want += [p for p in (bytes((b >> c) & 1 for c in range(n_comp)) for b in range(1, 1 << n_comp))
                     if p not in memo and p not in missing]
        rows = eng_of().normalize_weights(w, np.frombuffer(b"".join(want), dtype=np.bool_).reshape(len(want), n_comp))
        for p, r in zip(want, rows):
            memo[p] = r
    out = np.empty((n,) + w.shape, dtype=np.float32)
    for i, k in enumerate(keys):
        out[i] = memo[k]
    return out


def _pattern_rows(eng, w, hc):
    return _rows_from_memo(_memo_for(w), w, hc, lambda: eng)


def normalize_weights(weights, has_components, features=None):
    """float32 [n_rows, n_features, n_components]: weights masked by has_components and renormalised over
    components; has_components may have any number of rows (the reference also calls it with
    has_components[available], operators.py:1086).  `features` (optional) selects the engine explicitly."""
    has_components = np.asarray(has_components)
    if features is not None:
        eng = get_engine(features)
    else:
        eng = registry.engine_for_features(np.shape(weights)[0])
    if has_components.ndim == 2 and 0 < has_components.shape[0] <= _ROW_MEMO_ROWS:
        w = np.ascontiguousarray(weights, dtype=np.float32)
        if w.ndim == 2 and has_components.shape[1] == w.shape[1]:
            return _pattern_rows(eng, w, np.ascontiguousarray(has_components, dtype=np.bool_))
    return eng.normalize_weights(weights, has_components)


class NormalizedWeights(np.lib.mixins.NDArrayOperatorsMixin):
    """What update_weights(sample) returns: the float32 [n_objects, n_features, n_components] array of
    normalize_weights(weights, has_components) (likelihood.py:153-190), materialised LAZILY.

    Inside an MCMC step the reference only ever reads a few object rows of it -- `update_weights(sample)[object_subset]`
    (operators.py:520, 756-757, 819-842, 1790; prior.py:603) -- so indexing with an object subset (bool mask, index
    array, list, slice or int along axis 0) computes exactly those rows on the device (normalize_weights of
    has_components[subset]: bit-identical to the rows of the full array, each row depends only on its own
    has_components pattern) and the N * F * C * 4 bytes of the full array never cross PCIe.  Every other use
    (np.asarray, arithmetic, ufuncs, any other attribute or index form) materialises the full array once, on the
    device, and behaves like it.  The object is immutable: copies share it."""

    __array_priority__ = 100

    def __init__(self, weights, has_components, features=None, owner=None):
        self._weights = np.array(weights, dtype=np.float32)              # private snapshots: evaluation may come later
        self._has_components = np.array(has_components, dtype=bool)
        self._features = features
        self._full = None
        self._memo = None                                                 # (the pattern memo of these weights: _rows_from_memo)
        try:                                                              # the sample these weights were derived from
            self._owner = weakref.ref(owner) if owner is not None else None
        except TypeError:
            self._owner = None

    # Samples are pickled (MC3 pipes, StateDumper: sbayes/mcmc_setup.py:320-324, sampling/loggers.py:426-442): the
    # weights travel as their two small inputs; the sample reference and the feature block stay behind
    def __getstate__(self):
        return {"_weights": self._weights, "_has_components": self._has_components, "_features": None, "_full": None, "_owner": None, "_memo": None}

    def __setstate__(self, state):
        self.__dict__.update(state)

    def _rebind(self, sample):
        try:
            cur = self._owner() if self._owner is not None else None
            if cur is not sample:
                self._owner = weakref.ref(sample)
        except TypeError:
            self._owner = None

    def sample_if_current(self):
        """The sample update_weights() derived this array from, if it still has these weights and this has_components
        (device forms that work on the sample's resident state -- source_lh_by_feature -- use it instead of the
        [N, F, C] array); None otherwise."""
        sample = self._owner() if self._owner is not None else None
        if sample is None:
            return None
        if not (np.array_equal(np.asarray(sample.weights.value, dtype=np.float32), self._weights)
                and np.array_equal(sample.cache.has_components.value, self._has_components)):
            return None
        return sample

    # -- what the consumers look at without needing values
    @property
    def shape(self):
        return (self._has_components.shape[0],) + self._weights.shape

    dtype = np.dtype(np.float32)
    ndim = 3

    def __len__(self):
        return self._has_components.shape[0]

    def __copy__(self):
        return self

    def __deepcopy__(self, memo):
        return self

    def materialize(self):
        if self._full is None:
            full = normalize_weights(self._weights, self._has_components, self._features)
            full.flags.writeable = False
            self._full = full
        return self._full

    def __array__(self, dtype=None, copy=None):
        full = self.materialize()
        return full if dtype is None or dtype == full.dtype else full.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x.materialize() if isinstance(x, NormalizedWeights) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(x.materialize() if isinstance(x, NormalizedWeights) else x for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        def conv(x):
            if isinstance(x, NormalizedWeights):
                return x.materialize()
            if isinstance(x, (list, tuple)):
                return type(x)(conv(y) for y in x)
            return x
        return func(*conv(args), **{k: conv(v) for k, v in kwargs.items()})

    def _rows(self, idx):
        """has_components rows an axis-0 index selects, or None when `idx` is anything else."""
        if type(idx) is np.ndarray:
            if idx.ndim == 1 and (idx.dtype == np.bool_ or idx.dtype.kind in "iu"):
                return self._has_components[idx]
            return None
        if isinstance(idx, (int, np.integer)):
            return None                                                   # (drops the axis: rare, use the full array)
        if isinstance(idx, slice):
            return self._has_components[idx]
        if isinstance(idx, (list, np.ndarray)):
            arr = np.asarray(idx)
            if arr.ndim == 1 and (arr.dtype == np.bool_ or np.issubdtype(arr.dtype, np.integer)):
                return self._has_components[arr]
        return None

    def _engine(self):
        if self._features is not None:
            return get_engine(self._features)
        return registry.engine_for_features(self._weights.shape[0])

    def __getitem__(self, idx):
        if self._full is None:
            rows = self._rows(idx)
            if rows is not None:
                n = rows.shape[0]
                if n == 0:
                    return np.zeros((0,) + self._weights.shape, dtype=np.float32)
                if 2 * n <= self._has_components.shape[0]:
                    w = self._weights
                    if n <= _ROW_MEMO_ROWS and w.ndim == 2 and rows.ndim == 2 and rows.shape[1] == w.shape[1]:
                        # a few rows: each is its has_components pattern's row, from the memo of these weights (found once)
                        memo = self.__dict__.get("_memo")
                        if memo is None:
                            memo = self._memo = _memo_for(w)
                        return _rows_from_memo(memo, w, rows, self._engine)
                    return normalize_weights(w, rows, self._features)
        return self.materialize()[idx]

    def __getattr__(self, name):                                          # everything else an ndarray offers
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __repr__(self):
        return f"NormalizedWeights(shape={self.shape}, materialized={self._full is not None})"


def update_weights(sample, caching=True, features=None):
    """Normalised mixture weights of `sample`, cached on (has_components, weights) versions (likelihood.py:153-168);
    returned as a lazily materialised array (NormalizedWeights)."""
    cache = sample.cache.weights_normalized
    if (not caching) or _fast.node_outdated(cache):
        value = NormalizedWeights(sample.weights.value, sample.cache.has_components.value, features, owner=sample)
        _fast.node_update_value(cache, value)
        return value
    value = cache.value
    if type(value) is NormalizedWeights:
        # the node travels from sample to sample (Sample.copy -> CacheNode.assign_from shares the immutable value), so the sample
        # it was first computed for may be gone: whoever asks now is the sample it belongs to (sample_if_current still compares
        # weights and has_components).  Without this the device form of source_lh_by_feature lost its sample in ~6 % of the
        # GibbsSampleWeights proposals and fell back to the reference expression on the materialised [N, F, C] array (4 ms at
        # 1000 x 200 x 2): round 6.
        value._rebind(sample)
    return value


from . import conditionals  # noqa: E402  (conditionals imports this module: bound here, after everything it needs is defined)
