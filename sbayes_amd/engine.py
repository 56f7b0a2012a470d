"""Engine: thin NumPy-facing wrapper of the C-ABI handle (include/sbe_engine.h).

Shape / dtype validation happens here, before the call crosses the ABI (SURVEY.md 8(b)
"Errors"); every non-zero return code becomes a RuntimeError carrying sbe_last_error().
Device handles are created lazily per process and are never pickled (MC3 workers and
`copy(model)` re-create them: sbayes/mcmc_setup.py:299, sbayes/model/model.py:53-54).
"""
from __future__ import annotations

import ctypes as ct
import os

import numpy as np

from . import _fast, _lib, _proc

MIXTURE_PACKED, MIXTURE_ONEHOT, MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_TUPLE, MIXTURE_ONEHOT_GENERAL = 0, 1, 2, 3, 4
MIXTURE_PACKED_TUPLE_LDS = 5
MIXTURE_PACKED_V2 = 6
MIXTURE_PACKED_TUPLE_MFMA = 7
LOG_PER_OBS, LOG_PRODUCT = 0, 1
_OPT_KERNEL, _OPT_LOG, _OPT_DEFERRED = 1, 2, 3


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"sbe error {code}: {message}")
        self.code = code


class GroupOverlapError(EngineError):
    """SBE_ERR_DATA from sbe_set_groups / sbe_step*: an object is in several groups of one component.  Resident slot
    state keeps one group per object and component; the reference defines such input differently for counts (once per
    group, counts.py:28-30) and for a1 (last written group wins, likelihood.py:126-130), so it is rejected rather than
    collapsed.  The drop-in counts / collapsed-likelihood functions catch this and use the stateless device calls."""


def _c(a, dtype):
    """C-contiguous view/copy with the exact dtype the ABI expects."""
    if type(a) is np.ndarray and a.dtype == dtype and a.flags.c_contiguous:
        return a
    a = np.asarray(a)
    if a.dtype == np.bool_ and dtype == np.uint8:
        a = a.view(np.uint8) if a.flags.c_contiguous else np.ascontiguousarray(a).view(np.uint8)
    return np.ascontiguousarray(a, dtype=dtype)


def _as(a, dtype):
    """C-contiguous ndarray of exactly `dtype`: `a` itself when it already is one (the drop-in layer hands over arrays it
    has just built in that form; np.ascontiguousarray costs a microsecond per argument to find that out)."""
    if type(a) is np.ndarray and a.dtype == dtype and a.flags.c_contiguous:
        return a
    return np.ascontiguousarray(a, dtype=dtype)


# the buffer address as a plain int (every array argument of the ABI is declared c_void_p, which takes one):
# ndarray.ctypes.data_as builds two helper objects per call, 2 us apiece on the hosts measured, __array_interface__ a
# dict (0.9 us) -- with three to seven array arguments a quarter of a latency-bound call; the extension's buffer-protocol
# accessor (sbayes_amd/_fast.py) costs 0.08 us
_ptr = _fast.addr


def device_count() -> int:
    _proc.check_usable()                 # a fork()ed child of a HIP-initialised parent may not touch the runtime
    lib = _lib.load()
    n = ct.c_int(0)
    _proc.mark_hip_touched()             # marked on the ATTEMPT: hipGetDeviceCount initialises the runtime whatever it returns
    lib.sbe_device_count(ct.byref(n))
    return n.value


class Engine:
    """One resident one-hot feature block + `n_slots` sample states on one GPU."""

    def __init__(self, features, n_groups, n_slots=2, device=0):
        _proc.check_usable()             # ForkedWithHipError in a fork()ed child of a HIP-initialised parent
        self._lib = _lib.load()
        self._h = ct.c_void_p()
        self._pid = None                 # pid of the process the handle lives in (set once sbe_create succeeded)
        self.h2d_bytes = self.d2h_bytes = self.n_calls = 0
        features = np.asarray(features)
        if features.ndim != 3:
            raise ValueError(f"features must be [n_objects, n_features, n_states], got shape {features.shape}")
        if features.dtype != np.bool_ and features.dtype != np.uint8:
            raise TypeError(f"features must be bool (one-hot), got {features.dtype}")
        self.n_objects, self.n_features, self.n_states = (int(v) for v in features.shape)
        self.n_groups = [int(g) for g in n_groups]
        self.n_components = len(self.n_groups)
        self.n_slots = int(n_slots)
        self.device = int(device)
        feats = _c(features, np.uint8)
        ng = np.asarray(self.n_groups, dtype=np.int32)
        # marked on the ATTEMPT, not on success: a create that fails after the runtime came up (out of memory,
        # a bad shape behind hipSetDevice) has initialised HIP all the same, and a child forked afterwards must not be
        # taken for a fresh process
        _proc.mark_hip_touched()
        rc = self._lib.sbe_create(ct.byref(self._h), self.device, self.n_objects, self.n_features,
                                  self.n_states, self.n_components,
                                  ng.ctypes.data_as(ct.POINTER(ct.c_int32)), self.n_slots, self._i(feats))
        if rc != 0:
            msg = self._lib.sbe_last_error(None)
            self._h = ct.c_void_p()
            raise EngineError(rc, msg.decode() if msg else "sbe_create failed")
        self._pid = os.getpid()
        _proc.register_engine(self)
        self.group_offsets = np.concatenate([[0], np.cumsum(self.n_groups)]).astype(int)
        self.n_groups_total = int(self.group_offsets[-1])
        self._deferred = False
        self._bound = {}                    # slot -> what binding._bind_slot last uploaded there
        self._bound_conc = {}               # component -> concentration table last uploaded
        self._bound_unif = None             # uniform concentration last uploaded (binding._bind_uniform)
        self._mirror = {}                   # slot -> host mirrors of the slot's counts / source (delta uploads)

    # -- plumbing -----------------------------------------------------------------------------
    # -- bind cache (sbayes_amd.conditionals._bind_slot): what a slot was last bound to; every method that changes a
    #    slot's state drops the slot's entry, so a cached entry always describes the device state ------------------
    def _touch(self, slot):
        self._bound.pop(slot, None)
        self._mirror.pop(slot, None)

    # PCIe accounting at the ABI boundary: bytes of every caller buffer the library reads (h2d) / writes (d2h).
    # sampler_replay in bench.py reports them per MCMC step (SURVEY.md 8(b) "What crosses PCIe per step").
    def _i(self, a):
        self.h2d_bytes += a.nbytes
        return _ptr(a)

    def _o(self, a):
        self.d2h_bytes += a.nbytes
        return _ptr(a)

    def traffic(self, reset=False):
        """(bytes handed to the library, bytes written back by it, ABI calls) since creation / the last reset."""
        t = (self.h2d_bytes, self.d2h_bytes, self.n_calls)
        if reset:
            self.h2d_bytes = self.d2h_bytes = self.n_calls = 0
        return t

    def _check(self, rc):
        self.n_calls += 1
        if rc != 0:
            msg = self._lib.sbe_last_error(self._h)
            msg = msg.decode() if msg else "?"
            if rc == 4 and " is in groups " in msg:
                raise GroupOverlapError(rc, msg)
            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            if self._pid == os.getpid():         # (a handle that reached another process by any road is never destroyed there)
                self._lib.sbe_destroy(self._h)
            self._h = ct.c_void_p()

    def _forget(self, lib_face):
        """After fork(), in the child (_proc._after_fork_in_child): drop the inherited handle WITHOUT sbe_destroy -- its
        device memory, pinned arenas and stream belong to the parent -- and make every later call on this object raise."""
        self._h = ct.c_void_p()
        self._lib = lib_face
        self._bound = {}
        self._bound_conc = {}
        self._bound_unif = None
        self._mirror = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __getstate__(self):
        raise TypeError("Engine holds device memory and is not picklable; re-create it in the new process")

    def info(self):
        inf = _lib.SbeInfo()
        self._check(self._lib.sbe_get_info(self._h, ct.byref(inf)))
        return {k: (getattr(inf, k).decode() if k == "device_name" else getattr(inf, k)) for k, _ in inf._fields_}

    def set_option(self, kernel=None, log_mode=None, deferred_checks=None, step_form=None, step_derive=None, fuse_tables=None):
        if fuse_tables is not None:
            self._check(self._lib.sbe_set_option(self._h, 6, int(bool(fuse_tables))))  # SBE_OPT_FUSE_TABLES
        if step_derive is not None:
            self._check(self._lib.sbe_set_option(self._h, 5, int(step_derive)))    # SBE_OPT_STEP_DERIVE
        if step_form is not None:
            self._check(self._lib.sbe_set_option(self._h, 4, int(step_form)))      # SBE_OPT_STEP_FORM
        if deferred_checks is not None:
            self._check(self._lib.sbe_set_option(self._h, _OPT_DEFERRED, int(bool(deferred_checks))))
            self._deferred = bool(deferred_checks)          # mirrored: users of a shared engine restore what they found
        if kernel is not None:
            self._check(self._lib.sbe_set_option(self._h, _OPT_KERNEL, int(kernel)))
        if log_mode is not None:
            self._check(self._lib.sbe_set_option(self._h, _OPT_LOG, int(log_mode)))

    def sync(self):
        self._check(self._lib.sbe_sync(self._h))

    def na_values(self):
        out = np.empty((self.n_objects, self.n_features), dtype=np.bool_)
        self._check(self._lib.sbe_get_na(self._h, self._o(out)))
        return out

    # -- a1 -------------------------------------------------------------------------------------
    def component_lh(self, probs, groups, changed_groups, out, na_value=0.0):
        """compute_component_likelihood (likelihood.py:104-133): in-place partial update of the
        (possibly strided) float64 view `out` [N, F].  `na_value` is written for NA observations of
        the rows this call writes (0.0 = the literal a1 result, 1.0 = likelihood_per_component's)."""
        probs = np.asarray(probs)
        groups = np.asarray(groups)
        n_groups = groups.shape[0]
        if groups.shape != (n_groups, self.n_objects):
            raise ValueError(f"groups must be [n_groups, {self.n_objects}], got {groups.shape}")
        if probs.shape != (n_groups, self.n_features, self.n_states):
            raise ValueError(f"probs must be {(n_groups, self.n_features, self.n_states)}, got {probs.shape}")
        if not isinstance(out, np.ndarray) or out.dtype != np.float64 or out.shape != (self.n_objects, self.n_features):
            raise ValueError("out must be a float64 ndarray (view) of shape [n_objects, n_features]")
        if not out.flags.writeable:
            raise ValueError("out is read-only")
        f64 = probs.dtype == np.float64
        p = _c(probs, np.float64 if f64 else np.float32)
        g = _c(groups.astype(bool, copy=False), np.uint8)
        ch = _as(changed_groups, np.int64).reshape(-1)
        self._check(self._lib.sbe_component_lh(self._h, self._i(p), int(f64), n_groups, self._i(g), self._i(ch), ch.size,
                                               _ptr(out), out.strides[0], out.strides[1], float(na_value)))
        self.d2h_bytes += out.size * 8              # (rows the call leaves untouched are not written; counted as the upper bound)
        return out

    # -- slot state -----------------------------------------------------------------------------
    def set_groups(self, slot, component, groups):
        g = np.asarray(groups)
        if g.shape != (self.n_groups[component], self.n_objects):
            raise ValueError(f"groups of component {component} must be {(self.n_groups[component], self.n_objects)}, got {g.shape}")
        g = _c(g.astype(bool, copy=False), np.uint8)
        self._touch(slot)
        self._check(self._lib.sbe_set_groups(self._h, slot, component, self._i(g)))

    def set_group_ids(self, slot, component, ids):
        ids = _as(ids, np.int32)
        if ids.shape != (self.n_objects,):
            raise ValueError("ids must be [n_objects]")
        self._touch(slot)
        self._check(self._lib.sbe_set_group_ids(self._h, slot, component, self._i(ids)))

    def get_group_ids(self, slot, component):
        """The slot's ids of one component as the DEVICE holds them: int32 [N], -1 = in no group (inverse of set_group_ids)."""
        out = np.empty(self.n_objects, dtype=np.int32)
        self._check(self._lib.sbe_get_group_ids(self._h, slot, component, self._o(out)))
        return out

    def set_source(self, slot, source):
        s = np.asarray(source)
        if s.shape != (self.n_objects, self.n_features, self.n_components):
            raise ValueError(f"source must be {(self.n_objects, self.n_features, self.n_components)}, got {s.shape}")
        s = _c(s.astype(bool, copy=False), np.uint8)
        self._touch(slot)
        self._check(self._lib.sbe_set_source(self._h, slot, self._i(s)))

    def set_source_rows(self, slot, objects, rows):
        objects = _as(objects, np.int32).reshape(-1)
        rows = np.asarray(rows)
        if rows.shape != (objects.size, self.n_features, self.n_components):
            raise ValueError("rows must be [len(objects), n_features, n_components]")
        rows = _c(rows.astype(bool, copy=False), np.uint8)
        self._touch(slot)
        self._check(self._lib.sbe_set_source_rows(self._h, slot, self._i(objects), objects.size, self._i(rows)))

    def set_slot_delta(self, slot, groups_component=0, groups=None, count_idx=None, count_rows=None, update_probs=False,
                       source_objects=None, source_rows=None):
        """Several state-setting calls of one bind as ONE launch (sbe_set_slot_delta): set_groups(slot, groups_component, groups)
        when `groups` is given, set_counts_rows(slot, count_idx, count_rows, update_probs) when `count_idx` is, set_source_rows(slot,
        source_objects, source_rows) when `source_objects` is -- in that order, same checks and results."""
        comp, g = int(groups_component), None
        if groups is not None:
            g = np.asarray(groups)
            if g.shape != (self.n_groups[comp], self.n_objects):
                raise ValueError(f"groups of component {comp} must be {(self.n_groups[comp], self.n_objects)}, got {g.shape}")
            g = _c(g.astype(bool, copy=False), np.uint8)
        gi = r = None
        n_rows = 0
        if count_idx is not None:
            gi = _as(count_idx, np.int32).reshape(-1)
            r = _c(count_rows, np.float32)
            if r.shape != (gi.size, self.n_features, self.n_states):
                raise ValueError(f"rows must be [{gi.size}, {self.n_features}, {self.n_states}], got {r.shape}")
            n_rows = gi.size
        objs = sr = None
        n_src = 0
        if source_objects is not None:
            objs = _as(source_objects, np.int32).reshape(-1)
            sr = np.asarray(source_rows)
            if sr.shape != (objs.size, self.n_features, self.n_components):
                raise ValueError("rows must be [len(objects), n_features, n_components]")
            sr = _c(sr.astype(bool, copy=False), np.uint8)
            n_src = objs.size
        self._touch(slot)
        self._check(self._lib.sbe_set_slot_delta(self._h, slot, comp, self._i(g) if g is not None else None,
                                                 self._i(gi) if n_rows else None, n_rows, self._i(r) if n_rows else None,
                                                 1 if update_probs else 0, self._i(objs) if n_src else None, n_src,
                                                 self._i(sr) if n_src else None))

    def get_source_rows(self, slot, objects):
        """bool [n, F, C]: the listed objects' rows of the slot's source."""
        objects = _as(objects, np.int32).reshape(-1)
        rows = np.empty((objects.size, self.n_features, self.n_components), dtype=np.uint8)
        self._check(self._lib.sbe_get_source_rows(self._h, slot, self._i(objects), objects.size, self._o(rows)))
        return rows.view(bool)

    def recount(self, slot, component=-1):
        self._touch(slot)
        self._check(self._lib.sbe_recount(self._h, slot, component))

    def update_counts(self, slot_new, slot_old, objects):
        """update_feature_counts (counts.py:55-95); returns bool[G_total] of changed groups."""
        objects = _as(objects, np.int32).reshape(-1)
        changed = np.zeros(self.n_groups_total, dtype=np.uint8)
        self._touch(slot_new)
        self._check(self._lib.sbe_update_counts(self._h, slot_new, slot_old, self._i(objects), objects.size, self._o(changed)))
        return changed.astype(bool)

    def accumulate_counts(self, slot, objects, sign):
        objects = _as(objects, np.int32).reshape(-1)
        changed = np.zeros(self.n_groups_total, dtype=np.uint8)
        self._touch(slot)
        self._check(self._lib.sbe_accumulate_counts(self._h, slot, self._i(objects), objects.size, int(sign), self._o(changed)))
        return changed.astype(bool)

    def set_counts(self, slot, component, counts):
        c = _c(counts, np.float32)
        if c.shape != (self.n_groups[component], self.n_features, self.n_states):
            raise ValueError("bad counts shape")
        self._touch(slot)
        self._check(self._lib.sbe_set_counts(self._h, slot, component, self._i(c)))

    def get_counts(self, slot, component):
        out = np.empty((self.n_groups[component], self.n_features, self.n_states), dtype=np.float32)
        self._check(self._lib.sbe_get_counts(self._h, slot, component, self._o(out)))
        return out

    def get_counts_all(self, slot):
        """The float32 [G_c, F, S] count tables of every component, one call (views of one array)."""
        out = np.empty((self.n_groups_total, self.n_features, self.n_states), dtype=np.float32)
        self._check(self._lib.sbe_get_counts_all(self._h, slot, self._o(out)))
        off = self.group_offsets
        return tuple(out[int(off[c]):int(off[c + 1])] for c in range(self.n_components))

    def set_concentration(self, component, conc):
        conc = _c(conc, np.float64)
        fs = (self.n_features, self.n_states)
        if conc.shape == fs:
            per_group = 0
        elif conc.shape == (self.n_groups[component],) + fs:
            per_group = 1
        else:
            raise ValueError(f"concentration of component {component} must be {fs} or [G]+{fs}, got {conc.shape}")
        self._bound.clear()                 # tables of every slot depend on it
        self._bound_conc.pop(component, None)
        self._check(self._lib.sbe_set_concentration(self._h, component, self._i(conc), per_group))

    def update_probs(self, slot, component, temperature=None, prior_temperature=None, unif_counts=None):
        """probs = normalize(counts [/T] + prior['] ) on the device (conditionals.py:105-122, 175-179).  `component`:
        one index, or several (any iterable of indices): their tables in one call."""
        t = float(temperature) if temperature is not None else 0.0
        tp = float(prior_temperature) if prior_temperature is not None else 0.0
        u = None
        if prior_temperature is not None:
            if unif_counts is None:
                raise AssertionError("unif_counts required with prior_temperature (conditionals.py:114)")
            u = _c(unif_counts, np.float64)
            if u.shape != (self.n_features, self.n_states):
                raise ValueError("unif_counts must be [n_features, n_states]")
        if temperature is not None or prior_temperature is not None:
            self._touch(slot)               # tempered tables are not the ones the bind cache vouches for
        if isinstance(component, (int, np.integer)):
            self._check(self._lib.sbe_update_probs(self._h, slot, int(component), t, tp, self._i(u) if u is not None else None))
            return
        mask = 0
        for c in component:
            if not 0 <= int(c) < self.n_components:
                raise ValueError(f"component {c} out of range")
            mask |= 1 << int(c)
        self._check(self._lib.sbe_update_probs_mask(self._h, slot, mask, t, tp, self._i(u) if u is not None else None))

    def set_probs(self, slot, component, probs):
        p = _c(probs, np.float32)
        if p.shape != (self.n_groups[component], self.n_features, self.n_states):
            raise ValueError("bad probs shape")
        self._touch(slot)
        self._check(self._lib.sbe_set_probs(self._h, slot, component, self._i(p)))

    def get_probs(self, slot, component):
        out = np.empty((self.n_groups[component], self.n_features, self.n_states), dtype=np.float32)
        self._check(self._lib.sbe_get_probs(self._h, slot, component, self._o(out)))
        return out

    def set_weights(self, slot, weights):
        w = _c(weights, np.float32)
        if w.shape != (self.n_features, self.n_components):
            raise ValueError(f"weights must be {(self.n_features, self.n_components)}, got {w.shape}")
        self._touch(slot)
        self._check(self._lib.sbe_set_weights(self._h, slot, self._i(w)))

    def get_weights(self, slot):
        """The slot's resident mixture weights as set: float32 [F, C]."""
        out = np.empty((self.n_features, self.n_components), dtype=np.float32)
        self._check(self._lib.sbe_get_weights(self._h, slot, self._o(out)))
        return out

    def weights_normalized(self, slot):
        out = np.empty((self.n_objects, self.n_features, self.n_components), dtype=np.float32)
        self._check(self._lib.sbe_get_weights_normalized(self._h, slot, self._o(out)))
        return out

    # -- dense outputs ---------------------------------------------------------------------------
    def likelihood_per_component(self, slot, out=None):
        if out is None:
            out = np.empty((self.n_objects, self.n_features, self.n_components), dtype=np.float64)
        assert out.flags.c_contiguous and out.dtype == np.float64
        self._check(self._lib.sbe_likelihood_per_component(self._h, slot, self._o(out)))
        return out

    def likelihood_per_component_exact(self, slot):
        out = np.empty((self.n_objects, self.n_features, self.n_components), dtype=np.float64)
        self._check(self._lib.sbe_likelihood_per_component_exact(self._h, slot, self._o(out)))
        return out

    def observation_lh(self, slot):
        out = np.empty((self.n_objects, self.n_features), dtype=np.float64)
        self._check(self._lib.sbe_observation_lh(self._h, slot, self._o(out)))
        return out

    # -- north-star scalar -------------------------------------------------------------------------
    def mixture_loglik(self, slot=0) -> float:
        out = ct.c_double(0.0)
        self._check(self._lib.sbe_mixture_loglik(self._h, slot, ct.byref(out)))
        return out.value

    def mixture_loglik_batch(self, first_slot, n):
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.sbe_mixture_loglik_batch(self._h, first_slot, n, self._o(out)))
        return out

    def mixture_loglik_batch_async(self, first_slot, n):
        self._check(self._lib.sbe_mixture_loglik_batch_async(self._h, first_slot, n))

    def fetch_results(self, first_slot, n):
        out = np.empty(n, dtype=np.float64)
        self._check(self._lib.sbe_fetch_results(self._h, first_slot, n, self._o(out)))
        return out

    def collapsed_loglik(self, slot, component, per_feature=False):
        g = self.n_groups[component]
        per_group = np.empty(g, dtype=np.float64)
        pf = np.empty((g, self.n_features), dtype=np.float32) if per_feature else None
        self._check(self._lib.sbe_collapsed_loglik(self._h, slot, component, self._o(per_group),
                                                   self._o(pf) if pf is not None else None))
        return (per_group, pf) if per_feature else per_group

    def collapsed_loglik_all(self, slot):
        """float64 [G_total]: the collapsed per-group log-likelihoods of every component in one call and one
        synchronisation (Likelihood.__call__(caching=False) = their sum, likelihood.py:47-63)."""
        out = np.empty(self.n_groups_total, dtype=np.float64)
        self._check(self._lib.sbe_collapsed_loglik_all(self._h, slot, self._o(out)))
        return out

    # -- stateless forms of the reference's free functions -----------------------------------------
    def normalize_tables(self, counts, concentration, temperature=None, prior_temperature=None, unif_counts=None):
        """normalize(counts [/T] + prior['], axis=-1) -> float32 [G, F, S] (util.py:990-1007)."""
        counts = _c(counts, np.float32)
        fs = (self.n_features, self.n_states)
        if counts.ndim != 3 or counts.shape[1:] != fs:
            raise ValueError(f"counts must be [G, {fs[0]}, {fs[1]}], got {counts.shape}")
        conc = _c(concentration, np.float64)
        if conc.shape == fs:
            per_group = 0
        elif conc.shape == counts.shape:
            per_group = 1
        else:
            raise ValueError(f"concentration must be {fs} or {counts.shape}, got {conc.shape}")
        t = float(temperature) if temperature is not None else 0.0
        tp = float(prior_temperature) if prior_temperature is not None else 0.0
        u = None
        if prior_temperature is not None:
            assert unif_counts is not None
            u = _c(np.broadcast_to(unif_counts, counts.shape)[0] if np.ndim(unif_counts) == 3 else unif_counts, np.float64)
        out = np.empty(counts.shape, dtype=np.float32)
        self._check(self._lib.sbe_normalize_tables(self._h, self._i(counts), counts.shape[0], self._i(conc), per_group, t, tp,
                                                   self._i(u) if u is not None else None, self._o(out)))
        return out

    def dirichlet_logpdf(self, counts, concentration, per_group=False):
        """dirichlet_categorical_logpdf (util.py:1373-1394) for [G, F, S] counts -> float32 [G, F]
        (and the float32-summed float64 [G] when per_group)."""
        counts = _c(counts, np.float32)
        squeeze = counts.ndim == 2
        if squeeze:
            counts = counts[None]
        fs = (self.n_features, self.n_states)
        conc = _c(concentration, np.float64)
        if conc.shape == fs:
            pg = 0
        elif conc.shape == counts.shape:
            pg = 1
        else:
            raise ValueError(f"concentration must be {fs} or {counts.shape}, got {conc.shape}")
        pf = np.empty(counts.shape[:2], dtype=np.float32)
        g = np.empty(counts.shape[0], dtype=np.float64) if per_group else None
        self._check(self._lib.sbe_dirichlet_logpdf(self._h, self._i(counts), counts.shape[0], self._i(conc), pg, self._o(pf),
                                                   self._o(g) if g is not None else None))
        pf = pf[0] if squeeze else pf
        return (pf, g) if per_group else pf

    def effect_counts(self, group_assignment, source_is_component, object_subset=None):
        """compute_effect_counts (counts.py:10-32) -> float32 [G, F, S]."""
        g = np.asarray(group_assignment)
        if g.ndim != 2 or g.shape[1] != self.n_objects:
            raise ValueError(f"group_assignment must be [G, {self.n_objects}], got {g.shape}")
        g = _c(g.astype(bool, copy=False), np.uint8)
        m = np.asarray(source_is_component)
        if m.shape != (self.n_objects, self.n_features):
            raise ValueError("source_is_component must be [n_objects, n_features]")
        m = _c(m.astype(bool, copy=False), np.uint8)
        out = np.empty((g.shape[0], self.n_features, self.n_states), dtype=np.float32)
        if object_subset is None:
            objs, n_sub = None, -1
        else:
            objs = _as(object_subset, np.int32).reshape(-1)
            n_sub = objs.size
        self._check(self._lib.sbe_effect_counts(self._h, self._i(g), g.shape[0], self._i(m),
                                                self._i(objs) if objs is not None and n_sub > 0 else None, n_sub, self._o(out)))
        return out

    def normalize_weights(self, weights, has_components):
        """normalize_weights (likelihood.py:171-190) -> float32 [n_rows, F, C]; has_components may have any number
        of rows (all objects, or has_components[available] as in operators.py:1086)."""
        w = _c(weights, np.float32)
        hc = np.asarray(has_components)
        if w.ndim != 2 or w.shape[0] != self.n_features or hc.ndim != 2 or hc.shape[1] != w.shape[1]:
            raise ValueError("weights must be [n_features, C] and has_components [n_rows, C]")
        hc = _c(hc.astype(bool, copy=False), np.uint8)
        out = np.empty((hc.shape[0], self.n_features, w.shape[1]), dtype=np.float32)
        self._check(self._lib.sbe_normalize_weights(self._h, self._i(w), w.shape[1], self._i(hc), hc.shape[0], self._o(out)))
        return out

    def cluster_marginals(self, slot, table, objects, prior_temperature=1.0):
        """float64 [2, n]: log marginal likelihood of each listed object outside (z=0) / inside (z=1)
        the cluster whose candidate effect table is `table` [F, S] (operators.py:1035-1095)."""
        t = _c(table, np.float32).reshape(-1, self.n_states)
        if t.shape != (self.n_features, self.n_states):
            raise ValueError(f"table must be [{self.n_features}, {self.n_states}] (or [1, F, S])")
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((2, objs.size), dtype=np.float64)
        self._check(self._lib.sbe_cluster_marginals(self._h, slot, self._i(t), self._i(objs), objs.size,
                                                    float(prior_temperature), self._o(out)))
        return out

    def jump_lh(self, slot, pconf, p_source, p_target, objects, prior_temperature=1.0):
        """float64 [2, n]: log "stay" / log "jump" likelihood of each listed member of the source cluster
        (ClusterJump.get_jump_lh, operators.py:1679-1722).  pconf: float32 [G_total - K, F, S] tempered tables of every
        confounder group, p_source / p_target: float32 [F, S] (or [1, F, S]) tempered tables of the two clusters."""
        fs = (self.n_features, self.n_states)
        n_conf = self.n_groups_total - self.n_groups[0]
        pc = _c(pconf, np.float32).reshape((-1,) + fs) if n_conf else np.zeros((0,) + fs, dtype=np.float32)
        if pc.shape[0] != n_conf:
            raise ValueError(f"pconf must hold the {n_conf} confounder groups' tables, got {pc.shape}")
        ps, pt = _c(p_source, np.float32).reshape(fs), _c(p_target, np.float32).reshape(fs)
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((2, objs.size), dtype=np.float64)
        self._check(self._lib.sbe_jump_lh(self._h, slot, self._i(pc) if n_conf else None, self._i(ps), self._i(pt), self._i(objs),
                                          objs.size, float(prior_temperature), self._o(out)))
        return out

    def source_lh_by_feature(self, slot):
        """float32 [F]: per-feature log-likelihood of the slot's source assignment under its normalised weights
        (GibbsSampleWeights.source_lh_by_feature, operators.py:677-685)."""
        out = np.empty(self.n_features, dtype=np.float32)
        self._check(self._lib.sbe_source_lh_by_feature(self._h, slot, self._o(out)))
        return out

    def source_posterior(self, slot, objects, temperature=1.0, prior_temperature=1.0):
        """float32 [n, F, C]: posterior of the source assignment of the listed objects' observations
        (GibbsSampleSource.calculate_source_posterior, operators.py:554-574)."""
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((objs.size, self.n_features, self.n_components), dtype=np.float32)
        self._check(self._lib.sbe_source_posterior(self._h, slot, self._i(objs), objs.size, float(temperature),
                                                   float(prior_temperature), self._o(out)))
        return out

    def sample_source(self, slot, dst_slot, objects, z, temperature=1.0, prior_temperature=1.0, from_prior=False,
                      return_selected=False):
        """Draw new source assignments for the listed objects on the device (GibbsSampleSource._propose,
        operators.py:518-528, with sample_categorical's uniforms `z` [n, F] supplied by the caller) and
        write them into `dst_slot`'s source.  Returns log_q (float), or (log_q, p_selected float32
        [n, F]) with return_selected."""
        objs = _as(objects, np.int32).reshape(-1)
        zz = None
        if z is not None:                        # None: the engine's own Philox stream (set_rng)
            zz = _c(z, np.float64).reshape(objs.size, -1)
            if zz.shape != (objs.size, self.n_features):
                raise ValueError(f"z must be [{objs.size}, {self.n_features}]")
        sel = np.empty((objs.size, self.n_features), dtype=np.float32) if return_selected else None
        log_q = ct.c_double()
        self._touch(dst_slot)
        self._check(self._lib.sbe_sample_source(self._h, slot, dst_slot, self._i(objs), objs.size, float(temperature),
                                                float(prior_temperature), int(bool(from_prior)),
                                                self._i(zz) if zz is not None else None, ct.byref(log_q),
                                                self._o(sel) if return_selected else None))
        return (log_q.value, sel) if return_selected else log_q.value

    def set_rng(self, seed, draw=0):
        """Key and draw counter of the engine's Philox4x32-10 stream (sample_source with z=None)."""
        self._check(self._lib.sbe_set_rng(self._h, int(seed) & (2 ** 64 - 1), int(draw) & (2 ** 64 - 1)))

    def get_rng(self):
        seed, draw = ct.c_uint64(), ct.c_uint64()
        self._check(self._lib.sbe_get_rng(self._h, ct.byref(seed), ct.byref(draw)))
        return seed.value, draw.value

    def test_philox(self, ctr_key):
        ck = _c(ctr_key, np.uint32).reshape(-1, 6)
        out = np.empty((ck.shape[0], 4), dtype=np.uint32)
        self._check(self._lib.sbe_test_philox(self._h, self._i(ck), ck.shape[0], self._o(out)))
        return out

    def source_logprob(self, slot, src_slot, objects, temperature=1.0, prior_temperature=1.0, from_prior=False,
                       return_selected=False):
        """sum log p_slot[source of src_slot] over the listed objects' observations (log_q_back,
        operators.py:544-550)."""
        objs = _as(objects, np.int32).reshape(-1)
        sel = np.empty((objs.size, self.n_features), dtype=np.float32) if return_selected else None
        log_q = ct.c_double()
        self._check(self._lib.sbe_source_logprob(self._h, slot, src_slot, self._i(objs), objs.size, float(temperature),
                                                 float(prior_temperature), int(bool(from_prior)), ct.byref(log_q),
                                                 self._o(sel) if return_selected else None))
        return (log_q.value, sel) if return_selected else log_q.value

    def subset_lh(self, objects, tables, group_idx, temperature=1.0):
        """float32 [n, F, C]: likelihood of the listed objects' observations under per-component tables
        (`tables`: list of [G_c, F, S] float32; `group_idx`: int [C, n], -1 = no group)."""
        objs = _as(objects, np.int32).reshape(-1)
        tabs = [_c(t, np.float32).reshape(-1, self.n_features, self.n_states) for t in tables]
        offsets = np.concatenate([[0], np.cumsum([t.shape[0] for t in tabs])]).astype(np.int32)
        cat = np.ascontiguousarray(np.concatenate(tabs, axis=0))
        gi = _as(group_idx, np.int32)
        if gi.shape != (len(tabs), objs.size):
            raise ValueError("group_idx must be [n_components, n_objects_in_subset]")
        out = np.empty((objs.size, self.n_features, len(tabs)), dtype=np.float32)
        # (_i hands over a bare address: every array passed must be bound to a name that outlives the call)
        starts = np.ascontiguousarray(offsets[:-1])
        self._check(self._lib.sbe_subset_lh(self._h, self._i(objs), objs.size, len(tabs), self._i(cat), self._i(starts),
                                            int(offsets[-1]), self._i(gi), float(temperature), self._o(out)))
        return out

    # -- round 3: delta / resident forms for the drop-in host layer (what crosses PCIe per MCMC step: object lists and a
    #    few changed rows) ------------------------------------------------------------------------------------------
    def set_uniform_counts(self, unif_counts):
        """DirichletPrior.uniform_concentration_array (prior.py:184-186), float64 [F, S]: resident operand of the tempered
        tables of cluster_posterior_marginals / given_unchanged_lh / jump_lh_resident."""
        u = _c(unif_counts, np.float64)
        if u.shape != (self.n_features, self.n_states):
            raise ValueError(f"unif_counts must be {(self.n_features, self.n_states)}, got {u.shape}")
        self._check(self._lib.sbe_set_uniform_counts(self._h, self._i(u)))

    def counts_delta(self, objects, gid_old, gid_new, src_old, src_new, follow_slot=None, update_probs=False, update_source=False):
        """update_feature_counts (counts.py:55-95) for the listed objects, stateless: gid_* int [C, n] global group index
        per component (-1: none), src_* uint8 [n, F] source component per observation (255: none).  Returns
        (touched, diff): the sorted global indices of the groups any listed object is in (either state) and the float32
        rows [len(touched), F, S] of new_counts - old_counts; every other row of that difference is zero.
        `follow_slot`: that slot's resident counts take the difference on the device (and, `update_probs`, the probability
        rows of the touched groups are rebuilt; `update_source`: its source rows of the listed objects become src_new) -- for
        a slot that holds the counts the difference is added to on the host, so that the rows are not sent back
        (binding.counts_follow_*)."""
        objs = _as(objects, np.int32).reshape(-1)
        n = objs.size
        C, F = self.n_components, self.n_features
        go = _as(gid_old, np.int32).reshape(C, n)
        gn = _as(gid_new, np.int32).reshape(C, n)
        so = _as(src_old, np.uint8)
        sn = so if src_new is src_old else _as(src_new, np.uint8)
        if so.shape != (n, F) or sn.shape != so.shape:
            raise ValueError(f"src_old / src_new must be [{n}, {F}]")
        # the groups any listed object is in, in either state (sorted): one pass in the host helper
        touched = _fast.touched_groups(go, gn, self.n_groups_total)
        self.h2d_bytes += go.nbytes + gn.nbytes
        nt = touched.size
        diff = np.zeros((nt, F, self.n_states), dtype=np.float32)
        if follow_slot is not None:
            self._touch(follow_slot)
        if nt and n:
            if follow_slot is None:
                self._check(self._lib.sbe_counts_delta(self._h, self._i(objs), n, _ptr(go), _ptr(gn), self._i(so), self._i(sn),
                                                       self._i(touched), nt, self._o(diff)))
            else:
                self._check(self._lib.sbe_counts_delta_apply(self._h, int(follow_slot), 1 if update_probs else 0,
                                                             1 if update_source else 0, self._i(objs), n,
                                                             _ptr(go), _ptr(gn), self._i(so), self._i(sn), self._i(touched), nt,
                                                             self._o(diff)))
        return touched, diff

    def set_counts_rows(self, slot, group_idx, rows, update_probs=False):
        """Rows `group_idx` (global group indices) of the slot's resident counts <- float32 rows [n, F, S].
        update_probs=True: the probability rows of those groups are rebuilt in the same launch (untempered, like
        update_probs(slot, c)); the components' tables must exist (update_probs once) -- afterwards they are current again."""
        gi = _as(group_idx, np.int32).reshape(-1)
        r = _c(rows, np.float32)
        if r.shape != (gi.size, self.n_features, self.n_states):
            raise ValueError(f"rows must be [{gi.size}, {self.n_features}, {self.n_states}], got {r.shape}")
        self._touch(slot)
        if gi.size:
            fn = self._lib.sbe_set_counts_rows_probs if update_probs else self._lib.sbe_set_counts_rows
            self._check(fn(self._h, slot, self._i(gi), gi.size, self._i(r)))

    def given_unchanged_lh(self, slot, i_cluster, objects, temperature=1.0, prior_temperature=1.0):
        """component_likelihood_given_unchanged (operators.py:863-928) of the sample bound to `slot` (groups, source,
        counts, concentrations resident): float32 [n, F, C].  Only the object list goes up."""
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((objs.size, self.n_features, self.n_components), dtype=np.float32)
        self._check(self._lib.sbe_given_unchanged_lh(self._h, slot, int(i_cluster), self._i(objs), objs.size, float(temperature),
                                                     float(prior_temperature), self._o(out)))
        return out

    def given_unchanged_gibbs(self, slot, i_cluster, objects, hc_new, hc_old, src_old, z, temperature=1.0, prior_temperature=1.0,
                              from_prior=False, gid_old=None, gid_new=None, follow=False, update_probs=False):
        """ClusterOperator.gibbs_sample_source (operators.py:796-851) on the slot the NEW sample is bound to: returns
        (src_new uint8 [n, F] drawn component, 255 = NA observation; sel_new float32 [n, F] = p[drawn]; sel_back float32
        [n, F] = p_back[old source]).  hc_new / hc_old: bool [n, C] has_components rows of the new / old sample; src_old:
        uint8 [n, F] old source component (255: none); z: uniforms [n, F]."""
        objs = _as(objects, np.int32).reshape(-1)
        n, F, C = objs.size, self.n_features, self.n_components
        hn = _c(np.asarray(hc_new).astype(bool, copy=False), np.uint8)
        ho = _c(np.asarray(hc_old).astype(bool, copy=False), np.uint8)
        so = _as(src_old, np.uint8)
        zz = _c(z, np.float64)
        if zz.size != n * F:
            raise ValueError(f"z must hold {n} x {F} uniforms")
        zz = zz.reshape(n, F)
        if hn.shape != (n, C) or ho.shape != (n, C) or so.shape != (n, F):
            raise ValueError(f"hc_new / hc_old must be [{n}, {C}], src_old and z [{n}, {F}]")
        ids = np.empty((n, F), dtype=np.uint8)
        sel = np.empty((n, F), dtype=np.float32)
        back = np.empty((n, F), dtype=np.float32)
        if gid_old is not None:
            # ... and the count delta of the proposal (update_feature_counts, counts.py:55-95) from the same launch:
            # gid_old / gid_new int32 [C, n] global group ids of the objects in the old / new sample (-1: none); returns
            # additionally (touched int32 [T] ascending, rows float32 [T, F, S] = new counts - old counts of those groups)
            go = _as(gid_old, np.int32).reshape(C, n)
            gn = _as(gid_new, np.int32).reshape(C, n)
            touched = np.empty(self.n_groups_total, dtype=np.int32)
            rows = np.empty((min(self.n_groups_total, 2 * C * max(n, 1)), F, self.n_states), dtype=np.float32)
            nt = ct.c_int32(0)
            # follow=True: when the call touches any group, the slot itself takes the proposal -- counts += delta, the subset's
            # source rows = the drawn ids, update_probs: the touched groups' probability rows -- behind the same launch
            # (binding.counts_follow_*: the next bind has nothing to send)
            if follow:
                self._touch(slot)
            if n and follow:
                self._check(self._lib.sbe_given_unchanged_gibbs_apply(
                    self._h, slot, 1 if update_probs else 0, int(i_cluster), self._i(objs), n, float(temperature), float(prior_temperature),
                    int(bool(from_prior)), self._i(hn), self._i(ho), self._i(so), self._i(zz), self._i(go), self._i(gn), self._o(ids),
                    self._o(sel), self._o(back), _ptr(touched), ct.byref(nt), _ptr(rows)))
                self.d2h_bytes += nt.value * (4 + F * self.n_states * 4)
            elif n:
                self._check(self._lib.sbe_given_unchanged_gibbs_counts(
                    self._h, slot, int(i_cluster), self._i(objs), n, float(temperature), float(prior_temperature), int(bool(from_prior)),
                    self._i(hn), self._i(ho), self._i(so), self._i(zz), self._i(go), self._i(gn), self._o(ids), self._o(sel),
                    self._o(back), _ptr(touched), ct.byref(nt), _ptr(rows)))
                self.d2h_bytes += nt.value * (4 + F * self.n_states * 4)
            return ids, sel, back, touched[:nt.value], rows[:nt.value]
        if n:
            self._check(self._lib.sbe_given_unchanged_gibbs(self._h, slot, int(i_cluster), self._i(objs), n, float(temperature),
                                                            float(prior_temperature), int(bool(from_prior)), self._i(hn), self._i(ho),
                                                            self._i(so), self._i(zz), self._o(ids), self._o(sel), self._o(back)))
        return ids, sel, back

    def cluster_posterior_marginals(self, slot, i_cluster, objects, temperature=1.0, prior_temperature=1.0):
        """cluster_marginals with the candidate table conditional_effect_mean(prior, counts[[i_cluster]], unif, T_prior, T)
        built on the device from the slot's resident counts (operators.py:1046-1052): float64 [2, n]."""
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((2, objs.size), dtype=np.float64)
        self._check(self._lib.sbe_cluster_posterior_marginals(self._h, slot, int(i_cluster), float(temperature),
                                                              float(prior_temperature), self._i(objs), objs.size, self._o(out)))
        return out

    def jump_lh_resident(self, slot, i_source, i_target, objects, temperature=1.0, prior_temperature=1.0):
        """jump_lh with every tempered table built on the device from the slot's resident counts: float64 [2, n]."""
        objs = _as(objects, np.int32).reshape(-1)
        out = np.empty((2, objs.size), dtype=np.float64)
        self._check(self._lib.sbe_jump_lh_resident(self._h, slot, int(i_source), int(i_target), float(temperature),
                                                   float(prior_temperature), self._i(objs), objs.size, self._o(out)))
        return out

    def source_prior(self, slot):
        """float64 [N]: per-object log source prior (SourcePrior.__call__, prior.py:573-611)."""
        out = np.empty(self.n_objects, dtype=np.float64)
        self._check(self._lib.sbe_source_prior(self._h, slot, self._o(out)))
        return out

    def collapsed_and_source_prior(self, slot):
        """(collapsed_loglik_all(slot), source_prior(slot)) in one launch and one synchronisation: Model.__call__ asks
        for the likelihood and then the prior of the same sample (sbayes/model/model.py:47-51)."""
        per_group = np.empty(self.n_groups_total, dtype=np.float64)
        per_object = np.empty(self.n_objects, dtype=np.float64)
        self._check(self._lib.sbe_collapsed_and_source_prior(self._h, slot, self._o(per_group), self._o(per_object)))
        return per_group, per_object

    def observation_lh_exact(self, slot):
        """float64 [N, F]: sum_c w * lh_exact, the LikelihoodLogger row (loggers.py:354-359)."""
        out = np.empty((self.n_objects, self.n_features), dtype=np.float64)
        self._check(self._lib.sbe_observation_lh_exact(self._h, slot, self._o(out)))
        return out

    def step(self, cur_slot, cand_slot, clusters=None, changed_objects=None, source_rows=None, weights=None):
        """One MCMC step in one call: candidate = current + delta, evaluated on the device.
        Returns (group_logliks float64 [G_total], mixture_ll float, changed_groups bool [G_total])."""
        cl = None
        if clusters is not None:
            cl = np.asarray(clusters)
            if cl.shape != (self.n_groups[0], self.n_objects):
                raise ValueError(f"clusters must be {(self.n_groups[0], self.n_objects)}, got {cl.shape}")
            cl = _c(cl.astype(bool, copy=False), np.uint8)
        objs = rows = None
        n_changed = 0
        if changed_objects is not None and len(changed_objects):
            objs = _as(changed_objects, np.int32).reshape(-1)
            rows = np.asarray(source_rows)
            if rows.shape != (objs.size, self.n_features, self.n_components):
                raise ValueError("source_rows must be [len(changed_objects), n_features, n_components]")
            rows = _c(rows.astype(bool, copy=False), np.uint8)
            n_changed = objs.size
        w = None
        if weights is not None:
            w = _c(weights, np.float32)
            if w.shape != (self.n_features, self.n_components):
                raise ValueError("weights must be [n_features, n_components]")
        glh = np.empty(self.n_groups_total, dtype=np.float64)
        mix = ct.c_double(0.0)
        changed = np.zeros(self.n_groups_total, dtype=np.uint8)
        self._touch(cand_slot)
        self._check(self._lib.sbe_step(self._h, cur_slot, cand_slot, self._i(cl) if cl is not None else None,
                                       self._i(objs) if objs is not None else None, n_changed,
                                       self._i(rows) if rows is not None else None, self._i(w) if w is not None else None,
                                       self._o(glh), ct.byref(mix), self._o(changed)))
        return glh, mix.value, changed.astype(bool)

    def step_batch(self, cur_slots, cand_slots, clusters=None, clusters_mask=None, rows_ptr=None, changed_objects=None,
                   source_rows=None, weights=None, weights_mask=None):
        """One MCMC step for each of n chains in ONE call (sbe_step_batch).  Stacked inputs: clusters bool [n, K, N]
        (or None), clusters_mask bool [n] (or None = every chain), rows_ptr int [n+1] CSR over changed_objects /
        source_rows bool [total, F, C] (or None = no source change), weights float32 [n, F, C] + weights_mask.
        Returns (group_logliks float64 [n, G_total], mixture_ll float64 [n], changed_groups bool [n, G_total])."""
        cur = _as(cur_slots, np.int32).reshape(-1)
        cand = _as(cand_slots, np.int32).reshape(-1)
        n = cur.size
        if cand.size != n:
            raise ValueError("cur_slots and cand_slots must have the same length")
        cl = cm = None
        if clusters is not None:
            cl = np.asarray(clusters)
            if cl.shape != (n, self.n_groups[0], self.n_objects):
                raise ValueError(f"clusters must be {(n, self.n_groups[0], self.n_objects)}, got {cl.shape}")
            cl = _c(cl.astype(bool, copy=False), np.uint8)
            if clusters_mask is not None:
                cm = _c(np.asarray(clusters_mask, dtype=bool), np.uint8).reshape(n)
        if rows_ptr is None:
            ptr = np.zeros(n + 1, dtype=np.int32)
            objs = rows = None
        else:
            ptr = _as(rows_ptr, np.int32).reshape(-1)
            if ptr.size != n + 1:
                raise ValueError("rows_ptr must have n_chains + 1 entries")
            total = int(ptr[-1])
            objs = _as(changed_objects, np.int32).reshape(-1) if total else None
            rows = np.asarray(source_rows) if total else None
            if total and (objs.size != total or rows.shape != (total, self.n_features, self.n_components)):
                raise ValueError("changed_objects / source_rows do not match rows_ptr")
            if total:
                rows = _c(rows.astype(bool, copy=False), np.uint8)
        w = wm = None
        if weights is not None:
            w = _c(weights, np.float32)
            if w.shape != (n, self.n_features, self.n_components):
                raise ValueError(f"weights must be {(n, self.n_features, self.n_components)}")
            if weights_mask is not None:
                wm = _c(np.asarray(weights_mask, dtype=bool), np.uint8).reshape(n)
        glh = np.empty((n, self.n_groups_total), dtype=np.float64)
        mix = np.empty(n, dtype=np.float64)
        changed = np.zeros((n, self.n_groups_total), dtype=np.uint8)
        if self._bound or self._mirror:                              # (bind-cache entries of the candidate slots)
            for s in cand.tolist():
                self._touch(s)
        opt = lambda a: self._i(a) if a is not None else None          # noqa: E731
        self._check(self._lib.sbe_step_batch(self._h, n, self._i(cur), self._i(cand), opt(cl), opt(cm), self._i(ptr), opt(objs), opt(rows),
                                             opt(w), opt(wm), self._o(glh), self._o(mix), self._o(changed)))
        return glh, mix, changed.astype(bool)

    def step_delta(self, cur_slot, cand_slot, moved_objects=None, moved_cluster=None, changed_objects=None, source_rows=None,
                   weights=None):
        """sbe_step with the proposal in delta form: the objects that change cluster and their new cluster (-1: none), the
        objects whose source rows change (each once) with their rows.  Same return values as step()."""
        mo = np.ascontiguousarray(moved_objects if moved_objects is not None else [], dtype=np.int32).reshape(-1)
        mc = np.ascontiguousarray(moved_cluster if moved_cluster is not None else [], dtype=np.int32).reshape(-1)
        if mo.size != mc.size:
            raise ValueError("moved_objects and moved_cluster must have the same length")
        objs = rows = None
        n_changed = 0
        if changed_objects is not None and len(changed_objects):
            objs = _as(changed_objects, np.int32).reshape(-1)
            rows = np.asarray(source_rows)
            if rows.shape != (objs.size, self.n_features, self.n_components):
                raise ValueError("source_rows must be [len(changed_objects), n_features, n_components]")
            rows = _c(rows.astype(bool, copy=False), np.uint8)
            n_changed = objs.size
        w = None
        if weights is not None:
            w = _c(weights, np.float32)
            if w.shape != (self.n_features, self.n_components):
                raise ValueError("weights must be [n_features, n_components]")
        glh = np.empty(self.n_groups_total, dtype=np.float64)
        mix = ct.c_double(0.0)
        changed = np.zeros(self.n_groups_total, dtype=np.uint8)
        self._touch(cand_slot)
        opt = lambda a: self._i(a) if a is not None and a.size else None     # noqa: E731
        self._check(self._lib.sbe_step_delta(self._h, cur_slot, cand_slot, opt(mo), opt(mc), mo.size, opt(objs), n_changed, opt(rows),
                                             self._i(w) if w is not None else None, self._o(glh), ct.byref(mix), self._o(changed)))
        return glh, mix.value, changed.astype(bool)

    def step_batch_delta(self, cur_slots, cand_slots, moved_ptr, moved_objects, moved_cluster, rows_ptr=None,
                         changed_objects=None, source_rows=None, weights=None, weights_mask=None):
        """sbe_step_batch with the proposals in delta form: moved_ptr int [n+1] CSR over moved_objects / moved_cluster
        (new cluster index, -1 = none); rows_ptr / changed_objects / source_rows as in step_batch (each object once per
        chain); weights float32 [n, F, C] + weights_mask.  Same return values as step_batch."""
        cur = _as(cur_slots, np.int32).reshape(-1)
        cand = _as(cand_slots, np.int32).reshape(-1)
        n = cur.size
        mp = _as(moved_ptr, np.int32).reshape(-1)
        if cand.size != n or mp.size != n + 1:
            raise ValueError("cur_slots / cand_slots / moved_ptr do not match")
        mo = _as(moved_objects, np.int32).reshape(-1)
        mc = _as(moved_cluster, np.int32).reshape(-1)
        if mo.size != int(mp[-1]) or mc.size != mo.size:
            raise ValueError("moved_objects / moved_cluster do not match moved_ptr")
        if rows_ptr is None:
            ptr = np.zeros(n + 1, dtype=np.int32)
            objs = rows = None
        else:
            ptr = _as(rows_ptr, np.int32).reshape(-1)
            if ptr.size != n + 1:
                raise ValueError("rows_ptr must have n_chains + 1 entries")
            total = int(ptr[-1])
            objs = _as(changed_objects, np.int32).reshape(-1) if total else None
            rows = np.asarray(source_rows) if total else None
            if total and (objs.size != total or rows.shape != (total, self.n_features, self.n_components)):
                raise ValueError("changed_objects / source_rows do not match rows_ptr")
            if total:
                rows = _c(rows.astype(bool, copy=False), np.uint8)
        w = wm = None
        if weights is not None:
            w = _c(weights, np.float32)
            if w.shape != (n, self.n_features, self.n_components):
                raise ValueError(f"weights must be {(n, self.n_features, self.n_components)}")
            if weights_mask is not None:
                wm = _c(np.asarray(weights_mask, dtype=bool), np.uint8).reshape(n)
        glh = np.empty((n, self.n_groups_total), dtype=np.float64)
        mix = np.empty(n, dtype=np.float64)
        changed = np.zeros((n, self.n_groups_total), dtype=np.uint8)
        if self._bound or self._mirror:
            for s in cand.tolist():
                self._touch(s)
        opt = lambda a: self._i(a) if a is not None else None        # noqa: E731
        self._check(self._lib.sbe_step_batch_delta(self._h, n, self._i(cur), self._i(cand), self._i(mp), opt(mo if mo.size else None),
                                                   opt(mc if mc.size else None), self._i(ptr), opt(objs), opt(rows), opt(w), opt(wm),
                                                   self._o(glh), self._o(mix), self._o(changed)))
        return glh, mix, changed.astype(bool)

    def gibbs_propose_supported(self):
        """True if gibbs_propose applies to this engine's shape (its tables fit the fused table kernel of the chain)."""
        if getattr(self, "_gibbs_propose_ok", None) is None:
            self._gibbs_propose_ok = self._lib.sbe_gibbs_propose_supported(self._h) == 1
        return self._gibbs_propose_ok

    def gibbs_propose(self, cur_slot, cand_slot, objects, z, temperature=1.0, prior_temperature=1.0, from_prior=False, follow=False):
        """GibbsSampleSource._propose (operators.py:495-552) in one call (sbe_gibbs_propose): the listed objects' source is
        redrawn into `cand_slot` (= `cur_slot`'s state otherwise) with the uniforms z [n, F]; counts and tables follow on
        the device.  Returns (ids uint8 [n, F]: drawn component, 255 = NA observation; sel float32 [n, F] = p[drawn];
        sel_back float32 [n, F] = p_back[old source]; touched int32 [T]: the groups the objects are in, ascending;
        rows float32 [T, F, S]: candidate counts - current counts of those groups).  follow=True: when the proposal touches
        any group, `cur_slot` itself takes it on the device (counts, the touched groups' tables, the drawn source rows)."""
        objs = _as(objects, np.int32).reshape(-1)
        n, F, S = objs.size, self.n_features, self.n_states
        zz = _c(z, np.float64)
        if zz.size != n * F:
            raise ValueError(f"z must hold {n} x {F} uniforms")
        ids = np.empty((n, F), dtype=np.uint8)
        sel = np.empty((n, F), dtype=np.float32)
        back = np.empty((n, F), dtype=np.float32)
        touched = np.empty(self.n_groups_total, dtype=np.int32)
        rows = np.empty((min(self.n_groups_total, n * self.n_components), F, S), dtype=np.float32)
        nt = ct.c_int32(0)
        self._touch(cand_slot)
        if follow:
            self._touch(cur_slot)
        fn = self._lib.sbe_gibbs_propose_apply if follow else self._lib.sbe_gibbs_propose
        self._check(fn(self._h, cur_slot, cand_slot, self._i(objs), n, float(temperature),
                       float(prior_temperature), int(bool(from_prior)), self._i(zz), self._o(ids),
                       self._o(sel), self._o(back), _ptr(touched), ct.byref(nt), _ptr(rows)))
        self.d2h_bytes += nt.value * (4 + F * S * 4)
        return ids, sel, back, touched[:nt.value], rows[:nt.value]

    def gibbs_step(self, cur_slot, cand_slot, objects, z=None, temperature=1.0, prior_temperature=1.0, from_prior=False):
        """One Gibbs-source MCMC step in one call (sbe_gibbs_step): the listed objects' source is redrawn on the
        device into `cand_slot`, counts / tables / likelihoods follow.  z: uniforms [n, F] or None (engine's Philox
        stream).  Returns (log_q, log_q_back, group_logliks [G_total], mixture_ll, changed_groups bool [G_total])."""
        objs = _as(objects, np.int32).reshape(-1)
        zz = None
        if z is not None:
            zz = _c(z, np.float64).reshape(objs.size, -1)
            if zz.shape != (objs.size, self.n_features):
                raise ValueError(f"z must be [{objs.size}, {self.n_features}]")
        glh = np.empty(self.n_groups_total, dtype=np.float64)
        lq, lqb, mix = ct.c_double(0.0), ct.c_double(0.0), ct.c_double(0.0)
        changed = np.zeros(self.n_groups_total, dtype=np.uint8)
        self._touch(cand_slot)
        self._check(self._lib.sbe_gibbs_step(self._h, cur_slot, cand_slot, self._i(objs), objs.size, float(temperature),
                                             float(prior_temperature), int(bool(from_prior)),
                                             self._i(zz) if zz is not None else None, ct.byref(lq), ct.byref(lqb),
                                             self._o(glh), ct.byref(mix), self._o(changed)))
        return lq.value, lqb.value, glh, mix.value, changed.astype(bool)

    def test_fast_log(self, x):
        """(fast, library) fp64 logs of x computed on the device (self-test of the table-build log)."""
        x = _as(x, np.float64).reshape(-1)
        a, b = np.empty_like(x), np.empty_like(x)
        self._check(self._lib.sbe_test_fast_log(self._h, self._i(x), x.size, self._o(a), self._o(b)))
        return a, b

    def test_lgamma(self, x):
        """The engine's lgamma (Dirichlet-categorical terms) of x, computed on the device (self-test)."""
        x = _as(x, np.float64).reshape(-1)
        out = np.empty_like(x)
        self._check(self._lib.sbe_test_lgamma(self._h, self._i(x), x.size, self._o(out)))
        return out

    def test_tab_log(self, x):
        """Table-driven fp64 log of k_mixture_tuple64's table build, computed on the device (self-test)."""
        x = _as(x, np.float64).reshape(-1)
        out = np.empty_like(x)
        self._check(self._lib.sbe_test_tab_log(self._h, self._i(x), x.size, self._o(out)))
        return out

    def copy_slot(self, dst, src):
        self._touch(dst)
        self._check(self._lib.sbe_copy_slot(self._h, dst, src))

    # -- measurement ---------------------------------------------------------------------------------
    def timer_start(self):
        self._check(self._lib.sbe_timer_start(self._h))

    def timer_stop(self) -> float:
        ms = ct.c_float(0.0)
        self._check(self._lib.sbe_timer_stop(self._h, ct.byref(ms)))
        return ms.value

    def timer_mark(self):
        """Second event of the span opened by timer_start, recorded behind everything enqueued so far; no host wait."""
        self._check(self._lib.sbe_timer_mark(self._h))

    def timer_elapsed(self) -> float:
        """Wait for timer_mark's event; -> span in ms since timer_start's."""
        ms = ct.c_float(0.0)
        self._check(self._lib.sbe_timer_elapsed(self._h, ct.byref(ms)))
        return ms.value

    def kernel_timing_start(self):
        """Record one HIP event pair (engine stream) around the fused kernel of every mixture launch from now on;
        pairs recorded earlier are forgotten (kernel_timing_resume keeps them)."""
        self._check(self._lib.sbe_kernel_timing(self._h, 1, None, None))

    def kernel_timing_resume(self):
        """Bracket launches again after kernel_timing_pause, keeping the pairs recorded so far."""
        self._check(self._lib.sbe_kernel_timing(self._h, 3, None, None))

    def kernel_timing_pause(self):
        """Stop bracketing launches; the recorded pairs are kept (kernel_timing_resume continues)."""
        self._check(self._lib.sbe_kernel_timing(self._h, 2, None, None))

    def last_mixture_kernel(self) -> str:
        """Kernel form the most recent fused mixture launch ran, e.g. 'k_mixture_tuple64<packed stream, ...>'."""
        return self._lib.sbe_last_mixture_kernel(self._h).decode()

    def kernel_timing_stop(self):
        """-> (recorded launches, average duration of the fused kernel in ms)."""
        n, avg = ct.c_int(0), ct.c_float(0.0)
        self._check(self._lib.sbe_kernel_timing(self._h, 0, ct.byref(n), ct.byref(avg)))
        return n.value, avg.value

    def profile_mixture(self, first_slot, n, iters):
        total, avg = ct.c_float(0.0), ct.c_float(0.0)
        self._check(self._lib.sbe_profile_mixture(self._h, first_slot, n, iters, ct.byref(total), ct.byref(avg)))
        return total.value, avg.value

    # -- convenience: load a whole workload state into a slot ---------------------------------------
    def load_state(self, slot, groups, weights, source=None, counts=None, probs=None, recount=True):
        for c, g in enumerate(groups):
            self.set_groups(slot, c, g)
        if source is not None:
            self.set_source(slot, source)
        if counts is not None:
            for c, cnt in enumerate(counts):
                self.set_counts(slot, c, cnt)
        elif source is not None and recount:
            self.recount(slot)
        if probs is not None:
            for c, p in enumerate(probs):
                self.set_probs(slot, c, p)
        self.set_weights(slot, weights)
