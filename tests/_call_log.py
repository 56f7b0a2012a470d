"""Engine-level call logs of the REAL reference sampler -- TEST INFRASTRUCTURE.

Recording (build container only, tests/golden/make_golden.py:call_log_fixtures): the reference's sampler runs on the
drop-in host layer under patch.install(operators=True) with the device replaced by `RecordingEngine`, an oracle-backed
double (tests/_fake_engine.py) that writes down every Engine-level call the host layer makes: method, arguments
(arrays stored once, content-addressed) and the expected result (small results in full, large ones as SHA-1).  What
is recorded is DATA: call names, argument arrays, result arrays / digests -- no reference code.

Replay (tests/test_gpu_call_log.py on the GPU box; tests/test_call_log_cpu.py against the double): the same call
sequence against an engine, every result checked.  The sequence carries the reference's true aliasing and
caching pattern through conditionals._bind_slot (which slot state is re-sent when, which tables are left stale), so
the device's slot state machine is exercised exactly as the unchanged sampler would drive it.
"""
from __future__ import annotations

import hashlib
import json

import numpy as np

from tests._fake_engine import FakeEngine

SENTINEL = -7.25                      # fill of component_lh's `out` when its expected result is computed
FULL_RESULT_BYTES = 1 << 15           # results up to this size are stored in full, larger ones as a digest


def _digest(a: np.ndarray) -> str:
    a = np.ascontiguousarray(a)
    return hashlib.sha1(str((a.dtype.str, a.shape)).encode() + a.tobytes()).hexdigest()


# Uniform draws that reach an engine call as an ARGUMENT (the `z` of sample_source: np.random.random([n, F, 1]) drawn by
# operators.gibbs_sample_source exactly where the reference's sample_categorical draws, up to N * F doubles per call) are not
# stored as arrays -- incompressible, megabytes per call at the headline shape -- but as the Mersenne-Twister state they were
# drawn from: `record_uniform_draws()` wraps np.random.random while a log is being recorded; on replay the array is
# regenerated from that state (np.random.RandomState.random_sample is the same algorithm: the bits are identical).
_DRAWS = {}                            # sha1 of the drawn bytes -> (MT19937 key vector uint32[624], pos)


class record_uniform_draws:
    def __enter__(self):
        self._orig = np.random.random
        orig = self._orig

        def random(size=None):
            st = np.random.get_state()
            out = orig(size)
            if isinstance(out, np.ndarray) and out.nbytes >= 4096 and st[0] == "MT19937":
                _DRAWS[hashlib.sha1(out.tobytes()).hexdigest()] = (np.array(st[1], dtype=np.uint32), int(st[2]))
            return out
        np.random.random = random
        return self

    def __exit__(self, *exc):
        np.random.random = self._orig
        _DRAWS.clear()


def _regenerate_draw(keys, pos, shape):
    rs = np.random.RandomState()
    rs.set_state(("MT19937", np.asarray(keys, dtype=np.uint32), int(pos), 0, 0.0))
    return rs.random_sample(int(np.prod(shape))).reshape(shape)


class _Store:
    def __init__(self):
        self.arrays, self.index = [], {}

    def put(self, a) -> int:
        a = np.ascontiguousarray(a)
        key = _digest(a)
        if key not in self.index:
            self.index[key] = len(self.arrays)
            self.arrays.append(a.copy())
        return self.index[key]


# how each result is compared on replay: exact = bit for bit; otherwise (rtol, atol)
COMPARE = {
    "component_lh": "exact", "normalize_tables": "exact", "effect_counts": "exact", "normalize_weights": "exact",
    "observation_lh_exact": "exact", "na_values": "exact",
    "dirichlet_logpdf": (2e-6, 1e-6),        # float32 per feature in the reference
    "cluster_marginals": (1e-9, 1e-9),       # log-space sums on the device vs log of the linear-space product
    "jump_lh": (1e-9, 1e-9),                 # fp64 sums of logs of float32 values (table-driven log on the device)
    "source_lh_by_feature": (3e-5, 1e-5),    # float32 logs summed in float32 over the objects (N * 2^-24 / 2 at N = 1000)
    "source_posterior": "exact_at_t1",       # bit-exact at temperature 1, float32 powf tolerance when tempered
    "subset_lh": "exact_at_t1",
    "counts_delta": "exact", "given_unchanged_lh": "exact_at_t1", "get_counts": "exact",
    "collapsed_loglik": (2e-6, 1e-6),        # float32 per feature in the reference
    "collapsed_loglik_all": (2e-6, 1e-6),
    "source_prior": (2e-6, 1e-6),            # float32 logs, fp64 accumulation on the device
    "collapsed_and_source_prior": (2e-6, 1e-6),
    "get_counts_all": "exact",
    "cluster_posterior_marginals": (1e-9, 1e-9), "jump_lh_resident": (1e-9, 1e-9),
    # the Gibbs source proposal on slot state (patch.install(gibbs_source=True) -> operators.gibbs_sample_source)
    "sample_source": (2e-6, 1e-6),           # (log_q, selected probabilities): float32 posterior values; the DRAW itself is
    "source_logprob": (2e-6, 1e-6),          #   pinned exactly by get_source_rows below
    "update_counts": "exact", "get_source_rows": "exact",
    # ClusterOperator.gibbs_sample_source in one call: (drawn ids, p[drawn], p_back[old source]) -- bit-exact at
    # temperature 1 (ids always: a draw that differs changes the chain), float32 powf tolerance when tempered
    "given_unchanged_gibbs": "exact_at_t1",
    # GibbsSampleSource._propose in one call: (drawn ids, p[drawn], p_back[old source], touched groups, count rows)
    "gibbs_propose": "exact_at_t1",
}
SETTERS = {"set_groups", "set_concentration", "set_counts", "set_source", "set_weights", "update_probs", "set_counts_rows",
           "set_source_rows", "set_uniform_counts", "recount", "copy_slot", "set_slot_delta"}


_TEMPERATURE_ARG = {"source_posterior": 2, "subset_lh": 3, "given_unchanged_lh": 3,      # positional index of `temperature`
                    "given_unchanged_gibbs": 7, "gibbs_propose": 4}


def _temperature_of(name, args, kwargs):
    t = kwargs.get("temperature", None)
    i = _TEMPERATURE_ARG.get(name)
    if t is None and i is not None and len(args) > i:
        t = args[i]
    return 1.0 if t is None else float(t)


class RecordingEngine(FakeEngine):
    """FakeEngine that logs every call (see module docstring)."""

    def __init__(self, features, n_groups=None, n_slots=4, device=0):
        super().__init__(features, n_groups, n_slots, device)
        self.store = _Store()
        self.log = []

    # -- helpers --------------------------------------------------------------------------------------------
    def _arg(self, v):
        if isinstance(v, np.ndarray) and v.dtype == np.float64 and v.nbytes >= 4096 and _DRAWS:
            drawn = _DRAWS.get(hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest())
            if drawn is not None:                          # a recorded uniform draw: its generator state instead of its values
                assert np.array_equal(_regenerate_draw(drawn[0], drawn[1], v.shape), v)
                return {"mt": self.store.put(drawn[0]), "pos": drawn[1], "shape": list(v.shape)}
        if isinstance(v, np.ndarray) or isinstance(v, (list, tuple)) and len(v) and isinstance(v[0], np.ndarray):
            if isinstance(v, (list, tuple)):
                return {"list": [self.store.put(np.asarray(x)) for x in v]}
            return {"arr": self.store.put(v)}
        if v is None or isinstance(v, (bool, int, float, str)):
            return {"val": v}
        if isinstance(v, (np.integer, np.floating, np.bool_)):
            return {"val": v.item()}
        return {"arr": self.store.put(np.asarray(v))}

    def _result(self, name, r, exact=None):
        if r is None:
            return None
        if exact is None:
            exact = COMPARE[name] == "exact"
        if isinstance(r, tuple):
            return {"tuple": [self._result(name, x, exact) for x in r]}
        r = np.asarray(r)
        if r.nbytes <= FULL_RESULT_BYTES:
            return {"arr": self.store.put(r)}
        if not exact:                      # a large result compared at a tolerance: every k-th element (<= 4096 of them)
            stride = -(-r.size // 4096)
            return {"sub": self.store.put(np.ascontiguousarray(r.ravel()[::stride])), "stride": stride, "shape": list(r.shape),
                    "dtype": r.dtype.str}
        return {"sha": _digest(r), "shape": list(r.shape), "dtype": r.dtype.str}

    def mark_step(self, i_step, operator=None):
        """Step boundary of the sampler's MH loop (MCMCChain.step): everything logged since the previous marker is
        what the unchanged sampler asked of the engine for this one MCMC step."""
        self.log.append({"m": "__step__", "i": int(i_step), "op": operator})

    def _rec(self, name, args, kwargs, result):
        exact = None
        if COMPARE.get(name) == "exact_at_t1":          # bit-exact (hence a digest for a large result) at temperature 1
            exact = _temperature_of(name, args, kwargs) == 1.0
        self.log.append({"m": name, "a": [self._arg(a) for a in args], "k": {k: self._arg(v) for k, v in kwargs.items()},
                         "r": self._result(name, result, exact)})

    # -- logged surface (everything the host layer calls on an engine) -----------------------------------------
    def component_lh(self, probs, groups, changed_groups, out, na_value=0.0):
        res = super().component_lh(probs, groups, changed_groups, out, na_value)
        probe = np.full(out.shape, SENTINEL)              # expected result on a sentinel-filled buffer: rows the call
        FakeEngine.component_lh(self, probs, groups, changed_groups, probe, na_value)     # must not touch stay SENTINEL
        self.calls.pop()
        self._rec("component_lh", (np.asarray(probs), np.asarray(groups, dtype=bool),
                                   np.asarray(changed_groups, dtype=np.int64)), {"na_value": float(na_value)}, probe)
        return res


def _wrap(name):
    def method(self, *args, **kwargs):
        res = getattr(FakeEngine, name)(self, *args, **kwargs)
        self._rec(name, tuple(np.asarray(a) if isinstance(a, np.ndarray) else a for a in args), kwargs,
                  None if name in SETTERS else res)
        return res
    method.__name__ = name
    return method


for _name in ("normalize_tables", "dirichlet_logpdf", "effect_counts", "set_groups", "set_concentration", "set_counts",
              "set_source", "set_weights", "update_probs", "cluster_marginals", "source_posterior", "subset_lh",
              "normalize_weights", "observation_lh_exact", "jump_lh", "source_lh_by_feature", "set_counts_rows",
              "set_source_rows", "set_uniform_counts", "counts_delta", "collapsed_loglik", "collapsed_loglik_all", "source_prior",
              "given_unchanged_lh", "cluster_posterior_marginals", "jump_lh_resident", "recount", "get_counts",
              "copy_slot", "sample_source", "source_logprob", "update_counts", "get_source_rows", "given_unchanged_gibbs",
              "gibbs_propose", "collapsed_and_source_prior", "set_slot_delta", "get_counts_all"):
    setattr(RecordingEngine, _name, _wrap(_name))


def save(path, eng: RecordingEngine, meta: dict):
    arrays = {f"arr_{i}": a for i, a in enumerate(eng.store.arrays)}
    np.savez_compressed(path, calls=np.array(json.dumps(eng.log)), meta=np.array(json.dumps(meta)),
                        n_arrays=np.int64(len(eng.store.arrays)), **arrays)


# ---------------------------------------------------------------------------------------------------------------
# replay
# ---------------------------------------------------------------------------------------------------------------
def _load_arg(z, spec):
    if "mt" in spec:
        return _regenerate_draw(z[f"arr_{spec['mt']}"], spec["pos"], spec["shape"])
    if "arr" in spec:
        return z[f"arr_{spec['arr']}"]
    if "list" in spec:
        return [z[f"arr_{i}"] for i in spec["list"]]
    return spec["val"]


def _check(name, spec, got, z, where, temperature):
    if spec is None:
        return
    if "tuple" in spec:
        assert isinstance(got, tuple) and len(got) == len(spec["tuple"]), where
        for s, g in zip(spec["tuple"], got):
            _check(name, s, g, z, where, temperature)
        return
    got = np.asarray(got)
    mode = COMPARE[name]
    if mode == "exact_at_t1":
        mode = "exact" if temperature == 1.0 else (2e-6, 1e-7)
    if "sha" in spec:
        assert list(got.shape) == spec["shape"] and got.dtype.str == spec["dtype"], where
        assert _digest(got) == spec["sha"], f"{where}: result digest differs"
        return
    if "sub" in spec:
        assert list(got.shape) == spec["shape"] and got.dtype.str == spec["dtype"], where
        assert mode != "exact", where
        np.testing.assert_allclose(got.ravel()[::spec["stride"]], z[f"arr_{spec['sub']}"], rtol=mode[0], atol=mode[1], err_msg=where)
        return
    want = z[f"arr_{spec['arr']}"]
    if got.ndim == 0 and want.shape == (1,):          # (a scalar result was stored as a one-element array)
        got = got.reshape(1)
    assert got.shape == want.shape, (where, got.shape, want.shape)
    if mode == "exact":
        assert got.dtype == want.dtype and np.array_equal(got, want), f"{where}: result differs"
    else:
        np.testing.assert_allclose(got, want, rtol=mode[0], atol=mode[1], err_msg=where)


def replay(path, make_engine):
    """Run the recorded call sequence against `make_engine(n_groups)`; returns per-method call counts."""
    z = np.load(path, allow_pickle=False)
    calls = json.loads(str(z["calls"]))
    meta = json.loads(str(z["meta"]))
    eng = make_engine(meta["n_groups"])
    counts = {}
    try:
        for i, c in enumerate(calls):
            name = c["m"]
            if name == "__step__":
                counts["__step__"] = counts.get("__step__", 0) + 1
                continue
            args = [_load_arg(z, a) for a in c["a"]]
            kwargs = {k: _load_arg(z, v) for k, v in c["k"].items()}
            where = f"call {i} ({name})"
            counts[name] = counts.get(name, 0) + 1
            if name == "component_lh":
                out = np.full((eng.n_objects, eng.n_features), SENTINEL)
                eng.component_lh(args[0], args[1], args[2], out, kwargs.get("na_value", 0.0))
                got = out
            else:
                got = getattr(eng, name)(*args, **kwargs)
            _check(name, c["r"], got, z, where, _temperature_of(name, args, kwargs))
    finally:
        eng.close()
    return counts, meta


def replay_timed(path, make_engine, repeats=1, by_method=None, drain=False):
    """Timing-only replay (bench.py's sampler_replay block): the recorded call sequence against `make_engine(n_groups)`
    with no result checks (tests/test_gpu_call_log.py does those) and the argument arrays loaded beforehand.  Calls
    before the first step marker (model set-up, initialiser) are excluded.  Returns a dict: steps, calls, seconds of the
    MCMC-step part, and -- when the engine counts them (Engine.traffic) -- the bytes that crossed the ABI."""
    import time
    z = np.load(path, allow_pickle=False)
    calls = json.loads(str(z["calls"]))
    meta = json.loads(str(z["meta"]))
    arrays = {k: z[k] for k in z.files if k.startswith("arr_")}
    prepared = []
    for c in calls:
        if c["m"] == "__step__":
            prepared.append(None)
            continue
        args = [_load_arg(arrays, a) for a in c["a"]]
        kwargs = {k: _load_arg(arrays, v) for k, v in c["k"].items()}
        prepared.append((c["m"], args, kwargs))
    first_marker = next((i for i, c in enumerate(prepared) if c is None), len(prepared))
    n_steps = sum(1 for c in prepared if c is None)
    eng = make_engine(meta["n_groups"])
    out_buf = np.empty((eng.n_objects, eng.n_features))
    best = None
    try:
        for _ in range(max(1, repeats)):
            t_steps, n_calls = 0.0, 0
            for i, c in enumerate(prepared):
                if i == first_marker:
                    if hasattr(eng, "sync"):
                        eng.sync()
                    if hasattr(eng, "traffic"):
                        eng.traffic(reset=True)
                    t0 = time.perf_counter()
                if c is None:
                    continue
                name, args, kwargs = c
                if by_method is not None and i > first_marker:
                    t_call = time.perf_counter()
                if name == "component_lh":
                    eng.component_lh(args[0], args[1], args[2], out_buf, kwargs.get("na_value", 0.0))
                else:
                    getattr(eng, name)(*args, **kwargs)
                if by_method is not None and i > first_marker:
                    acc = by_method.setdefault(name, [0, 0.0, 0.0])
                    acc[0] += 1
                    t_ret = time.perf_counter()
                    acc[1] += t_ret - t_call
                    if drain:                       # what the call left queued on the stream, charged to the call itself
                        eng.sync()
                        acc[2] += time.perf_counter() - t_ret
                n_calls += i > first_marker
            if hasattr(eng, "sync"):
                eng.sync()
            t_steps = time.perf_counter() - t0 if n_steps else 0.0
            tr = eng.traffic() if hasattr(eng, "traffic") else None
            if best is None or t_steps < best["seconds"]:
                best = {"steps": n_steps, "calls": n_calls, "seconds": t_steps, "traffic": tr}
    finally:
        eng.close()
    best["meta"] = meta
    return best
