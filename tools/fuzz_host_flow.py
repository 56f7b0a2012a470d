#!/usr/bin/env python3
"""Open-ended fuzz of the host layer's native per-step flow ON THE DEVICE (GPU box): random shapes and layouts, a random accept /
reject sequence of proposals, through update_feature_counts -> Likelihood.__call__ -> binds (with the source lineage), three ways --
the native flow on the real engine, the Python forms on the real engine (same engine calls: same bits), the native flow on the
oracle-backed double (collapsed log-likelihood at 2e-6 relative: float32 sums, H1) -- and after every case the DEVICE slot's group
ids / weights / source rows read back must be those of the sample bound last.  tests/test_gpu_native_host_flow.py is one fixed case
of this; the oracle is used here as the checker only.

  python tools/fuzz_host_flow.py --seconds 600 --seed 1"""
from __future__ import annotations

import argparse
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import pytest  # noqa: E402  (MonkeyPatch only)

from sbayes_amd import _fast, binding, conditionals, counts as my_counts, likelihood, model as sbm, registry  # noqa: E402
from sbayes_amd.engine import Engine  # noqa: E402
from sbayes_amd.synthetic import make_workload  # noqa: E402
from tests._fake_engine import FakeEngine, make_get_engine  # noqa: E402
from tests.test_native_host_flow_cpu import _move, _python_forms  # noqa: E402


def real_engine(features, n_groups):
    return Engine(features, n_groups if n_groups is not None else [1], n_slots=4)


def drive(mp, wl, cls, python_forms, seed, n_moves):
    engines = {}
    get_engine = make_get_engine(engines, cls)
    for mod in (registry, likelihood, conditionals, my_counts, binding):
        mp.setattr(mod, "get_engine", get_engine, raising=True)
    names = ["clusters"] + [f"conf{i}" for i in range(1, wl.n_components)]
    model, sample = sbm.build(wl.features, wl.states_per_feature, names, list(wl.groups), list(wl.concentration), wl.weights, wl.source)
    feats = model.data.features.values
    my_counts.recalculate_feature_counts(feats, sample)
    if python_forms:
        _python_forms(mp)
    rng = np.random.default_rng(seed)
    trace = [float(model.likelihood(sample))]
    eng = next(iter(engines.values()))
    for it in range(n_moves):
        new, objs = _move(rng, sample, wl)
        subset = objs if rng.random() < 0.5 else np.isin(np.arange(wl.shape[0]), objs)
        my_counts.update_feature_counts(sample, new, feats, subset)
        trace.append(float(model.likelihood(new)))
        r = rng.random()
        if r < 0.25:                                       # ClusterJump's pattern: new -> old -> new with the source
            for s in (sample, new):
                binding._bind_slot(eng, model, s, 0, with_source=True)
        elif r < 0.4:
            with new.weights.edit() as w:                  # a weights move on top (everything re-normalised)
                w[...] = rng.dirichlet(np.ones(w.shape[1]), size=w.shape[0]).astype(np.float32)
            trace.append(float(model.likelihood(new)))
        if rng.random() < 0.4:
            sample = new
    binding._bind_slot(eng, model, sample, 0, with_source=True)
    return trace, sample, eng


def one_case(rng, stats):
    n, f, s = int(rng.integers(8, 400)), int(rng.integers(2, 80)), int(rng.integers(2, 9))
    k = int(rng.integers(1, 6))
    extra = tuple(int(rng.integers(1, 6)) for _ in range(int(rng.integers(0, 3))))
    wl = make_workload("fuzz", data_seed=int(rng.integers(0, 1 << 30)), state_seed=int(rng.integers(0, 1 << 30)),
                       shape=(n, f, s, k, extra, bool(rng.integers(0, 2))))
    seed, n_moves = int(rng.integers(0, 1 << 30)), int(rng.integers(5, 60))
    runs = {}
    for key, cls, py in (("device native", real_engine, False), ("device python", real_engine, True), ("double native", None, False)):
        mp = pytest.MonkeyPatch()
        try:
            trace, bound, eng = drive(mp, wl, cls, py, seed, n_moves)
            if cls is real_engine:
                ids = np.stack([eng.get_group_ids(0, c) for c in range(eng.n_components)])
                want = np.stack([np.where(g.any(axis=0), g.argmax(axis=0), -1) for g in
                                 [bound.clusters.value, *[c.group_assignment for c in bound.confounders.values()]]])
                assert np.array_equal(ids, want), (key, "group ids")
                assert np.array_equal(eng.get_weights(0), np.asarray(bound.weights.value, dtype=np.float32)), (key, "weights")
                assert np.array_equal(eng.get_source_rows(0, np.arange(n, dtype=np.int32)).astype(bool), bound.source.value), (key, "source")
                eng.close()
            runs[key] = trace
        finally:
            mp.undo()
    a, b, c = runs["device native"], runs["device python"], runs["double native"]
    assert a == b, "native and Python host flows differ on the device"
    np.testing.assert_allclose(a, c, rtol=2e-6, atol=1e-9)
    stats["cases"] += 1
    stats["likelihoods"] += len(a)
    stats["moves"] += n_moves


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    assert _fast.HAVE_EXTENSION, "sbayes_amd._sbe_pyhost is not built"
    rng = np.random.default_rng(args.seed)
    stats = {"cases": 0, "likelihoods": 0, "moves": 0}
    t0 = last = time.time()
    while time.time() - t0 < args.seconds:
        one_case(rng, stats)
        if time.time() - last > 30:
            last = time.time()
            print(f"[fuzz host flow] {time.time() - t0:5.0f} s  {stats}", flush=True)
    print(f"[fuzz host flow] done, no mismatch: {stats}", flush=True)


if __name__ == "__main__":
    main()
