"""The native per-step control flow of the host layer (csrc/sbe_pyhost.c: likelihood_call, store_per_object, update_counts, the
bind with its source lineage) on the REAL engine: the same seeded accept / reject sequence of proposals as
tests/test_native_host_flow_cpu.py, three ways --
  native host flow on the device      |  the Python forms on the device  |  native host flow on the oracle-backed double
The two device routes issue the same engine calls and must return the same bits; against the oracle double the collapsed
log-likelihood (likelihood.py:65-101 over util.py:1373-1394, float32-summed like the reference: H1) agrees at 2e-6 relative, and
the state the DEVICE slot holds after a final bind of the chain's current sample (group ids, weights, source rows read back) is
that sample's."""
import numpy as np
import pytest

from sbayes_amd import _fast, binding, counts as my_counts
from sbayes_amd.engine import Engine
from tests._fake_engine import FakeEngine
from tests.test_native_host_flow_cpu import _move, _problem, _python_forms

pytestmark = pytest.mark.gpu


def _real_engine(features, n_groups):
    return Engine(features, n_groups if n_groups is not None else [1], n_slots=4)


def _drive(mp, engine_cls, python_forms):
    engines = {}
    wl, model, sample = _problem(mp, engines, cls=None if engine_cls is FakeEngine else _real_engine)
    if python_forms:
        _python_forms(mp)
    feats = model.data.features.values
    rng = np.random.default_rng(17)
    trace = [float(model.likelihood(sample))]
    bound = sample
    for it in range(60):
        new, objs = _move(rng, sample, wl)
        subset = objs if it % 2 else np.isin(np.arange(wl.shape[0]), objs)
        my_counts.update_feature_counts(sample, new, feats, subset)
        trace.append(float(model.likelihood(new)))
        bound = new
        if it % 5 == 4:
            trace.append(float(model.likelihood(sample)))
            bound = sample
        if rng.random() < 0.4:
            sample = new
    eng = next(iter(engines.values()))
    # (a likelihood of an already evaluated sample is a cached answer and binds nothing: bind the chain's current sample explicitly --
    # through the lineage, the slot holds the last proposal)
    binding._bind_slot(eng, model, sample, 0, with_source=True)
    return trace, sample, eng, wl


@pytest.mark.skipif(not _fast.HAVE_EXTENSION, reason="sbayes_amd._sbe_pyhost is not built")
def test_native_host_flow_on_the_device(monkeypatch):
    runs = {}
    for key, cls, py in (("device native", Engine, False), ("device python", Engine, True), ("double native", FakeEngine, False)):
        with monkeypatch.context() as mp:
            trace, bound, eng, wl = _drive(mp, cls, py)
            extra = None
            if cls is Engine:
                ids = np.stack([eng.get_group_ids(0, c) for c in range(eng.n_components)])
                extra = (ids, eng.get_weights(0), eng.get_source_rows(0, np.arange(wl.shape[0], dtype=np.int32)).astype(bool))
                # the slot holds the last sample the flow bound: group ids, weights, source
                want_ids = np.stack([np.where(g.any(axis=0), g.argmax(axis=0), -1) for g in
                                     [bound.clusters.value, *[c.group_assignment for c in bound.confounders.values()]]])
                assert np.array_equal(ids, want_ids)
                assert np.array_equal(extra[1], np.asarray(bound.weights.value, dtype=np.float32))
                assert np.array_equal(extra[2], bound.source.value)
                eng.close()
            runs[key] = (trace, extra)
    a, b, c = runs["device native"][0], runs["device python"][0], runs["double native"][0]
    assert len(a) == len(b) == len(c)
    assert a == b                                                        # same engine calls, same bits
    np.testing.assert_allclose(a, c, rtol=2e-6)                          # the oracle's collapsed likelihood (float32 sums: H1)
    for x, y in zip(runs["device native"][1], runs["device python"][1]):
        assert np.array_equal(x, y)
