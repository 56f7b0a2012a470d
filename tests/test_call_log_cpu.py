"""The recorded engine-level call logs of the real sampler (tests/golden/*_calls.npz, tests/_call_log.py) load and
replay consistently against the oracle-backed double: checks the replay harness itself on CPU; the same replay runs
against the real Engine in tests/test_gpu_call_log.py."""
import numpy as np
import pytest

from tests._call_log import replay
from tests._fake_engine import FakeEngine
from tests._fixtures import GOLDEN, crc, load_npz


def features_of(tag):
    tag = tag.replace("_gibbs", "")                  # (<shape>_gibbs_calls.npz: the same data, patch.install(gibbs_source=True))
    if tag in ("cfg1", "headline"):
        from sbayes_amd.synthetic import make_workload
        return make_workload(tag).features
    return load_npz(tag).features


@pytest.mark.parametrize("tag", ["test_files", "south_america", "cfg1", "headline"])
def test_call_log_replays_on_the_double(tag):
    feats = features_of(tag)
    counts, meta = replay(GOLDEN / f"{tag}_calls.npz", lambda n_groups: FakeEngine(feats, n_groups))
    assert crc(feats) == meta["features_crc"] and list(feats.shape) == meta["shape"]
    # the operator forms and the collapsed likelihood really ran through the engine surface
    assert {"cluster_marginals", "cluster_posterior_marginals", "source_posterior", "given_unchanged_lh", "collapsed_loglik_all",
            "counts_delta", "source_prior", "collapsed_and_source_prior", "set_counts_rows", "set_slot_delta", "set_groups",
            "set_weights", "__step__"} <= set(counts)
    # round 3: no whole [N, F] mask and no stateless whole-table call is left on the per-step path
    assert not {"effect_counts", "dirichlet_logpdf"} & set(counts)
    assert {"AlterCluster", "GibbsSampleSource"} <= set(meta["operators"])
    # through the bind cache: fewer uploads than evaluations
    assert counts["set_groups"] < counts["cluster_marginals"] + counts["cluster_posterior_marginals"] + counts["source_posterior"]


@pytest.mark.parametrize("tag", ["south_america_gibbs", "headline_gibbs", "cfg1_gibbs"])
def test_gibbs_source_call_log_replays_on_the_double(tag):
    """The logs recorded under patch.install(gibbs_source=True): GibbsSampleSource._propose's body as ONE engine call on slot
    state (gibbs_propose, with the uniforms regenerated from their recorded generator state) and the cluster operators'
    source resampling as another (given_unchanged_gibbs)."""
    feats = features_of(tag)
    counts, meta = replay(GOLDEN / f"{tag}_calls.npz", lambda n_groups: FakeEngine(feats, n_groups))
    assert meta["gibbs_source"] and crc(feats) == meta["features_crc"]
    assert {"gibbs_propose", "counts_delta", "given_unchanged_gibbs"} <= set(counts)
    assert not {"copy_slot", "sample_source", "update_counts", "source_logprob", "get_source_rows"} & set(counts)
    assert counts["gibbs_propose"] >= 10
    assert "GibbsSampleSource" in meta["operators"]
