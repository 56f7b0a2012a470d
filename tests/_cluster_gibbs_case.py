"""ClusterOperator.gibbs_sample_source (sbayes/sampling/operators.py:796-851) against the REFERENCE's recorded outputs --
TEST INFRASTRUCTURE shared by the CPU (oracle-backed double) and GPU tests.  tests/golden/gibbs_source.npz, cases `cg_*`
(make_golden.py gibbs_source_fixture): the reference's own method on south_america with pinned moved objects and a pinned
np.random stream -- objects grown into / shrunk out of a cluster, MC3 temperatures, sample_from_prior; recorded: the new
source array, the updated counts of every component and the float32 log_q / log_q_back."""
import numpy as np

from sbayes_amd import model as sbm
from tests._fixtures import load_npz

CASES = ["grow", "shrink", "mc3", "prior"]


def run_case(tag):
    from sbayes_amd.operators import cluster_gibbs_sample_source
    fx = load_npz("gibbs_source")
    z = fx.z
    model, sample = sbm.build(fx.features, fx.states_per_feature, fx.meta["component_names"], fx.groups, fx.conc,
                              fx.weights, fx.source, counts=fx.counts)
    objects = z[f"cg_{tag}_objects"]
    i_cluster = int(z[f"cg_{tag}_i_cluster"])
    temp, ptemp, from_prior = (float(v) for v in z[f"cg_{tag}_temps"])
    sample_new = sample.copy()
    with sample_new.clusters.edit_cluster(i_cluster) as row:               # the proposal's cluster move (add / remove objects)
        row[:] = z[f"cg_{tag}_clusters_new"][i_cluster]
    assert np.array_equal(sample_new.clusters.value, z[f"cg_{tag}_clusters_new"])
    out, log_q, log_q_back = cluster_gibbs_sample_source(model, sample_new, sample, i_cluster, objects, temp, ptemp,
                                                         bool(from_prior), z=z[f"cg_{tag}_z"])
    assert out is sample_new
    assert np.array_equal(out.source.value, z[f"cg_{tag}_new_source"]), tag          # the draws, draw for draw
    for c, name in enumerate(sample.component_names):
        assert np.array_equal(out.feature_counts[name].value, z[f"cg_{tag}_counts_{c}"]), (tag, name)
    assert log_q.dtype == np.float32 and log_q_back.dtype == np.float32
    exact = temp == 1.0 and ptemp == 1.0
    return exact, (log_q, float(z[f"cg_{tag}_log_q"])), (log_q_back, float(z[f"cg_{tag}_log_q_back"])), sample, objects, fx
