"""Diagnostic: time sbe_cluster_marginals at the headline shape; run under rocprofv3 for
the per-kernel split."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload
wl = make_workload("headline")
eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=1)
for c in range(wl.n_components):
    eng.set_concentration(c, wl.concentration[c])
eng.load_state(0, wl.groups, wl.weights, source=wl.source)
for c in range(wl.n_components):
    eng.update_probs(0, c)
available = np.flatnonzero((~wl.clusters.any(axis=0)) | wl.clusters[0])
table = eng.get_probs(0, 0)[0]
for _ in range(50):
    eng.cluster_marginals(0, table, available)
t0 = time.perf_counter(); n = 1000
for _ in range(n):
    eng.cluster_marginals(0, table, available)
print("cluster_marginals", round((time.perf_counter() - t0) / n * 1e6, 1), "us/call for", available.size, "objects")
# the operator form (what patch.install(operators=True) puts under AlterCluster): binds the sample, builds the
# candidate table, evaluates
from sbayes_amd import model as sbm
from sbayes_amd.counts import recalculate_feature_counts
from sbayes_amd.operators import calculate_source_posterior, compute_cluster_posterior
model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration, wl.weights, wl.source)
recalculate_feature_counts(model.data.features.values, sample)
avail = np.zeros(wl.shape[0], dtype=bool); avail[available] = True
for _ in range(20):
    compute_cluster_posterior(model, sample, 0, avail)
t0 = time.perf_counter(); n = 300
for _ in range(n):
    compute_cluster_posterior(model, sample, 0, avail)
print("operators.compute_cluster_posterior", round((time.perf_counter() - t0) / n * 1e6, 1), "us/call")
objs = np.arange(0, 1000, 50)
t0 = time.perf_counter()
for _ in range(n):
    calculate_source_posterior(model, sample, objs)
print("operators.calculate_source_posterior (20 objects)", round((time.perf_counter() - t0) / n * 1e6, 1), "us/call")
