"""Diagnostic: time the device form of the Gibbs source operator at the headline shape."""
import sys, time; sys.path.insert(0, ".")
import numpy as np
from sbayes_amd import model as sbm
from sbayes_amd.counts import recalculate_feature_counts
from sbayes_amd.operators import gibbs_sample_source, calculate_source_posterior
from sbayes_amd.synthetic import make_workload
wl = make_workload("headline")
model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration, wl.weights, wl.source)
recalculate_feature_counts(model.data.features.values, sample)
rng = np.random.default_rng(0)
for device_rng in (False, True):
    for n_obj in (20, 1000):
        objs = np.sort(rng.choice(1000, size=n_obj, replace=False))
        for _ in range(5):
            gibbs_sample_source(model, sample, objs, device_rng=device_rng)
        t0 = time.perf_counter(); n = 50
        for _ in range(n):
            gibbs_sample_source(model, sample, objs, device_rng=device_rng)
        print(f"gibbs_sample_source device_rng={device_rng} n={n_obj}: {(time.perf_counter() - t0) / n * 1e6:.0f} us/call")
objs = np.arange(1000)
t0 = time.perf_counter()
for _ in range(50):
    calculate_source_posterior(model, sample, objs)
print(f"calculate_source_posterior n=1000: {(time.perf_counter() - t0) / 50 * 1e6:.0f} us/call")
# resident form: nothing of the sample is re-uploaded
from sbayes_amd.resident import ResidentChain
chain = ResidentChain(model, sample)
for device_rng in (False, True):
    for n_obj in (20, 1000):
        objs = np.sort(rng.choice(1000, size=n_obj, replace=False))
        for _ in range(5):
            chain.propose_gibbs_source(objs, device_rng=device_rng); chain.reject()
        t0 = time.perf_counter(); n = 100
        for _ in range(n):
            cand, lq, lqb = chain.propose_gibbs_source(objs, device_rng=device_rng)
            ll, mix = cand.collapsed_loglik(), cand.mixture_loglik()
            chain.accept()
        print(f"resident gibbs step device_rng={device_rng} n={n_obj}: {(time.perf_counter() - t0) / n * 1e6:.0f} us/step (incl. collapsed + mixture eval)")
for device_rng in (False, True):
    for n_obj in (20, 1000):
        objs = np.sort(rng.choice(1000, size=n_obj, replace=False))
        for _ in range(5):
            chain.gibbs_step(objs, device_rng=device_rng); chain.reject()
        t0 = time.perf_counter(); n = 200
        for _ in range(n):
            chain.gibbs_step(objs, device_rng=device_rng)
            chain.accept()
        print(f"one-call gibbs step device_rng={device_rng} n={n_obj}: {(time.perf_counter() - t0) / n * 1e6:.0f} us/step")
