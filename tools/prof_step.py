import sys, time; sys.path.insert(0, ".")
import numpy as np
from sbayes_amd import model as sbm
from sbayes_amd.resident import ResidentChain
from sbayes_amd.synthetic import make_workload
wl = make_workload("headline")
model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration, wl.weights, wl.source)
chain = ResidentChain(model, sample)
clusters = wl.clusters.copy()
objs = np.arange(0, 1000, 50).astype(np.int32)
rows = wl.source[objs]
eng = chain.eng
for variant in ("full", "no_clusters", "no_rows", "nothing"):
    kw = {}
    if variant in ("full", "no_rows"): kw["clusters"] = clusters
    if variant in ("full", "no_clusters"): kw["source_rows"] = (objs, rows)
    for _ in range(50):
        chain.step(**kw); chain.accept()
    t0 = time.perf_counter(); n = 500
    for _ in range(n):
        chain.step(**kw); chain.accept()
    print(variant, round((time.perf_counter() - t0) / n * 1e6, 1), "us/step")
# the same step in delta form (sbe_step_delta): one object changes cluster back and forth, the same 20 rows
cl_of = np.where(clusters.any(axis=0), clusters.argmax(axis=0), -1)
obj = int(np.flatnonzero(cl_of >= 0)[0])
alt = [(int(cl_of[obj]) + 1) % clusters.shape[0], int(cl_of[obj])]
for _ in range(50):
    chain.step_delta([obj], [alt[_ % 2]], (objs, rows)); chain.accept()
t0 = time.perf_counter(); n = 500
for k in range(n):
    chain.step_delta([obj], [alt[k % 2]], (objs, rows)); chain.accept()
print("delta form: move + 20 rows", round((time.perf_counter() - t0) / n * 1e6, 1), "us/step")
t0 = time.perf_counter()
for k in range(n):
    chain.step_delta(None, None, (objs, rows)); chain.accept()
print("delta form: 20 rows", round((time.perf_counter() - t0) / n * 1e6, 1), "us/step")
# raw engine calls (no ResidentChain bookkeeping): matrix form against delta form
mo = np.array([obj], dtype=np.int32)
for form in ("matrix", "delta"):
    cur, cand = chain.cur, chain.cand
    t0 = time.perf_counter()
    for k in range(n):
        if form == "matrix":
            eng.step(cur, cand, clusters=clusters, changed_objects=objs, source_rows=rows)
        else:
            eng.step_delta(cur, cand, mo, np.array([alt[k % 2]], dtype=np.int32), objs, rows)
        cur, cand = cand, cur
    print("Engine-level", form, round((time.perf_counter() - t0) / n * 1e6, 1), "us/step")
chain.cur, chain.cand = cur, cand
t0 = time.perf_counter()
for _ in range(500): eng.copy_slot(chain.cand, chain.cur)
eng.sync(); print("copy_slot", round((time.perf_counter() - t0) / 500 * 1e6, 1), "us")
t0 = time.perf_counter()
for _ in range(500): eng.mixture_loglik(chain.cur)
print("mixture", round((time.perf_counter() - t0) / 500 * 1e6, 1), "us")
# C-level phase times of the one-call step (SBE_STEP_TIMING=1 prints to stderr every 2000 steps)
import os
if os.environ.get("SBE_STEP_TIMING") == "1":
    kw = {"clusters": clusters, "source_rows": (objs, rows)}
    for _ in range(4100):
        chain.step(**kw); chain.accept()
