import sys, numpy as np
sys.path.insert(0, ".")
from sbayes_amd.engine import Engine
from tests.test_gpu_delta_forms import _workload
feats, groups, conc, weights, source, counts, unif = _workload("headline")
ng = [g.shape[0] for g in groups]
eng = Engine(feats, ng, n_slots=2)
for c in range(len(groups)):
    eng.set_concentration(c, conc[c]); eng.set_groups(0, c, groups[c]); eng.set_counts(0, c, counts[c])
eng.set_source(0, source); eng.set_weights(0, weights); eng.set_uniform_counts(unif); eng.update_probs(0, range(len(groups)))
av = np.flatnonzero(~groups[0].any(axis=0) | groups[0][0]).astype(np.int32)
for n in (16, 587):
    rows = []
    for rep in range(200):
        o = eng.cluster_posterior_marginals(0, 0, av[:n], 1.0, 1.0)
        rows.append(o[0, :10].copy())
    r = np.median(np.array(rows[50:]), axis=0) / 100.0   # 100 MHz -> us
    print(f"n={n}: objects[i] {r[4]:.2f}  gid/pid/bits {r[5]:.2f}  state x {r[6]:.2f}  fetch (weights rows, probs) {r[0]:.2f} | builder: loads {r[7]:.2f} sum {r[8]:.2f} divisions+stores {r[9]:.2f} done {r[1]:.2f} | after barrier {r[2]:.2f} us, end {r[3]:.2f} us")
