import sys, ctypes as ct, numpy as np
sys.path.insert(0, ".")
from sbayes_amd.engine import Engine
from sbayes_amd import _lib
from tests.test_gpu_delta_forms import _workload
name = sys.argv[1] if len(sys.argv) > 1 else "headline"
feats, groups, conc, weights, source, counts, unif = _workload(name)
ng = [g.shape[0] for g in groups]
eng = Engine(feats, ng, n_slots=2)
for c in range(len(groups)):
    eng.set_concentration(c, conc[c]); eng.set_groups(0, c, groups[c]); eng.set_counts(0, c, counts[c])
eng.set_source(0, source); eng.set_weights(0, weights); eng.set_uniform_counts(unif); eng.update_probs(0, range(len(groups)))
lib = _lib.load()
lib.sbe_debug_gu_clk.argtypes = [ct.c_void_p, ct.c_void_p]
N, F, C = source.shape
rng = np.random.default_rng(0)
has = np.stack([g.any(axis=0) for g in groups], axis=1)
for n in (12, 100):
    objs = np.sort(rng.choice(N, size=min(n, N), replace=False)).astype(np.int32)
    hc = has[objs].copy(); hc[~hc.any(axis=1), 1] = True
    so = np.where(source[objs].any(-1), source[objs].argmax(-1), 255).astype(np.uint8)
    z = rng.random((objs.size, F))
    for kind in ("lh", "gibbs"):
        rows = []
        for rep in range(200):
            if kind == "lh": eng.given_unchanged_lh(0, 0, objs, 1.0, 1.0)
            else: eng.given_unchanged_gibbs(0, 0, objs, hc, hc, so, z, 1.0, 1.0, False)
            out = np.zeros(16, dtype=np.uint64)
            lib.sbe_debug_gu_clk(eng._h, out.ctypes.data)
            rows.append((out[1:6].astype(np.int64) - np.int64(out[0])) / 100.0)
        r = np.median(np.array(rows[50:]), axis=0)
        print(f"{name} n={n} {kind}: lists+bitmap-zero {r[0]:.2f}  bitmap {r[1]:.2f}  histograms (row 0 walk + subset) {r[2]:.2f}  normalise {r[3]:.2f}  consumer done {r[4]:.2f} us")
