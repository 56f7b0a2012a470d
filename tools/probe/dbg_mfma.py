import sys, numpy as np
sys.path.insert(0, '.')
from sbayes_amd.engine import Engine, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA
def run(N, F, S, K, seed=0, uniform=False, na=0.0):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, S, size=(N, F))
    feats = np.zeros((N, F, S), dtype=bool)
    feats[np.arange(N)[:, None], np.arange(F)[None, :], x] = True
    feats[rng.random((N, F)) < na] = False
    a = rng.integers(0, 2 * K, size=N)
    groups = [np.stack([a == k for k in range(K)]), np.ones((1, N), dtype=bool)]
    w = rng.dirichlet(np.ones(2), size=F).astype(np.float32)
    if uniform:
        probs = [np.full((K, F, S), 1.0 / S, dtype=np.float32), np.full((1, F, S), 1.0 / S, dtype=np.float32)]
    else:
        probs = [rng.dirichlet(np.ones(S), size=(K, F)).astype(np.float32), rng.dirichlet(np.ones(S), size=(1, F)).astype(np.float32)]
    with Engine(feats, [K, 1], n_slots=1) as eng:
        eng.load_state(0, groups, w, probs=probs)
        out = {}
        for name, k in (("t64", MIXTURE_PACKED_TUPLE), ("mfma", MIXTURE_PACKED_TUPLE_MFMA)):
            eng.set_option(kernel=k)
            out[name] = eng.mixture_loglik(0)
    print(N, F, S, K, "uniform" if uniform else "random", out, "rel", abs(out["mfma"] - out["t64"]) / abs(out["t64"]), flush=True)
run(16, 8, 4, 1, uniform=True)
run(16, 8, 4, 1)
run(32, 8, 4, 1)
run(33, 8, 4, 1)
run(50, 30, 5, 2)
run(50, 30, 5, 2, na=0.1)
run(200, 70, 7, 3, na=0.05)
