import os, sys, numpy as np
sys.path.insert(0, '.')
os.environ["SBE_MFMA_DEBUG_FILE"] = "gpurun_out/mfma_dbg.bin"
from sbayes_amd.engine import Engine, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_MFMA
N, F, S, K = 16, 8, 4, 1
rng = np.random.default_rng(0)
x = rng.integers(0, S, size=(N, F))
feats = np.zeros((N, F, S), dtype=bool)
feats[np.arange(N)[:, None], np.arange(F)[None, :], x] = True
a = rng.integers(0, 2 * K, size=N)
groups = [np.stack([a == k for k in range(K)]), np.ones((1, N), dtype=bool)]
w = rng.dirichlet(np.ones(2), size=F).astype(np.float32)
probs = [np.full((K, F, S), 0.25, dtype=np.float32), np.full((1, F, S), 0.5, dtype=np.float32)]
with Engine(feats, [K, 1], n_slots=1) as eng:
    eng.load_state(0, groups, w, probs=probs)
    eng.set_option(kernel=MIXTURE_PACKED_TUPLE_MFMA)
    print(eng.mixture_loglik(0))
d = np.fromfile("gpurun_out/mfma_dbg.bin", dtype=np.float64).reshape(2, 4, 16, 64, 8)
np.set_printoptions(linewidth=250, precision=3, suppress=True)
cl = np.where(groups[0].any(0), groups[0].argmax(0), K)
print("tuple ids", cl)
for t in range(K + 1):
    print("want cnt t", t, feats[cl == t].sum(0).reshape(-1))
r, m = 0, 0
for reg in range(16):
    print("reg", reg, "cnt lanes0-31", d[r, m, reg, :32, 0].astype(int), "| lanes32-63", d[r, m, reg, 32:, 0].astype(int))
for reg in (0, 8):
    for k, nm in enumerate(["cnt", "v", "w0", "p0", "w1", "p1", "col4", "fw4"]):
        print("reg", reg, nm, d[r, m, reg, :40, k])
print("weights", w)
