"""Where do the microseconds of a 64-chain sbe_step_batch sweep go?  (GPU box)
   raw C call (arguments marshalled once)  |  Engine.step_batch (argument checks / conversions)  |
   ResidentChainBatch.step_arrays + accept (the Python face)  |  device chain (HIP events are not used: the C call is
   synchronous, its duration is host prepare + enqueue + device chain + sync)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from sbayes_amd import model as sbm                                 # noqa: E402
from sbayes_amd.resident import ResidentChainBatch                  # noqa: E402
from sbayes_amd.synthetic import make_workload                      # noqa: E402

wl = make_workload(sys.argv[1] if len(sys.argv) > 1 else "headline")
n_chains = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration, wl.weights, wl.source)
batch = ResidentChainBatch(model, [sample] * n_chains)
rng = np.random.default_rng(0)
n_obj = wl.shape[0]
cl = np.broadcast_to(wl.clusters, (n_chains,) + wl.clusters.shape).copy()
objs_all, ptr = [], [0]
for i in range(n_chains):
    n = int(rng.integers(0, n_obj))
    cl[i][:, n] = False
    cl[i][int(rng.integers(0, cl.shape[1])), n] = True
    objs = np.unique(np.append(rng.integers(0, n_obj, size=19), n)).astype(np.int32)
    objs_all.append(objs); ptr.append(ptr[-1] + objs.size)
objs_cat = np.concatenate(objs_all); ptr = np.array(ptr, dtype=np.int32)
rows = np.ascontiguousarray(wl.source[objs_cat])


def rate(fn, n=200):
    for _ in range(10):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e6


def face():
    batch.step_arrays(cl, None, ptr, objs_cat, rows)
    batch.accept()


def engine_only():
    batch.eng.step_batch(batch.cur, batch.cand, cl, None, ptr, objs_cat, rows)


eng = batch.eng
import ctypes as ct                                                  # noqa: E402
from sbayes_amd.engine import _c, _ptr                                # noqa: E402
cur = np.ascontiguousarray(batch.cur); cand = np.ascontiguousarray(batch.cand)
clu = _c(cl, np.uint8); rowsu = _c(rows, np.uint8)
glh = np.empty((n_chains, eng.n_groups_total)); mix = np.empty(n_chains); changed = np.zeros((n_chains, eng.n_groups_total), dtype=np.uint8)
args = (eng._h, n_chains, _ptr(cur), _ptr(cand), _ptr(clu), None, _ptr(ptr), _ptr(objs_cat), _ptr(rowsu), None, None, _ptr(glh), _ptr(mix), _ptr(changed))


def raw():
    eng._lib.sbe_step_batch(*args)


# the same sweep in delta form: moved objects + new cluster instead of the [K, N] matrices
mptr, mobj, mcl = [0], [], []
for i in range(n_chains):
    moved = np.flatnonzero((cl[i] != wl.clusters).any(axis=0)).astype(np.int32)
    mobj.append(moved)
    mcl.append(np.where(cl[i][:, moved].any(axis=0), cl[i][:, moved].argmax(axis=0), -1).astype(np.int32))
    mptr.append(mptr[-1] + moved.size)
mptr = np.array(mptr, dtype=np.int32); mobj = np.concatenate(mobj); mcl = np.concatenate(mcl)
dargs = (eng._h, n_chains, _ptr(cur), _ptr(cand), _ptr(mptr), _ptr(mobj), _ptr(mcl), _ptr(ptr), _ptr(objs_cat), _ptr(rowsu), None, None,
         _ptr(glh), _ptr(mix), _ptr(changed))


def raw_delta():
    eng._lib.sbe_step_batch_delta(*dargs)


def face_delta():
    batch.step_delta(mptr, mobj, mcl, ptr, objs_cat, rows)
    batch.accept()


print(f"{n_chains} chains, {wl.name}, DELTA form: raw C call {rate(raw_delta):8.1f} us | ResidentChainBatch.step_delta + accept {rate(face_delta):8.1f} us")
print(f"{n_chains} chains, {wl.name}: raw C call {rate(raw):8.1f} us | Engine.step_batch {rate(engine_only):8.1f} us | "
      f"ResidentChainBatch.step_arrays + accept {rate(face):8.1f} us")
batch.close()
