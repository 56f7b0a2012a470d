"""Large results streamed by the kernel (include/sbe_engine.h "how a large result reaches the caller"):
`sbe_component_lh` (a1) and `sbe_likelihood_per_component` (a3) store their `[N, F]` / `[N, F, C]` float64 result into
host-mapped staging themselves, chunk after chunk in order, and raise one flag per completed chunk; host threads copy a chunk
to the caller as soon as its flag shows the call's sequence number.  If a flag could overtake its chunk's data, the host would copy what the PREVIOUS call left
in the staging buffer.  So every call here produces a result that differs from the previous call's in (nearly) every
element -- two table variants, alternating -- and is compared with the expected array in full, a few thousand times per
path; the caller's array is poisoned before every call.  The forms must agree bit for bit: streamed + pool, streamed on the
calling thread alone (SBE_D2H_THREADS=1), copy engine (SBE_STREAM_RESULTS=0)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
REPS = int(os.environ.get("SBE_STREAM_REPS", "3000"))        # (soak runs: SBE_STREAM_REPS=300000)


def _run(shape, reps):
    from oracle import sbayes_oracle as orc
    wl = make_workload("stream", shape=shape)
    N, F, S = wl.shape
    C = wl.n_components
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    rng = np.random.default_rng(2)
    with Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=2) as eng:
        # two states whose tables differ everywhere: slot 0 and slot 1
        variants = []
        for v in range(2):
            cts = [c + (7.0 * v) * (1 + np.arange(c.shape[-1], dtype=np.float32)) for c in counts]
            for c in range(C):
                eng.set_concentration(c, wl.concentration[c])
            eng.load_state(v, wl.groups, wl.weights, source=wl.source, counts=cts)
            for c in range(C):
                eng.update_probs(v, c)
            probs = [eng.get_probs(v, c) for c in range(C)]
            lh = orc.likelihood_per_component(wl.features, wl.na_values, wl.groups, cts, wl.concentration)
            a1 = np.full((N, F), -3.0)
            orc.compute_component_likelihood(wl.features, probs[0], wl.groups[0], np.arange(wl.groups[0].shape[0]), a1)
            variants.append((probs, lh, a1))
        assert (variants[0][1] != variants[1][1])[~wl.na_values].mean() > 0.5           # the two results differ (nearly) everywhere
        buf = np.empty((N, F, C))
        view = np.empty((N, F, 3))                                                       # a1 writes a strided view of it
        all_groups = np.arange(wl.groups[0].shape[0])
        for i in range(reps):
            v = i & 1
            probs, lh, a1 = variants[v]
            buf.fill(np.nan)
            got = eng.likelihood_per_component(v, buf)
            assert got.tobytes() == lh.tobytes(), ("a3", i)
            view.fill(-3.0)                                                               # (rows of no-group objects are zeroed, the rest written)
            eng.component_lh(probs[0], wl.groups[0], all_groups, view[..., 1])
            assert view[..., 1].tobytes() == a1.tobytes(), ("a1", i)
            assert (view[..., 0] == -3.0).all() and (view[..., 2] == -3.0).all()        # the neighbours in the strided buffer are untouched
        return eng.last_mixture_kernel() is not None


@pytest.mark.parametrize("shape", [(1000, 200, 10, 5, (), False),          # headline: a1 1.6 MB / a3 3.2 MB, 13 / 13 chunks
                                   (700, 131, 7, 3, (4,), True),            # odd sizes: a partial last chunk, an odd element count
                                   (5000, 500, 20, 10, (20, 20), False)],   # stress: 20 / 80 MB (the 2 ms + 10 GB/s fallback window)
                         ids=["headline", "odd", "stress"])
def test_streamed_results_are_never_stale(shape):
    reps = REPS if shape[0] <= 1000 else max(20, REPS // 100)
    assert _run(shape, reps)


_CHILD = r"""
import sys
sys.path.insert(0, %r)
from tests.test_gpu_streamed_results import _run
assert _run((1000, 200, 10, 5, (), False), 400)
assert _run((700, 131, 7, 3, (4,), True), 400)
print("OK")
"""


@pytest.mark.parametrize("env", [{"SBE_D2H_THREADS": "1"}, {"SBE_STREAM_RESULTS": "0"}, {"SBE_STREAM_RESULTS": "0", "SBE_D2H_THREADS": "1"},
                                 {"SBE_STEP_THREADS": "3"}, {"SBE_STREAM_ORDERED": "0"}, {"SBE_STREAM_CHUNKS": "16"}],
                         ids=["one_thread", "copy_engine", "copy_engine_one_thread", "pool_of_3", "unordered_chunks", "16_chunks"])
def test_other_forms_give_the_same_bits(env):
    res = subprocess.run([sys.executable, "-c", _CHILD % str(REPO)], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, **env), cwd=str(REPO))
    assert res.returncode == 0 and "OK" in res.stdout, res.stderr[-3000:]
