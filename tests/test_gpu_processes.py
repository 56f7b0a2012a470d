"""The engine under the reference's process model, on hardware.

* MC3's shape on one GPU: two FRESH worker processes (spawn context, the way sbayes_amd.patch.install(mp_start_method=...)
  sets the reference up; mcmc_setup.py:271-299) each receive a pickled (model, sample) over a Pipe, create their own engine
  on device 0 and replay 50 steps of the recorded south_america trace; their scalars equal the single-process run bit for
  bit, while the parent keeps using its own engine.
* fork() from a process with a live HIP context: the child forgets the inherited handle (no sbe_destroy), sees an empty
  registry, the guard names the fix; the parent's engine keeps returning the same bits.  The child makes NO HIP call and
  leaves through os._exit (multiprocessing's fork children do the same), so the parent's context is never touched."""
import json
import multiprocessing as mp
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import model as sbm
from sbayes_amd.registry import release_all
from tests import _mc3_worker
from tests._fixtures import load_npz, load_trace

pytestmark = pytest.mark.gpu
REPO = Path(__file__).resolve().parent.parent
N_STEPS = 50


def test_two_spawned_workers_each_with_their_own_engine():
    name = "south_america"
    fx, tr = load_npz(name), load_trace(name)
    names = fx.meta.get("component_names")
    model, sample = sbm.build(fx.features, fx.states_per_feature, names, fx.groups, fx.conc, fx.weights, fx.source,
                              counts=fx.counts)
    try:
        want_lik = float(model.likelihood(sample, caching=False))          # the parent has a live engine of its own
        want = _mc3_worker.replay(model, sample, fx, tr, N_STEPS)
        ctx = mp.get_context("spawn")
        workers = []
        for _ in range(2):
            parent_conn, child_conn = ctx.Pipe()
            proc = ctx.Process(target=_mc3_worker.worker_main, args=(child_conn, name, N_STEPS))
            proc.start()
            parent_conn.send((model, sample))                             # pickled: no handle travels (Engine.__getstate__ raises)
            workers.append((proc, parent_conn))
        # the parent keeps evaluating while the workers run (swap_chains does, mcmc_setup.py:389-395)
        assert float(model.likelihood(sample, caching=False)) == want_lik
        pids = set()
        for proc, conn in workers:
            assert conn.poll(600), "worker did not answer"
            msg = conn.recv()
            proc.join(60)
            assert msg[0] == "ok", msg[-1]
            assert proc.exitcode == 0
            _, pid, lik_ll, got = msg
            pids.add(pid)
            assert lik_ll == want_lik
            assert got == want                                            # collapsed, mixture and per-group scalars, bit for bit
        assert len(pids) == 2 and os.getpid() not in pids
        for i in range(N_STEPS):
            assert abs(want[i][1] - tr.mixture_ll[i]) <= 1e-10 * abs(tr.mixture_ll[i])
            assert abs(want[i][0] - tr.last_lh[i]) <= 1e-6 * abs(tr.last_lh[i])
    finally:
        release_all()


_FORK_PROBE = r"""
import json, os, sys
sys.path.insert(0, {repo!r})
import numpy as np
from sbayes_amd import _proc, registry
from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload

wl = make_workload("cfg1")
eng = registry.get_engine(wl.features, [g.shape[0] for g in wl.groups])
for c in range(wl.n_components):
    eng.set_concentration(c, wl.concentration[c])
eng.load_state(0, wl.groups, wl.weights, source=wl.source)
for c in range(wl.n_components):
    eng.update_probs(0, c)
before = eng.mixture_loglik(0)
handle = eng._h.value
r, w = os.pipe()
pid = os.fork()
if pid == 0:                                  # child: NO HIP call is made here
    os.close(r)
    out = dict(registry_empty=not registry._ENGINES, handle_nulled=not bool(eng._h), forked_from=_proc._FORKED_FROM)
    for tag, fn in (("inherited", lambda: eng.mixture_loglik(0)),
                    ("create", lambda: registry.get_engine(wl.features, [g.shape[0] for g in wl.groups])),
                    ("direct", lambda: Engine(wl.features, [1]))):
        try:
            fn()
            out[tag] = "no error"
        except _proc.ForkedWithHipError as exc:
            out[tag] = str(exc)
    eng.close()
    os.write(w, json.dumps(out).encode())
    os._exit(0)
os.close(w)
child = json.loads(os.read(r, 1 << 16).decode())
_, status = os.waitpid(pid, 0)
after = eng.mixture_loglik(0)
print(json.dumps(dict(child=child, status=status, before=before, after=after, same_handle=eng._h.value == handle,
                      parent=os.getpid())))
eng.close()
"""


def test_fork_from_a_process_with_a_live_context():
    res = subprocess.run([sys.executable, "-c", _FORK_PROBE.format(repo=str(REPO))], capture_output=True, text=True,
                         timeout=600, cwd=str(REPO))
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    child = out["child"]
    assert out["status"] == 0 and out["same_handle"] and out["before"] == out["after"]
    assert child["registry_empty"] and child["handle_nulled"] and child["forked_from"] == out["parent"]
    for tag in ("inherited", "create", "direct"):
        assert "forkserver" in child[tag] and "fork()" in child[tag], child[tag]
