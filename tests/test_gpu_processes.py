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


_CHAIN_PROBE = r"""
import json, os, sys, time
sys.path.insert(0, {repo!r})
import numpy as np
from sbayes_amd.engine import Engine
from sbayes_amd.synthetic import make_workload

wl = make_workload("headline")
eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=2, device=0)
for c in range(wl.n_components):
    eng.set_concentration(c, wl.concentration[c])
eng.load_state(0, wl.groups, wl.weights, source=wl.source)
for c in range(wl.n_components):
    eng.update_probs(0, c)
ll = eng.mixture_loglik(0)
lh = eng.likelihood_per_component(0)
print("READY", flush=True)
start = float(sys.stdin.readline())                # every process starts its loops at the same wall-clock instant
while time.time() < start:
    pass
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < {seconds}:
    for _ in range(50):
        eng.mixture_loglik(0)                      # host-synchronous: the caller spins on the completion flag
    n += 50
evals = n / (time.perf_counter() - t0)
t0 = time.perf_counter(); m = 0
while time.perf_counter() - t0 < {seconds}:
    for _ in range(10):
        eng.likelihood_per_component(0)            # the streamed [N, F, C] result: the engine's host pool copies the chunks out
    m += 10
lh_calls = m / (time.perf_counter() - t0)
print(json.dumps(dict(pid=os.getpid(), ll=ll, lh_sum=float(lh.sum()), evals_per_s=evals, lh_calls_per_s=lh_calls,
                      cpus=os.cpu_count(), affinity=len(os.sched_getaffinity(0)),
                      cpu_max=(open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else None),
                      step_threads=os.environ.get("SBE_STEP_THREADS"))), flush=True)
eng.close()
"""


def _run_chain_processes(n, seconds=1.5):
    import time
    env = {k: v for k, v in os.environ.items() if k != "SBE_STEP_THREADS"}            # the engine's own default
    procs = [subprocess.Popen([sys.executable, "-c", _CHAIN_PROBE.format(repo=str(REPO), seconds=seconds)], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(REPO), env=env) for _ in range(n)]
    try:
        for p in procs:
            assert p.stdout.readline().strip() == "READY", p.stderr.read()[-2000:]
        start = time.time() + 0.3
        for p in procs:
            p.stdin.write(f"{start}\n"); p.stdin.flush()
        out = []
        for p in procs:
            so, se = p.communicate(timeout=300)
            assert p.returncode == 0, se[-2000:]
            out.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
        return out
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()


def test_four_chain_processes_on_one_card_with_default_spin_waits():
    """every synchronous call spins a core for up to 300 us and the engine's pool workers poll
    after a call; several chains = several such processes on one host.  Fresh child processes (never a fork or exec of a
    process that touched the GPU), each with its own engine on device 0 and the engine's DEFAULT thread settings, run the
    same host-synchronous loops at the same time; the per-process rates go to gpurun_out/ for the record.  Four, not eight:
    a GPU box of this pool admits at most six processes on its card, and the test runner itself (engines of earlier tests) is
    one of them (six children were measured in a run of their own: profiles/r5/six_processes_one_card_yielding_spins.json).
    Asserted: identical results in every process, no process starved, and the four together deliver at least what one
    delivers alone (the card serialises their kernels; their spin-waits must not make it worse than that)."""
    solo = _run_chain_processes(1)[0]
    six = _run_chain_processes(4)
    assert len({r["pid"] for r in six}) == 4
    assert all(r["ll"] == solo["ll"] and r["lh_sum"] == solo["lh_sum"] for r in six)
    agg_evals = sum(r["evals_per_s"] for r in six)
    agg_lh = sum(r["lh_calls_per_s"] for r in six)
    record = dict(what="one vs four concurrent single-chain processes on ONE MI355X, headline shape, default SBE_STEP_THREADS",
                  solo=solo, concurrent=six, aggregate_evals_per_s=agg_evals, aggregate_lh_calls_per_s=agg_lh,
                  evals_ratio_concurrent_over_solo=agg_evals / solo["evals_per_s"], lh_ratio_concurrent_over_solo=agg_lh / solo["lh_calls_per_s"])
    out_dir = REPO / "gpurun_out"
    try:
        out_dir.mkdir(exist_ok=True)
        (out_dir / "four_processes_one_card.json").write_text(json.dumps(record, indent=1))
    except OSError:
        pass
    print(json.dumps(record))
    slowest = min(r["evals_per_s"] for r in six)
    assert slowest >= solo["evals_per_s"] / 16, record                    # nobody starved (a fair share would be 1/4)
    assert agg_evals >= 0.8 * solo["evals_per_s"], record                 # contention does not eat the card
    # the streamed [N, F, C] result (3.2 MB per call through ONE PCIe link and the host's copy threads): measured 0.50 of
    # the solo rate for SIX processes, with and without yielding spin-waits (profiles/r5/six_processes_one_card*.json) --
    # recorded, and bounded here only against a collapse
    assert agg_lh >= 0.3 * solo["lh_calls_per_s"], record
