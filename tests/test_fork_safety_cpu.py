"""The engine under the reference's own process model (sbayes/mcmc_setup.py:271-299, cli.py:101-109).

MC3 workers and run pools are started with multiprocessing's default method -- fork on Linux.  These tests drive the REAL
`sbayes_amd.engine.Engine` / `registry` / `_proc` code over a stand-in for the ctypes library (a handle is an integer,
`sbe_destroy` calls are recorded in a file both processes can see), so what is checked is the host logic around fork():
the child forgets inherited handles without destroying them, sees an empty registry, cannot create engines when the
parent held a live HIP context -- and can when it did not; the parent's handle is untouched throughout."""
import ctypes as ct
import multiprocessing as mp
import os
import pickle
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import _lib, _proc, engine as engine_mod, registry

REPO = Path(__file__).resolve().parent.parent


class _StandInLib:
    """What Engine.__init__ / close() / na_values() need of the C ABI, without a device."""

    def __init__(self, log_path):
        self.log_path = log_path
        self.next_handle = 0x1000

    def _log(self, what):
        with open(self.log_path, "a") as fh:
            fh.write(f"{os.getpid()} {what}\n")

    def sbe_create(self, href, device, n, f, s, c, ng, n_slots, feats):
        self.next_handle += 0x10
        ct.cast(href, ct.POINTER(ct.c_void_p))[0] = self.next_handle
        self._log(f"create {self.next_handle:#x}")
        return 0

    def sbe_destroy(self, h):
        self._log(f"destroy {h.value:#x}")
        return 0

    def sbe_get_na(self, h, out):
        self._log(f"get_na {h.value:#x}")
        return 0

    def sbe_set_option(self, h, key, value):
        return 0

    def sbe_last_error(self, h):
        return b""


@pytest.fixture
def standin(tmp_path, monkeypatch):
    lib = _StandInLib(tmp_path / "calls.log")
    monkeypatch.setattr(_lib, "load", lambda: lib)
    monkeypatch.setattr(_proc, "_HIP_PID", None)
    monkeypatch.setattr(_proc, "_FORKED_FROM", None)
    registry._ENGINES.clear()
    registry._KNOWN.clear()
    yield lib
    for eng, _ in list(registry._ENGINES.values()):
        eng._h = ct.c_void_p()              # nothing real to destroy
    registry._ENGINES.clear()
    registry._KNOWN.clear()


def _calls(lib):
    if not os.path.exists(lib.log_path):
        return []
    return [ln.split() for ln in open(lib.log_path).read().splitlines()]


def _child_report(conn, feats):
    """Runs in the fork()ed child."""
    out = {"registry_empty": len(registry._ENGINES) == 0 and len(registry._KNOWN) == 0,
           "forked_from": _proc._FORKED_FROM, "hip_touched": _proc.hip_touched()}
    eng = _child_report.inherited
    out["handle_nulled"] = not bool(eng._h)
    try:
        eng.na_values()
        out["inherited_call"] = "no error"
    except _proc.ForkedWithHipError as exc:
        out["inherited_call"] = str(exc)
    try:
        registry.get_engine(feats, [2, 1])
        out["create"] = "no error"
    except _proc.ForkedWithHipError as exc:
        out["create"] = str(exc)
    try:
        engine_mod.Engine(feats, [1])
        out["create_direct"] = "no error"
    except _proc.ForkedWithHipError as exc:
        out["create_direct"] = str(exc)
    eng.close()                               # must not reach sbe_destroy
    del eng
    conn.send(out)
    conn.close()


def test_fork_after_hip_was_touched_child_forgets_and_guard_fires(standin):
    feats = np.zeros((6, 4, 3), dtype=bool)
    eng = registry.get_engine(feats, [2, 1])
    handle = eng._h.value
    assert handle and _proc.hip_touched() and eng._pid == os.getpid()
    _child_report.inherited = eng
    ctx = mp.get_context("fork")
    parent_conn, child_conn = ctx.Pipe()
    proc = ctx.Process(target=_child_report, args=(child_conn, feats))
    proc.start()
    out = parent_conn.recv()
    proc.join(30)
    assert proc.exitcode == 0
    assert out["registry_empty"] and out["handle_nulled"] and out["forked_from"] == os.getpid() and not out["hip_touched"]
    for key in ("inherited_call", "create", "create_direct"):
        assert "fork" in out[key] and "forkserver" in out[key] and "set_start_method" in out[key], out[key]
    assert f"process {os.getpid()}" in out["create"]
    # nothing of the parent changed: same handle, same registry entry, still usable; the child destroyed nothing
    assert eng._h.value == handle and registry.get_engine(feats, [2, 1]) is eng
    eng.na_values()
    calls = _calls(standin)
    assert [c for c in calls if c[1] == "destroy"] == []
    assert {c[0] for c in calls} == {str(os.getpid())}           # every library call so far came from the parent
    eng.close()
    assert [c[1:] for c in _calls(standin) if c[1] == "destroy"] == [["destroy", f"{handle:#x}"]]
    registry._ENGINES.clear()


def _child_creates(conn, feats):
    try:
        eng = registry.get_engine(feats, [1])
        conn.send(("ok", bool(eng._h), _proc.hip_touched(), len(registry._ENGINES)))
        eng.close()
    except Exception as exc:               # noqa: BLE001
        conn.send(("error", repr(exc)))
    conn.close()


def test_fork_before_hip_was_touched_child_is_a_fresh_process(standin):
    """MC3's first run: the workers are forked before the parent's first swap_chains touches the GPU."""
    feats = np.zeros((5, 3, 2), dtype=bool)
    registry.note_features(feats, [1])
    assert not _proc.hip_touched()
    ctx = mp.get_context("fork")
    parent_conn, child_conn = ctx.Pipe()
    proc = ctx.Process(target=_child_creates, args=(child_conn, feats))
    proc.start()
    out = parent_conn.recv()
    proc.join(30)
    assert proc.exitcode == 0 and out == ("ok", True, True, 1), out
    assert not _proc.hip_touched() and len(registry._ENGINES) == 0      # the parent still has not touched anything


def test_a_handle_is_never_destroyed_outside_its_process(standin):
    eng = engine_mod.Engine(np.zeros((2, 2, 2), dtype=bool), [1])
    eng._pid = os.getpid() + 1             # as if the object had reached this process by some other road
    eng.close()
    assert not eng._h and [c for c in _calls(standin) if c[1] == "destroy"] == []


def test_failed_create_marks_the_process(standin, monkeypatch):
    """The process is marked on the ATTEMPT: a create that fails after hipSetDevice / hipMalloc has initialised the
    runtime all the same, and a child forked afterwards must not be taken for a fresh process."""
    monkeypatch.setattr(standin, "sbe_create", lambda *a: 7)
    monkeypatch.setattr(standin, "sbe_last_error", lambda h: b"out of memory")
    assert not _proc.hip_touched()
    with pytest.raises(engine_mod.EngineError):
        engine_mod.Engine(np.zeros((2, 2, 2), dtype=bool), [1])
    assert _proc.hip_touched()


def test_start_method_helper(monkeypatch):
    from sbayes_amd import patch
    state = {"method": None}
    monkeypatch.setattr(mp, "get_start_method", lambda allow_none=False: state["method"])
    monkeypatch.setattr(mp, "set_start_method", lambda m, force=False: state.__setitem__("method", m))
    assert patch.set_mp_start_method("auto") == "forkserver" and state["method"] == "forkserver"
    assert patch.set_mp_start_method("forkserver") == "forkserver"
    with pytest.raises(RuntimeError, match="already fixed to 'forkserver'"):
        patch.set_mp_start_method("spawn")
    with pytest.raises(ValueError):
        patch.set_mp_start_method("fork")
    state["method"] = "fork"
    with pytest.warns(RuntimeWarning, match="fixed to 'fork'"):
        assert patch.set_mp_start_method("auto") == "fork"


REF = "/root/reference"
_UNPICKLE_IN_FRESH_INTERPRETER = r"""
import pickle, sys
sys.path.insert(0, {repo!r}); sys.path.insert(0, {golden!r})
import _ref_stubs; _ref_stubs.install()
import sbayes.model                                   # (the reference's import order: model before sampling.state)
from sbayes_amd import patch
assert patch.installed() is None
lik = pickle.load(open({path!r}, "rb"))
import sbayes.sampling.conditionals as cond, sbayes.sampling.operators as ops, sbayes.model.model as mm
import sbayes_amd.likelihood as my
assert patch.installed() == {{"operators": True, "gibbs_source": False}}, patch.installed()
assert cond.compute_component_likelihood is my.compute_component_likelihood
assert mm.Likelihood is my.Likelihood and type(lik) is my.Likelihood
assert ops.component_likelihood_given_unchanged.__module__ == "sbayes_amd.operators"
print("REINSTALLED")
"""


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sBayes not present")
def test_unpickling_a_likelihood_reinstalls_the_patch_in_a_fresh_interpreter(tmp_path):
    """spawn / forkserver workers are fresh interpreters: the pickled model's Likelihood carries how the sender had
    sBayes patched and re-installs that when it is unpickled (mcmc_setup.py:299, :554 send the model first)."""
    sys.path.insert(0, str(REPO / "tests" / "golden"))
    import _ref_stubs
    _ref_stubs.install()
    import sbayes.model  # noqa: F401
    from sbayes_amd import model as sbm, patch
    from tests._fixtures import load_npz
    fx = load_npz("cfg1")
    model, _ = sbm.build(fx.features, fx.states_per_feature, ["clusters", "universal"], fx.groups, fx.conc, fx.weights,
                         fx.source)
    assert "_sbayes_amd_patch" in model.likelihood.__getstate__() and model.likelihood.__getstate__()["_sbayes_amd_patch"] is None
    patch.install(operators=True)
    try:
        blob = pickle.dumps(model.likelihood)
    finally:
        patch.uninstall()
    path = tmp_path / "likelihood.pickle"
    path.write_bytes(blob)
    code = _UNPICKLE_IN_FRESH_INTERPRETER.format(repo=str(REPO), golden=str(REPO / "tests" / "golden"), path=str(path))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(REPO))
    assert res.returncode == 0 and "REINSTALLED" in res.stdout, res.stderr[-3000:]
