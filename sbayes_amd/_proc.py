"""Process model of the engine: what happens to device handles when the host process forks.

The reference forks.  MC3 starts one `MCMCChainProcess(multiprocessing.Process)` worker per chain with `proc.start()`
(sbayes/mcmc_setup.py:271-282) and `multiprocessing.Pool` fans runs out (sbayes/cli.py:104-109); on Linux both use
the `fork` start method unless the application chose another one.  With `processes <= 1` the run x K loop
(cli.py:101-103) calls `sample_mc3` repeatedly in ONE parent, and that parent evaluates `model.likelihood(...)` itself
in `swap_chains` (mcmc_setup.py:389-395, :418) -- so from the second run on, workers are forked from a process that
has a live HIP context.  A HIP context does not survive fork(): the child inherits the parent's address space
(engine handles, pinned arenas, the runtime's queues and threads' state) but none of the runtime's threads, and any
HIP call on the inherited state hangs or corrupts the parent's device memory.

What this module guarantees ("device handles lazily (re)created per process, never pickled"):

* every `Engine` registers here; `os.register_at_fork(after_in_child=...)` makes the child FORGET every inherited
  handle -- `Engine._h` is nulled and the engine's library face is replaced by one that raises -- WITHOUT calling
  `sbe_destroy` (the handle's device memory belongs to the parent), and empties `registry._ENGINES` / `_KNOWN`;
* if the parent had touched the HIP runtime through this package before the fork (`mark_hip_touched`), the first
  attempt of the child to create an engine or count devices raises `ForkedWithHipError`, naming the fix;
* if the parent had NOT touched the runtime (MC3's first run: workers are forked before the parent's first
  `swap_chains`), the child is as good as a fresh process and creates its own engines.
The parent's state is never modified by a fork."""
from __future__ import annotations

import os
import weakref

_ENGINE_REFS: "weakref.WeakSet" = weakref.WeakSet()     # live Engine objects of this process
_HIP_PID = None         # pid of the process in which this package first touched the HIP runtime (None: not yet)
_FORKED_FROM = None     # in a forked child whose parent had touched HIP: the parent's pid
_CLEAR_HOOKS = []       # callables run in the child after a fork (registry.py: forget _ENGINES / _KNOWN)

FIX = ("Fix: choose a start method that does not copy the parent's HIP context BEFORE the first worker is started -- "
       "`multiprocessing.set_start_method('forkserver')` (or 'spawn'); `sbayes_amd.patch.install(mp_start_method=...)` "
       "does that -- or create engines only in the worker processes (never evaluate the likelihood in the parent "
       "before forking).  Each worker then creates its own engine on first use; handles are never shared.")


class ForkedWithHipError(RuntimeError):
    """Raised in a fork()ed child when it tries to use the GPU although its parent held a live HIP context."""


def mark_hip_touched():
    """Called right before this package's first HIP-initialising library call in a process."""
    global _HIP_PID
    check_usable()
    if _HIP_PID is None:
        _HIP_PID = os.getpid()


def hip_touched() -> bool:
    return _HIP_PID == os.getpid()


def check_usable():
    """Raise in a forked child of a HIP-initialised parent; free otherwise."""
    if _FORKED_FROM is not None:
        raise ForkedWithHipError(
            f"sbayes_amd: process {os.getpid()} was fork()ed from process {_FORKED_FROM}, which had already initialised "
            f"the HIP runtime (it created an engine or counted devices).  A HIP context does not survive fork(): the "
            f"engines of the parent were forgotten here (not destroyed), and this process cannot use the GPU.  {FIX}")


def register_engine(engine):
    _ENGINE_REFS.add(engine)


def on_fork_clear(fn):
    """Register a callable the child runs after fork (module-level caches that hold engines)."""
    _CLEAR_HOOKS.append(fn)
    return fn


class _ForgottenLib:
    """Library face of an engine inherited through fork(): every entry point raises."""

    def __init__(self, parent_pid):
        self._parent_pid = parent_pid

    def __getattr__(self, name):
        raise ForkedWithHipError(
            f"sbayes_amd: this Engine was created in process {self._parent_pid} and inherited through fork() by process "
            f"{os.getpid()}; its device handle is not valid here and was forgotten (sbe.{name} not called).  {FIX}")


def _after_fork_in_child():
    global _FORKED_FROM, _HIP_PID
    parent = _HIP_PID
    for eng in list(_ENGINE_REFS):
        try:
            eng._forget(_ForgottenLib(parent if parent is not None else os.getppid()))
        except Exception:                       # a half-constructed engine: nothing to forget
            pass
    _ENGINE_REFS.clear()
    for fn in _CLEAR_HOOKS:
        try:
            fn()
        except Exception:
            pass
    if parent is not None:
        _FORKED_FROM = parent
    _HIP_PID = None


os.register_at_fork(after_in_child=_after_fork_in_child)
