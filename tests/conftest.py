import os
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The engine library is a build artefact (git-ignored): (re)build it in-tree when it is missing or older than
    its sources and hipcc is here (hipcc cross-compiles gfx950 without a GPU; the GPU box receives the built file)."""
    import shutil
    sys.path.insert(0, str(REPO))
    import __graft_entry__ as entry
    if entry.stale() and (os.path.exists(entry.HIPCC) or shutil.which("hipcc")):
        entry.build()
    # the host layer's CPython extension (plain C): same rule; without a compiler the ctypes route of _fast.py serves
    if shutil.which(os.environ.get("CC", "gcc")):
        try:
            entry.build_pyhost()
        except Exception as exc:                         # (no Python.h here: the fallback is exercised instead)
            print(f"[conftest] sbayes_amd._sbe_pyhost not built: {exc}", file=sys.stderr)


@pytest.fixture(scope="session")
def engine_lib():
    """The C-ABI shared library (built in-tree by __graft_entry__.build())."""
    from sbayes_amd import _lib
    return _lib.load()
