"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/sbe_engine.h declares, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as ct
import re
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import _lib

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "sbe_engine.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbe_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sbe_engine.h but not exported"
    # and the ctypes binding covers exactly the header
    assert sorted(_lib.PROTOTYPES) == names


def test_abi_version_and_error_text():
    lib = _lib.load()
    assert lib.sbe_abi_version() == _lib.ABI_VERSION == 3
    assert lib.sbe_get_info(None, None) != 0
    assert b"null engine" in lib.sbe_last_error(None)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("SBAYES_AMD_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_lib, "_LIB", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_create_validates_arguments_before_touching_the_device():
    lib = _lib.load()
    h = ct.c_void_p()
    feats = np.zeros((2, 2, 2), dtype=np.uint8)
    ng = np.array([1], dtype=np.int32)
    ngp = ng.ctypes.data_as(ct.POINTER(ct.c_int32))
    assert lib.sbe_create(ct.byref(h), 0, 0, 2, 2, 1, ngp, 1, feats.ctypes.data_as(ct.c_void_p)) == 1
    assert b"empty feature block" in lib.sbe_last_error(None)
    assert lib.sbe_create(ct.byref(h), 0, 2, 2, 300, 1, ngp, 1, feats.ctypes.data_as(ct.c_void_p)) == 1
    assert lib.sbe_create(ct.byref(h), 0, 2, 2, 2, 9, ngp, 1, feats.ctypes.data_as(ct.c_void_p)) == 1
    assert not h


def test_engine_without_gpu_raises():
    from sbayes_amd.engine import Engine, EngineError, device_count
    if device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(EngineError, match="no CPU fallback"):
        Engine(np.zeros((4, 3, 2), dtype=bool), n_groups=[1], n_slots=1)
