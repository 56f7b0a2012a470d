"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/*.h declare, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as ct
import re
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import _lib

REPO = Path(__file__).resolve().parent.parent


HEADERS = ("sbe_engine.h", "sbe_engine_steps.h", "sbe_engine_diag.h")


def declared_symbols(headers=HEADERS):
    names = set()
    for h in headers:
        text = (REPO / "include" / h).read_text()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(sbe_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), f"{name} declared under include/ but not exported"
    # and the ctypes binding covers exactly the headers
    assert sorted(_lib.PROTOTYPES) == names


def test_production_header_holds_no_test_hooks_and_no_step_family():
    """include/sbe_engine.h is the drop-in boundary -- self-tests, measurement hooks and the one-call
    step family (no caller in the reference) live in their own headers."""
    prod = declared_symbols(("sbe_engine.h",))
    assert not [n for n in prod if n.startswith("sbe_test_") or n.startswith("sbe_timer_") or "profile" in n or "kernel_timing" in n]
    assert not [n for n in prod if re.match(r"sbe_(gibbs_)?step", n)]
    steps = declared_symbols(("sbe_engine_steps.h",))
    assert steps == sorted(["sbe_step", "sbe_step_batch", "sbe_step_delta", "sbe_step_batch_delta", "sbe_gibbs_step"])
    assert "NO CALLER IN THE REFERENCE" in (REPO / "include" / "sbe_engine_steps.h").read_text()


def test_abi_version_and_error_text():
    lib = _lib.load()
    assert lib.sbe_abi_version() == _lib.ABI_VERSION == 6
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


@pytest.mark.parametrize("shape,n_groups,text", [
    ((2, 2, 255), [1], r"n_states=255 unsupported \(1\.\.254; state index is one byte, 0xFF = NA\)"),
    ((2, 2, 2), [1] * 9, r"n_components=9 unsupported \(1\.\.8\)"),
    ((2, 2, 2), [40000, 25535], r"65535 groups in total exceed the 16-bit group index"),
    ((2, 2, 2), [2, -1], r"component 1 has -1 groups"),
    ((2, 2, 2), [0, 0], r"no component has any group"),
])
def test_engine_limits_the_reference_does_not_have_are_reported_by_name(shape, n_groups, text):
    """n_states <= 254, n_components <= 8, G_total <= 65534 (u8 state ids with 0xFF = NA, 8-bit has_components patterns,
    u16 group ids with 0xFFFF = none): fine for every BASELINE config, absent from the reference -- a dataset beyond
    them fails at sbe_create with the limit in the message, before the device is touched (VERDICT r3 missing #5)."""
    from sbayes_amd.engine import Engine, EngineError
    with pytest.raises(EngineError, match=text) as info:
        Engine(np.zeros(shape, dtype=bool), n_groups)
    assert info.value.code == 1


def test_every_array_handed_to_the_library_is_bound_to_a_name():
    """Engine._i / _o pass a bare address (no ctypes helper object keeps the array alive): the argument must be a plain
    local name, never an expression whose temporary dies before the library reads it (engine.py subset_lh)."""
    import ast
    import inspect
    from sbayes_amd import engine
    tree = ast.parse(inspect.getsource(engine))
    bad = []
    for node in ast.walk(tree):
        if (isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr in ("_i", "_o")
                and isinstance(node.func.value, ast.Name) and node.func.value.id == "self"):
            if len(node.args) != 1 or not isinstance(node.args[0], ast.Name):
                bad.append((node.lineno, ast.unparse(node)))
    assert not bad, bad
