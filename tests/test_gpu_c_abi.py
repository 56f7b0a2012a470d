"""The C ABI used from plain C (no Python / NumPy / torch in the caller): tests/c/abi_smoke.c is
compiled with gcc against include/sbe_engine.h, dlopens the engine library, evaluates the cfg1
fixture and must reproduce the reference's golden values."""
import re
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

from sbayes_amd import _lib
from tests._fixtures import load_npz

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_plain_c_caller(tmp_path):
    exe = tmp_path / "abi_smoke"
    subprocess.run(["gcc", "-O2", "-std=c99", "-Wall", "-I", str(REPO / "include"), str(REPO / "tests" / "c" / "abi_smoke.c"),
                    "-o", str(exe), "-ldl"], check=True)
    fx = load_npz("cfg1")
    n, f, s = fx.features.shape
    case = tmp_path / "case.bin"
    with open(case, "wb") as fh:
        fh.write(np.array([n, f, s, fx.n_comp], dtype=np.int32).tobytes())
        fh.write(np.array([g.shape[0] for g in fx.groups], dtype=np.int32).tobytes())
        fh.write(np.ascontiguousarray(fx.features).view(np.uint8).tobytes())
        for c in range(fx.n_comp):
            fh.write(np.ascontiguousarray(fx.groups[c]).view(np.uint8).tobytes())
            conc = fx.conc[c] if fx.conc[c].ndim == 3 else np.broadcast_to(fx.conc[c], (fx.groups[c].shape[0], f, s))
            fh.write(np.ascontiguousarray(conc, dtype=np.float64).tobytes())
        fh.write(np.ascontiguousarray(fx.source).view(np.uint8).tobytes())
        fh.write(np.ascontiguousarray(fx.weights, dtype=np.float32).tobytes())
    out = subprocess.run([str(exe), str(_lib.lib_path()), str(case)], check=True, capture_output=True, text=True, timeout=120).stdout
    vals = dict(re.findall(r"^(\w+) (.+)$", out, flags=re.M))
    assert abs(float(vals["mixture_ll"]) - fx.meta["mixture_ll"]) <= 1e-10 * abs(fx.meta["mixture_ll"])
    assert abs(float(vals["collapsed_ll"]) - fx.meta["collapsed_ll"]) <= 1e-6 * abs(fx.meta["collapsed_ll"])
    assert int(vals["n_na"]) == int(fx.na_values.sum())
    assert "slot 7 out of range" in vals["error_text"]
    # ABI 6: an overlapping matrix is taken (last group = id, slot marked); the count-deriving call refuses the marked slot by name
    # through the plain-C boundary; the host helpers answer like NumPy
    assert int(vals["overlap_set_rc"]) == 0
    assert int(vals["overlap_rc"]) == 4 and re.search(r"object \d+ is in groups 0 and 1 of component 0", vals["overlap_text"])
    cl = fx.groups[0][:, :5]
    want_gids = np.where(cl.any(axis=0), cl.argmax(axis=0), -1)
    assert [int(v) for v in vals["host_gids"].split()] == want_gids.tolist()
    src0 = fx.source[:5, 0, :]
    assert [int(v) for v in vals["host_sids"].split()] == np.where(src0.any(-1), src0.argmax(-1), 255).tolist()
    moved = np.where(want_gids < 0, -1, (want_gids + 1) % fx.groups[0].shape[0])
    assert int(vals["host_touched"]) == len(set(want_gids[want_gids >= 0]) | set(moved[moved >= 0]))
    # second session: the one-call Gibbs proposal through the plain-C boundary = the Python wrapper's answer for the same input
    from sbayes_amd.engine import Engine
    with Engine(fx.features, [g.shape[0] for g in fx.groups], n_slots=2) as eng:
        for c in range(fx.n_comp):
            eng.set_concentration(c, fx.conc[c])
            eng.set_groups(0, c, fx.groups[c])
        eng.set_source(0, fx.source)
        eng.recount(0)
        eng.update_probs(0, range(fx.n_comp))
        eng.set_weights(0, fx.weights)
        per_group, per_object = eng.collapsed_and_source_prior(0)
        assert float(vals["fused_collapsed_ll"]) == sum(per_group.tolist())       # (left to right, like the C loop)
        assert abs(float(vals["fused_collapsed_ll"]) - fx.meta["collapsed_ll"]) <= 1e-6 * abs(fx.meta["collapsed_ll"])
        assert float(vals["fused_source_prior"]) == sum(per_object.tolist())
        if eng.gibbs_propose_supported():
            objs = np.arange(min(n, 5), dtype=np.int32)
            ids, sel, _back, touched, rows = eng.gibbs_propose(0, 1, objs, np.full((objs.size, f), 0.5))
            assert [int(v) for v in vals["propose_ids"].split()] == ids.reshape(-1)[:24].tolist()
            assert int(vals["propose_touched"]) == touched.size and float(vals["propose_row_sum"]) == float(rows.sum()) == 0.0
            assert abs(float(vals["propose_sel0"]) - float(sel[0, 0])) <= 1e-7
