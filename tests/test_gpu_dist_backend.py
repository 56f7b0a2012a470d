"""The opt-in RCCL branch of chains.init_process_group on hardware.  The job's default process
group is gloo at every rank count (the data path has no collective); SBAYES_AMD_DIST_BACKEND=nccl selects RCCL.  A
one-rank torch.distributed.run of bench.py with a forced process group runs that branch -- set_device, the CUDA-tensor
all_reduce of max_over_ranks, the barrier -- once on the GPU box, as a fresh child process."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_bench_under_a_forced_process_group(backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SBAYES_AMD_DIST_BACKEND=backend, SBAYES_AMD_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(REPO / "bench.py"), "--gpus", "1",
                          "--steps", "3", "--warmup", "1", "--batch", "64", "--no-secondary", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(REPO))
    assert res.returncode == 0, res.stderr[-3000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["parity_timed_kernel_max_rel_err"] <= 1e-10
    assert line["dist_backend"] in (backend, f"gloo (after {backend} failed to initialise)")
