"""bench.py's rank logic at world_size 8 on CPU (gloo), the device replaced by a stub engine: the launch line of the
driver's scaling run (`python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8`) starts, every rank
sets up its engine on its own device, the barrier / max-over-ranks / whole-job aggregation work, and rank 0 prints
exactly one JSON line with the contract's keys.  No GPU involved."""
import json
import os
import socket
import subprocess
import sys
import textwrap
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent

STUB_RUNNER = textwrap.dedent('''
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %(repo)r)
    import bench
    import sbayes_amd.engine as engine_mod

    class StubEngine:
        """Stands in for sbayes_amd.engine.Engine: remembers its device, returns the oracle-free constant results."""
        def __init__(self, wl, batch, device):
            self.device, self.batch = device, batch
            self.n_timed = 0
        def info(self):
            return {"device": self.device, "device_name": "stub"}
        def sync(self): pass
        def mixture_loglik(self, slot):
            time.sleep(0.0008 if self.device == 5 else 0.0002)
            return -1.0
        def mixture_loglik_batch_async(self, first, n): time.sleep(0.004 if self.device == 5 else 0.001)      # rank 5 is slow
        def fetch_results(self, first, n): return np.full(n, -1.0)
        def profile_mixture(self, first, n, iters): return 0.0, 0.0
        def timer_start(self): pass
        def timer_mark(self): pass
        def timer_elapsed(self): return 6 * 0.05                   # ms: span of the K = 6 launches of a repetition
        def kernel_timing_start(self): self.n_timed = 0
        def kernel_timing_resume(self): self.n_timed += 1
        def kernel_timing_pause(self): pass
        def kernel_timing_stop(self): return self.n_timed, 0.05
        def last_mixture_kernel(self): return "stub"
        def close(self): pass

    bench.setup_engine = lambda wl, batch, device, kernel="packed", log_mode="product", n_slots=None: StubEngine(wl, batch, device)
    bench.verify_results = lambda eng, wl, slots, values, what, want0=None, tol=1e-10: (0.0, [int(s) for s in slots])
    engine_mod.device_count = lambda: 8                       # "an 8-GPU node"
    import sbayes_amd.chains as chains
    _orig = chains.init_process_group
    chains.init_process_group = lambda backend=None: _orig("gloo")
    bench.oracle_eval = lambda wl: (type("O", (), {"mixture_loglik": staticmethod(lambda *a: -1.0)}), ())
    sys.argv = ["bench.py", "--gpus", "8", "--steps", "6", "--warmup", "2", "--workload", "cfg1", "--batch", "4"]
    bench.main()
''')


def test_bench_rank_logic_world_size_8(tmp_path):
    runner = tmp_path / "runner.py"
    runner.write_text(STUB_RUNNER % {"repo": str(REPO)})
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(runner)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=str(tmp_path))
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout                     # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["warmup"] == 2 and line["scaling"] == "weak"
    assert line["config"]["evals_per_step"] == 4
    assert line["n_reps"] >= 5 and line["n_reps"] % 2 == 1 and line["dist_backend"] == "gloo"
    # whole-job aggregate: 8 ranks x 6 steps x 4 evals over the (median repetition's) max-over-ranks time
    assert abs(line["value"] - 8 * 6 * 4 / (line["ms_per_step"] * 6 / 1e3)) <= 1e-3 * line["value"]
    assert line["cpu_baseline"] is None and "per_config" not in line      # N > 1: no CPU leg, no secondary figures
    for key in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert key in line
    # per-chain figures of the N > 1 line: every rank's own rate, kernel time and single-chain rate;
    # the deliberately slow rank (device 5) is visible, and the whole-job value is bounded by it
    pr = line["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and sorted(r["device"] for r in pr) == list(range(8))
    slow = next(r for r in pr if r["device"] == 5)
    others = [r for r in pr if r["device"] != 5]
    assert all(slow["evals_per_s"] < 0.6 * r["evals_per_s"] for r in others)
    assert all(slow["single_chain_evals_per_s"] < 0.6 * r["single_chain_evals_per_s"] for r in others)
    assert line["per_rank_evals_per_s_min"] == slow["evals_per_s"] and line["per_chain_evals_per_s_min"] == slow["single_chain_evals_per_s"]
    assert line["per_rank_evals_per_s_max"] == max(r["evals_per_s"] for r in pr)
    assert all(r["kernel_avg_us"] == 50.0 for r in pr)
    assert line["value"] <= 8 * slow["evals_per_s"] * 1.05                # whole-job aggregate over the max-over-ranks time
    rf = line["roofline"]
    for key in ("frac", "frac_contract", "unique_bytes_per_launch", "frac_traffic", "bound"):
        assert key in rf
    # the kernel time is the span of a repetition's K launches / K and fits inside the step
    assert rf["kernel_avg_us"] == 50.0 and rf["kernel_avg_us"] <= line["ms_per_step"] * 1e3
    assert line["parity_timed_kernel_max_rel_err"] == 0.0 and 0 in line["parity_timed_kernel"]["slots"]
