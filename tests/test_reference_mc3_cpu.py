"""The reference's MC3 flow on the drop-in layer, end to end (build container only).

`sbayes.cli.main` with `mc3.activate: true`, two runs of two chains: the reference forks one `MCMCChainProcess` per chain and
run (mcmc_setup.py:271-282), pickles the patched model to each (`send_initialize_chain`, :299, :554), receives pickled samples
back after every swap interval (:320-324), evaluates `model.likelihood` / `model.prior` on them in the parent (`swap_chains`,
:389-395) and sends them out again.  What must hold for "the sampler drops onto it unchanged": the run completes, twice in
one parent; every object of this package that rides in a sample (`NormalizedWeights` in the cache, the `Likelihood` inside
the model) survives the pipes; and the likelihood the patched chain logged for its final state is the UNPATCHED reference's
likelihood of that state.  (Bit-for-bit equality with an unpatched MC3 run is not available: the reference's workers reseed
`random` at fork, two unpatched runs already differ.)  The device is the oracle-backed double; hardware: test_gpu_processes.py."""
import json
import os
import signal
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent
REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference sBayes not present")


@pytest.mark.parametrize("mode", ["operators", "gibbs_source"])
def test_reference_mc3_runs_on_the_drop_in_layer(tmp_path, mode):
    proc = subprocess.Popen([sys.executable, str(REPO / "tests" / "_mc3_reference_run.py"), str(tmp_path / "work"), mode],
                            stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=str(REPO), start_new_session=True)
    try:
        stdout, stderr = proc.communicate(timeout=900)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)             # the child and every MC3 worker it forked (own session)
        proc.communicate()
        pytest.fail("the MC3 run did not finish in 900 s")
    finally:
        try:
            os.killpg(proc.pid, signal.SIGKILL)         # no worker outlives the test
        except ProcessLookupError:
            pass
    assert proc.returncode == 0, stderr[-4000:]
    out = json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])
    # (the parent's swap_chains evaluates model.likelihood on samples whose per-group caches the workers filled: with the
    #  reference's own cache protocol kept, it mostly answers from the cache -- `parent_calls` lists what did reach an engine)
    assert isinstance(out["parent_calls"], list)
    assert len(out["runs"]) == 2
    for run in out["runs"]:
        assert run["n_logged"] >= 6 and run["i_step"] == 300 and run["swaps_file"] and run["hot_chain_stats"]
        # the worker ran the patched update_weights: its lazily materialised array came back in the sample's cache
        assert run["weights_cache_type"] == "sbayes_amd.likelihood.NormalizedWeights", run
        # the final state's likelihood as the patched chain logged it (8 decimals) == the unpatched reference's
        assert abs(run["logged_likelihood"] - run["recomputed_likelihood"]) <= 2e-6 * abs(run["recomputed_likelihood"]), run
