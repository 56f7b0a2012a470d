"""Child of tests/test_reference_mc3_cpu.py -- TEST INFRASTRUCTURE (build container only: needs /root/reference).

Runs the reference's OWN command-line entry point (`sbayes.cli.main`, sbayes/cli.py:64-109) with MC3 switched on -- two runs
of two chains, each chain an `MCMCChainProcess` forked by the reference (mcmc_setup.py:271-282), models and samples pickled
through its pipes in both directions (mcmc_setup.py:299, :320-324, :554-560), swaps decided in the parent -- on the drop-in
layer under patch.install(operators=True), the device replaced by the oracle-backed double (tests/_fake_engine.py).  Prints
one JSON line."""
import json
import os
import pickle
import random
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests" / "golden"))

import numpy as np  # noqa: E402
import yaml  # noqa: E402

import _ref_stubs  # noqa: E402

_ref_stubs.install()
import sbayes.model  # noqa: E402,F401  (the reference's import order)


def main(work: Path):
    from sbayes_amd import binding, conditionals, counts, likelihood, patch, registry
    from tests._fake_engine import FakeEngine, make_engine_for_observations, make_get_engine
    if work.exists():
        shutil.rmtree(work)
    shutil.copytree("/root/reference/experiments/south_america", work)
    cfg = yaml.safe_load(open(work / "config.yaml"))
    cfg["mcmc"].update(steps=300, samples=6, runs=2)
    cfg["mcmc"]["warmup"].update(warmup_steps=20, warmup_chains=2)
    cfg["mcmc"]["mc3"].update(activate=True, chains=2, swap_interval=50)
    cfg["results"] = {"path": "results", "log_source": False, "log_likelihood": False}     # (the .h5 logger needs PyTables)
    yaml.safe_dump(cfg, open(work / "config.yaml", "w"))

    engines = {}
    get_engine = make_get_engine(engines)
    for mod in (registry, likelihood, conditionals, counts, binding):
        mod.get_engine = get_engine                      # (inherited by the forked workers, like the patch itself)
    registry.engine_for_features = lambda f: (next((e for e in engines.values() if e.n_features == f), None)
                                              or FakeEngine(np.zeros((1, f, 1), dtype=bool)))
    registry.engine_for_observations = make_engine_for_observations(engines)
    gibbs_source = len(sys.argv) > 2 and sys.argv[2] == "gibbs_source"
    patch.install(operators=True, gibbs_source=gibbs_source)
    import sbayes.sampling.initializers as ref_init
    import sbayes.util as ref_util
    np.random.seed(5)
    random.seed(5)
    ref_util.RNG.bit_generator.state = np.random.default_rng(5).bit_generator.state
    ref_init.RNG.bit_generator.state = np.random.default_rng(6).bit_generator.state
    from sbayes.cli import main as cli_main
    os.chdir(work)
    cli_main(config=work / "config.yaml", experiment_name="mc3", processes=1)
    parent_calls = sorted({c[0] for e in engines.values() for c in e.calls})
    patch.uninstall()

    # ---- what the run left behind, checked with the UNPATCHED reference ----
    from sbayes.experiment_setup import Experiment
    from sbayes.load_data import Data
    from sbayes.model import Model
    experiment = Experiment(config_file=work / "config.yaml", experiment_name="check", log=False)
    data = Data.from_config(experiment.config)
    model = Model(data, experiment.config.model)
    assert type(model.likelihood).__module__ == "sbayes.model.likelihood"
    out = {"parent_calls": parent_calls, "runs": []}
    res = work / "results" / "mc3" / "K3"
    for run in (0, 1):
        with open(res / f"state_K3_{run}.pickle", "rb") as fh:
            sample = pickle.load(fh)
        cached = sample.cache.weights_normalized.value
        stats = [ln.split("\t") for ln in open(res / f"stats_K3_{run}.txt").read().splitlines()]
        col = stats[0].index("likelihood")
        sample.everything_changed()
        out["runs"].append({
            "logged_likelihood": float(stats[-1][col]), "n_logged": len(stats) - 1,
            "recomputed_likelihood": float(model.likelihood(sample, caching=False)),
            "weights_cache_type": f"{type(cached).__module__}.{type(cached).__name__}",
            "i_step": int(sample.i_step),
            "swaps_file": (res / f"mc3_swaps_K3_{run}.txt").exists(),
            "hot_chain_stats": (res / "hot_chains" / f"stats_K3_{run}.chain1.txt").exists(),
        })
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main(Path(sys.argv[1]))
