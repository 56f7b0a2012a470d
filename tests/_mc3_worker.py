"""Worker side of tests/test_gpu_processes.py -- TEST INFRASTRUCTURE.

The shape of an MC3 worker (sbayes/mcmc_setup.py:428-470: `MCMCChainProcess.run` receives a pickled model over a Pipe,
builds its chain, steps it, sends results back): a FRESH process (spawn) unpickles (model, sample), creates its OWN engine
on device 0 on first use, replays the first `n_steps` steps of a recorded reference trace through ResidentChain.step and
sends the scalars back."""
import os

import numpy as np


def replay(model, sample, fx, tr, n_steps):
    from sbayes_amd.resident import ResidentChain
    chain = ResidentChain(model, sample)
    out = []
    prev_clusters, prev_source, prev_weights = fx.groups[0], fx.source, fx.weights
    for i in range(n_steps):
        clusters, source, weights = tr.clusters(i), tr.source(i), tr.weights[i]
        changed_src = np.flatnonzero((source != prev_source).any(axis=(1, 2)))
        ll, group_lh, mix = chain.step(clusters=clusters if not np.array_equal(clusters, prev_clusters) else None,
                                       source_rows=(changed_src, source[changed_src]),
                                       weights=weights if not np.array_equal(weights, prev_weights) else None)
        out.append((float(ll), float(mix), [float(v) for v in group_lh]))
        chain.accept()
        prev_clusters, prev_source, prev_weights = clusters, source, weights
    return out


def worker_main(conn, name, n_steps):
    """Target of a spawn-context Process (the child imports this module by name; nothing is inherited)."""
    try:
        from sbayes_amd import _proc, registry
        from tests._fixtures import load_npz, load_trace
        assert not _proc.hip_touched() and not registry._ENGINES
        model, sample = conn.recv()                       # unpickled here: Likelihood.__setstate__ ran in this process
        lik_ll = float(model.likelihood(sample, caching=False))      # the model's own engine, created lazily HERE
        eng = model.likelihood.engine
        assert eng._pid == os.getpid() and _proc.hip_touched()
        fx, tr = load_npz(name), load_trace(name)
        out = replay(model, sample, fx, tr, n_steps)
        conn.send(("ok", os.getpid(), lik_ll, out))
    except BaseException as exc:                          # noqa: BLE001  (reported to the parent, like MCMCChainProcess.run)
        import traceback
        conn.send(("error", os.getpid(), repr(exc), traceback.format_exc()))
    finally:
        conn.close()
