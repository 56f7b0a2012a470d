#!/usr/bin/env python3
"""Device cost of the UNCHANGED reference sampler: replay the recorded engine-level call logs of the real sampler
(tests/golden/*_calls.npz) against the real Engine and -- as the same-host CPU baseline -- against the oracle-backed
double, and print wall time per recorded MCMC step, calls and ABI bytes per step.  (bench.py's sampler_replay block
runs the same function.)   python tools/replay_bench.py [tag ...]"""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

from tests._call_log import replay_timed          # noqa: E402
from tests._fake_engine import FakeEngine          # noqa: E402
from tests.test_call_log_cpu import features_of    # noqa: E402


def run(tag, path=None, cpu=True, repeats=3, device=0):
    from sbayes_amd.engine import Engine
    path = path or REPO / "tests" / "golden" / f"{tag}_calls.npz"
    base = tag.replace("_before", "").replace("_gibbs", "")
    feats = features_of(base)

    def make(n_groups):                                # the engine as the drop-in layer creates it (registry.get_engine)
        eng = Engine(feats, n_groups, n_slots=4, device=device)
        eng.set_option(deferred_checks=True)
        return eng
    gpu = replay_timed(path, make, repeats=repeats)
    out = {"tag": tag, "steps": gpu["steps"], "calls_per_step": round(gpu["calls"] / max(1, gpu["steps"]), 1),
           "gpu_us_per_step": round(gpu["seconds"] / max(1, gpu["steps"]) * 1e6, 1)}
    if gpu["traffic"]:
        h2d, d2h, n_abi = gpu["traffic"]
        out.update(h2d_bytes_per_step=round(h2d / max(1, gpu["steps"])), d2h_bytes_per_step=round(d2h / max(1, gpu["steps"])),
                   abi_calls_per_step=round(n_abi / max(1, gpu["steps"]), 1))
    if cpu:
        ref = replay_timed(path, lambda n_groups: FakeEngine(feats, n_groups), repeats=1)
        out["cpu_us_per_step"] = round(ref["seconds"] / max(1, ref["steps"]) * 1e6, 1)
        out["speedup"] = round(out["cpu_us_per_step"] / out["gpu_us_per_step"], 2)
    return out


def by_method(tag, drain=False):
    """Where the per-step time goes: wall time inside each Engine method over one replay (setters are asynchronous:
    their cost shows up in the next call that synchronises -- unless `drain`: then the stream is synchronised after
    every call and what the call left queued is reported as its `drain_us_per_call`)."""
    from sbayes_amd.engine import Engine
    feats = features_of(tag)
    acc = {}
    def make(n_groups):
        eng = Engine(feats, n_groups, n_slots=4)
        eng.set_option(deferred_checks=True)
        return eng
    res = replay_timed(REPO / "tests" / "golden" / f"{tag}_calls.npz", make, 1, acc, drain)
    steps = max(1, res["steps"])
    return {k: {"calls_per_step": round(n / steps, 2), "us_per_call": round(t / n * 1e6, 1), "us_per_step": round(t / steps * 1e6, 1),
                **({"drain_us_per_call": round(d / n * 1e6, 1)} if drain else {})}
            for k, (n, t, d) in sorted(acc.items(), key=lambda kv: -kv[1][1])}


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] in ("--by-method", "--by-method-drain"):
        for tag in args[1:] or ["headline"]:
            print(json.dumps({"tag": tag, "by_method": by_method(tag, drain=args[0].endswith("drain"))}), flush=True)
    else:
        for tag in args or ["cfg1", "south_america", "headline"]:
            print(json.dumps(run(tag)), flush=True)
