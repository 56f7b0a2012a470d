#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sBayes likelihood engine.

Metric (BASELINE.json): log-likelihood evals/sec at 1000 sites x 200 feats x 10 states,
1/2/4/8-GPU chains.  One "eval" = one uncached mixture log-likelihood of one sample state
(SURVEY.md 8(d)):  LL = sum_{n,f not NA} log sum_c w[n,f,c] * p_c[g_c(n), f, x(n,f)].

A "step" = one pass of the hot path over one batch: `--batch` B distinct resident sample
states (independent chains / candidate states of the sampler, sbayes/sampling/mcmc.py:239-241)
evaluated by one launch sequence of the fused kernel (default B = 1024: four generations of
workgroups per launch; `batch_sweep` in the output line reports B = 1 .. 1024).  Everything (feature block, group ids,
probability tables, weights) is resident in HBM before the timed region; the B scalars are
fetched to the host inside the timed region.

The roofline's kernel duration comes from HIP event pairs recorded (on the engine's own stream) around the
dominant kernel of K launches identical to the timed ones, issued right after the timed region (`--events-in-loop`
records them inside the timed loop instead; the event records between back-to-back launches then cost the loop
several us per step).

  python bench.py                       # 1 GPU, defaults finish in well under a minute
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: independent chains shard one engine per GPU, no data-path collective ("weak"
scaling); torch.distributed is used only for the barrier and the
max-over-ranks of the elapsed time.  Rank 0 prints ONE JSON line on stdout.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=["cfg1", "headline", "stress"])
    ap.add_argument("--batch", type=int, default=1024,
                    help="resident sample states (chains x candidate states) evaluated per step")
    ap.add_argument("--kernel", default="packed", choices=["packed", "packed_general", "packed_tuple_lds", "onehot", "onehot_general"],
                    help="packed: state-index stream, group-tuple form when it applies (default); "
                         "packed_general: never the group-tuple form; onehot: stream the one-hot block")
    ap.add_argument("--log-mode", default="product", choices=["product", "per_obs"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--events-in-loop", action="store_true",
                    help="record the roofline's HIP event pairs inside the timed loop itself (perturbs it: an event "
                         "record between back-to-back launches costs several us per step) instead of in an identical "
                         "loop of K launches right after it")
    ap.add_argument("--explore", action="store_true", help="also print the batch sweep to stderr")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (latency, sweep, a1/a7 rates)")
    return ap.parse_args()


def setup_engine(wl, batch, device, kernel, log_mode):
    from sbayes_amd.engine import (LOG_PER_OBS, LOG_PRODUCT, MIXTURE_ONEHOT, MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED,
                                   MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_TUPLE_LDS, Engine)
    from sbayes_amd.synthetic import make_state

    eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=batch, device=device)
    eng.set_option(kernel={"onehot": MIXTURE_ONEHOT, "onehot_general": MIXTURE_ONEHOT_GENERAL, "packed": MIXTURE_PACKED,
                           "packed_general": MIXTURE_PACKED_GENERAL, "packed_tuple_lds": MIXTURE_PACKED_TUPLE_LDS}[kernel],
                   log_mode=LOG_PRODUCT if log_mode == "product" else LOG_PER_OBS)
    for c in range(wl.n_components):
        eng.set_concentration(c, wl.concentration[c])
    states = []
    for b in range(batch):
        if b == 0:
            clusters, weights, source = wl.clusters, wl.weights, wl.source
        else:
            clusters, weights, source = make_state(wl.features, wl.groups[1:], wl.clusters.shape[0], seed=1000 + b)
        groups = [clusters] + wl.groups[1:]
        eng.load_state(b, groups, weights, source=source)      # counts on the device (a9)
        for c in range(wl.n_components):
            eng.update_probs(b, c)                             # tables on the device (a4)
        states.append((groups, weights, source))
    return eng, states


def cpu_baseline(wl, seconds):
    """The CPU oracle (NumPy restatement of the reference path, validated against the reference's
    golden vectors) timed single-threaded on this host.  Checker/baseline only."""
    from oracle import sbayes_oracle as orc
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    args = (wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
    ll = orc.mixture_loglik(*args)          # warm-up call
    n, t0 = 0, time.perf_counter()
    while True:
        orc.mixture_loglik(*args)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 100000:
            break
    return ll, n / el, n, el


def _rate(fn, min_time=0.3, min_calls=5):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if n >= min_calls and el >= min_time:
            return n / el


def secondary_figures(eng, wl, B, args):
    """Figures SURVEY.md 8(d) asks to report beside the headline: single-eval latency, batch sweep,
    the literal a1 / a3 / a7 call rates and the PCIe-inclusive eval (state re-uploaded per eval).
    Rank 0, N=1 only; bounded to a few seconds."""
    out = {}
    for _ in range(20):
        eng.mixture_loglik(0)
    out["single_eval_sync_us"] = round(1e6 / _rate(lambda: eng.mixture_loglik(0), 0.2, 200), 2)
    _t, k1 = eng.profile_mixture(0, 1, 100)
    out["single_eval_kernel_us"] = round(k1 * 1e3, 3)
    sweep = {}
    for b in (1, 8, 64, 256, 1024):
        if b > B:
            break
        eng.sync()
        reps = 100
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.mixture_loglik_batch_async(0, b)
        eng.fetch_results(0, b)
        dt = time.perf_counter() - t1
        _t, kb = eng.profile_mixture(0, b, 50)
        sweep[str(b)] = {"evals_per_s": round(reps * b / dt), "kernel_us": round(kb * 1e3, 2)}
        if args.explore:
            log(f"[explore] batch {b:4d}: {reps * b / dt:12.0f} evals/s  main kernel {kb * 1e3:9.2f} us")
    out["batch_sweep"] = sweep
    # literal call surface (results cross PCIe every call: SURVEY.md H3)
    n_obj, n_feat, _ = wl.shape
    probs0 = eng.get_probs(0, 0)
    buf = np.empty((n_obj, n_feat, wl.n_components))
    all_groups = np.arange(wl.groups[0].shape[0])
    out["a1_component_lh_calls_per_s"] = round(_rate(lambda: eng.component_lh(probs0, wl.groups[0], all_groups, buf[..., 0])), 1)
    out["a3_likelihood_per_component_per_s"] = round(_rate(lambda: eng.likelihood_per_component(0, buf)), 1)

    def collapsed():
        eng.recount(0)
        return sum(eng.collapsed_loglik(0, c).sum() for c in range(wl.n_components))
    out["a7_collapsed_uncached_per_s"] = round(_rate(collapsed), 1)
    # PCIe-inclusive eval: groups + counts + weights re-uploaded, tables rebuilt, one scalar back
    counts = [eng.get_counts(0, c) for c in range(wl.n_components)]

    def pcie_eval():
        for c in range(wl.n_components):
            eng.set_groups(0, c, wl.groups[c])
            eng.set_counts(0, c, counts[c])
            eng.update_probs(0, c)
        eng.set_weights(0, wl.weights)
        return eng.mixture_loglik(0)
    out["pcie_inclusive_evals_per_s"] = round(_rate(pcie_eval), 1)
    # cluster-membership marginals of all available objects of one cluster
    available = np.flatnonzero((~wl.clusters.any(axis=0)) | wl.clusters[0])
    table = probs0[0]
    out["f1_cluster_marginals_calls_per_s"] = round(_rate(lambda: eng.cluster_marginals(0, table, available)), 1)
    out["f1_cluster_marginals_objects"] = int(available.size)
    # resident step flow -- delta in, collapsed + mixture log-likelihood out
    from sbayes_amd import model as sbm
    from sbayes_amd.registry import release_all
    from sbayes_amd.resident import ResidentChain
    model, sample = sbm.build(wl.features, wl.states_per_feature, wl.component_names, wl.groups, wl.concentration,
                              wl.weights, wl.source)
    chain = ResidentChain(model, sample)
    rng = np.random.default_rng(5)
    clusters = wl.clusters.copy()

    def resident_step():
        n = int(rng.integers(0, n_obj))
        clusters[:, n] = False
        clusters[int(rng.integers(0, clusters.shape[0])), n] = True
        objs = np.unique(np.append(rng.integers(0, n_obj, size=19), n))
        cand = chain.propose(clusters=clusters, source_rows=(objs, wl.source[objs]))
        ll, mix = cand.collapsed_loglik(), cand.mixture_loglik()
        chain.accept()
        return ll, mix
    out["f2_resident_steps_per_s"] = round(_rate(resident_step), 1)

    def one_call_step():
        n = int(rng.integers(0, n_obj))
        clusters[:, n] = False
        clusters[int(rng.integers(0, clusters.shape[0])), n] = True
        objs = np.unique(np.append(rng.integers(0, n_obj, size=19), n))
        res = chain.step(clusters=clusters, source_rows=(objs, wl.source[objs]))
        chain.accept()
        return res
    out["f2_one_call_steps_per_s"] = round(_rate(one_call_step), 1)

    # SURVEY.md 8(f) rank 3 on the resident state: the Gibbs source operator as one engine call (20 objects'
    # source redrawn on the device with the caller's uniforms, counts / tables / likelihoods of the candidate)
    def one_call_gibbs_step():
        objs = np.unique(rng.integers(0, n_obj, size=20))
        res = chain.gibbs_step(objs)
        chain.accept()
        return res
    out["f3_one_call_gibbs_steps_per_s"] = round(_rate(one_call_gibbs_step), 1)
    release_all()
    return out


def main():
    args = parse()
    from sbayes_amd import chains
    rank, local_rank, world = chains.env_rank()
    if world != args.gpus and world > 1:
        log(f"[bench] WORLD_SIZE={world} differs from --gpus {args.gpus}; using WORLD_SIZE")
    n_gpus = world
    dist = chains.init_process_group()

    from sbayes_amd.engine import device_count
    from sbayes_amd.synthetic import algorithmic_bytes, make_workload

    device = chains.device_for(local_rank, device_count())

    wl = make_workload(args.workload)
    n_obj, n_feat, n_states = wl.shape
    B = args.batch
    eng, states = setup_engine(wl, B, device, args.kernel, args.log_mode)
    info = eng.info()

    # ---- parity gate reported with the timing (rank-local, cheap): slot 0 vs the oracle ------
    parity = None
    if rank == 0:
        from oracle import sbayes_oracle as orc
        counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
        want = orc.mixture_loglik(wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
        got = eng.mixture_loglik(0)
        parity = abs(got - want) / abs(want)
        log(f"[bench] parity slot0: gpu {got!r} oracle {want!r} rel.err {parity:.3e}")
        if parity > 1e-10:
            raise RuntimeError(f"parity gate failed: rel.err {parity:.3e} > 1e-10")

    def barrier():
        eng.sync()
        chains.barrier(dist)
        eng.sync()

    def step():
        eng.mixture_loglik_batch_async(0, B)

    for _ in range(args.warmup):
        step()
    eng.fetch_results(0, B)

    barrier()
    if args.events_in_loop:
        eng.kernel_timing_start()              # one HIP event pair per launch of the dominant kernel, engine stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    results = eng.fetch_results(0, B)          # D2H of the B scalars + stream sync, inside the timed region
    barrier()
    elapsed = chains.max_over_ranks(time.perf_counter() - t0, dist)
    if not args.events_in_loop:                # the same K launches again, each bracketed by an event pair
        eng.kernel_timing_start()
        for _ in range(args.steps):
            step()
        eng.fetch_results(0, B)
    n_timed, kern_ms = eng.kernel_timing_stop()
    assert n_timed == args.steps
    assert np.all(np.isfinite(results))

    evals = args.steps * B * n_gpus
    value = evals / elapsed
    ms_per_step = elapsed / args.steps * 1e3

    # ---- roofline of the dominant kernel, HIP events on the engine's stream ---------------------
    has_comp = np.stack([g.any(axis=0) for g in wl.groups], axis=1)
    n_pat = len(np.unique(has_comp, axis=0))          # distinct has_components rows (likelihood.py:183)
    packed = not args.kernel.startswith("onehot")
    b_eval = algorithmic_bytes(n_obj, n_feat, n_states, [g.shape[0] for g in wl.groups], n_pat, packed=packed)
    achieved = b_eval * B / (kern_ms * 1e-3) / 1e9
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
        "kernel": {"packed": "k_mixture_tuple64 (group-tuple form, 64-feature tiles; k_mixture_combo / k_mixture_v2 when not applicable)",
                   "packed_tuple_lds": "k_mixture_combo (group-tuple form, LDS-metadata variant)",
                   "packed_general": f"k_mixture_v2<{args.log_mode}>",
                   "onehot": "k_mixture_combo<onehot> (group-tuple form; k_mixture_onehot_v2 when not applicable)",
                   "onehot_general": f"k_mixture_onehot_v2<{args.log_mode}>"}[args.kernel],
        "kernel_avg_us": round(kern_ms * 1e3, 3),
        "algorithmic_bytes_per_eval": b_eval, "evals_per_launch": B,
        "representation": "packed state index (N*F bytes)" if packed else "one-hot (N*F*S bytes)",
    }
    traffic_file = REPO / "profiles" / "traffic_latest.json"
    if traffic_file.exists():
        try:
            tr = json.loads(traffic_file.read_text())
            key = f"{args.workload}:{args.kernel}:{B}"
            if key in tr:
                roofline["traffic"] = tr[key]
        except Exception:
            pass

    extra = {}
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        extra = secondary_figures(eng, wl, B, args)

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        ll_cpu, cpu_rate, n_cpu, cpu_el = cpu_baseline(wl, args.cpu_seconds)
        cpu = {"value": round(cpu_rate, 3), "unit": "evals/s", "cores": 1, "kind": "port",
               "sample": f"{n_cpu} uncached mixture-LL evals of the {args.workload} workload in {cpu_el:.1f} s, "
                         f"single-thread NumPy oracle (oracle/sbayes_oracle.py), host has {os.cpu_count()} cores"}

    if rank == 0:
        line = {
            "metric": "log-likelihood evals/sec at 1000 sites x 200 feats x 10 states; 1/2/4/8-GPU chains",
            "value": round(value, 2), "unit": "evals/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{args.workload} synthetic {n_obj}x{n_feat}x{n_states}, K={wl.clusters.shape[0]}, "
                                   f"C={wl.n_components} (BASELINE.json configs[2])" if args.workload == "headline"
                       else f"{args.workload} synthetic {n_obj}x{n_feat}x{n_states}",
                       "evals_per_step": B, "chains_per_gpu": B, "kernel": args.kernel, "log_mode": args.log_mode,
                       "parallelism": f"{n_gpus} independent engine(s), one per GPU, no collectives"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_rel_err": parity,
            "device": info["device_name"],
        }
        if cpu:
            line["speedup_vs_cpu_baseline"] = round(value / cpu["value"], 1)
        line.update(extra)
        print(json.dumps(line), flush=True)

    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
