#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X sBayes likelihood engine.

Metric (BASELINE.json): log-likelihood evals/sec at 1000 sites x 200 feats x 10 states,
1/2/4/8-GPU chains.  One "eval" = one uncached mixture log-likelihood of one sample state
(SURVEY.md 8(d)):  LL = sum_{n,f not NA} log sum_c w[n,f,c] * p_c[g_c(n), f, x(n,f)].

A "step" = one pass of the hot path over one batch: `--batch` B distinct resident sample
states (independent chains / candidate states of the sampler, sbayes/sampling/mcmc.py:239-241)
evaluated by ONE launch of the fused kernel (default B = 4096: 256 blocks of 16 states, one per CU, each walking the whole
(feature, state) axis; measured 2048 / 4096 / 8192 / 16384 -> 46.8 / 54.2 / 41.0 / 41.8 M evals/s -- up to 4096 states the
per-state tables, 221 MB, stay in the 256 MB Infinity Cache between launches, beyond that they stream from HBM; `batch_sweep`
in the output line).  Everything (feature block, group ids, probability tables, weights) is
resident in HBM before the timed region; the B scalars are fetched to the host inside it.
`single_chain` in the output line is what ONE chain sees (B = 1, host-synchronous), `per_config`
covers every 1-GPU BASELINE config (cfg1, south_america, headline, stress) at B = 1 / 8 / 64.

Kernel time: ONE HIP event pair on the engine's own stream around the K back-to-back launches of every timed repetition
(sbe_timer_start before the first launch, sbe_timer_mark behind the last one, read after the results are in); span / K is the
dominant kernel's duration INCLUDING the dispatch gap between consecutive kernels, and it lies inside the host-timed region,
so kernel_avg_us <= ms_per_step * 1000 by construction (asserted).  rocprofv3's kernel trace of the same command
(profiles/) reads the kernel alone, a few us less.

Roofline: `roofline.frac` = UNIQUE bytes per launch (the shared feature block once + every state's tables, ids and result)
/ kernel time / 8 TB/s -- a fraction that cannot exceed 1.  `frac_contract` is SURVEY.md 8(d)'s per-eval figure x evals per
launch (it counts the shared block once per eval and exceeds 1 for a batched kernel: no roofline meaning, kept for the
contract).  `frac_traffic` and `roofline_valu` (what actually bounds the headline kernel) are STATIC figures from the
committed rocprofv3 PMC passes (profiles/), labelled with their source file and withheld when this build's kernel differs.

Parity: the results of the TIMED launches are checked, outside the timed region: slot 0 (the workload's own state) and 15
random slots against oracle values computed from state READ BACK from the device (group ids, source rows -> counts recounted
by the oracle and compared bit for bit with the device's, weights); 1e-10 relative or the run fails.

Two further legs beside the headline figure (rank 0, N = 1): `hbm_regime` -- 2 x B resident states per launch, whose tables
(442 MB at the default) no longer fit the 256 MB Infinity Cache and stream from HBM on every launch -- and `changing_tables`
-- every step first moves one object of EVERY state to another cluster (count delta, the two touched clusters' tables rebuilt:
sbayes/sampling/counts.py:55-95, conditionals.py:171-188) and then evaluates all of them: no launch re-reads frozen tables.

  python bench.py                       # 1 GPU, defaults finish in about a minute
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
from types import SimpleNamespace

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec peak
N_SIMDS = 1024             # 256 CUs x 4 SIMD-32
HEADLINE_BATCH = 4096      # resident states per launch (tools/summarize_profiles.py keys the PMC passes on it)
NOMINAL_CLOCK_GHZ = 2.4    # MI355X_MICROARCH.md "Max clock"
I8_MFMA_PEAK_TOPS = 5000.0 # MI355X_MICROARCH.md: dense i8 MFMA = 2 x the bf16 rate (~2.5 PF): ~5 POP/s
FP4_MFMA_PEAK_TOPS = 10000.0   # MI355X_MICROARCH.md: dense FP4 (v_mfma_f32_32x32x64_f8f6f4) = 4 x the bf16 rate: ~10 PF
PMC_FILE = "profiles/r6/pmc_summary.json"
TRAFFIC_FILE = "profiles/traffic_latest.json"


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="headline", choices=["cfg1", "south_america", "headline", "stress"])
    ap.add_argument("--batch", type=int, default=None,
                    help="resident sample states (chains x candidate states) evaluated per step "
                         "(default: 4096 for cfg1 / south_america / headline, 64 for stress)")
    ap.add_argument("--kernel", default="packed", choices=["packed", "packed_general", "packed_v2", "packed_tuple", "packed_tuple_lds", "packed_tuple_mfma", "onehot", "onehot_general"],
                    help="packed: state-index stream, group-tuple form when it applies (default); "
                         "packed_tuple: group-tuple form on the vector pipe forced (k_mixture_tuple64: what `packed` ran before round 5); "
                         "packed_tuple_mfma: its matrix-pipe form forced (k_mixture_tuple_mfma: what `packed` picks for >= 320 states); "
                         "packed_general: never the group-tuple form (k_mixture_rows); packed_v2: the older general "
                         "kernel k_mixture_v2; onehot: stream the one-hot block")
    ap.add_argument("--log-mode", default="product", choices=["product", "per_obs"])
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--event-stride", type=int, default=0,
                    help="secondary figure kernel_event_pairs_us: HIP event pair around the dominant kernel of every n-th launch "
                         "of the timed loop (1: every launch; 0, default: none in the timed loop -- an identical loop right after "
                         "it instead).  roofline.kernel_avg_us never comes from these pairs: it is the span of the K back-to-back "
                         "launches of a timed repetition / K")
    ap.add_argument("--no-legs", action="store_true", help="skip the hbm_regime and changing_tables legs (and their extra slots)")
    ap.add_argument("--legs", action="store_true", help="run the two legs even with --no-secondary (tests)")
    ap.add_argument("--min-time", type=float, default=0.05,
                    help="the K-step timed loop is repeated (each repetition bracketed by barrier + sync on both sides) "
                         "until the repetitions together cover this many seconds; the MEDIAN repetition is reported")
    ap.add_argument("--reps", type=int, default=0, help="force the number of repetitions of the K-step loop (0: from --min-time)")
    ap.add_argument("--explore", action="store_true", help="also print the batch sweep to stderr")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary figures (latency, sweep, per_config, a1/a7 rates)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------
# workloads
# ------------------------------------------------------------------------------------------------------------
def load_workload(name):
    """The synthetic BASELINE workloads (seeded generator) or the south_america fixture (the reference's own loader
    output, recorded by tests/golden/make_golden.py: real CSV data + real Dirichlet prior tables)."""
    from sbayes_amd.synthetic import make_workload
    if name != "south_america":
        return make_workload(name)
    z = np.load(REPO / "tests" / "golden" / "south_america.npz", allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    C = len(meta["groups"])
    wl = SimpleNamespace(name=name, features=z["features"], states_per_feature=z["states_per_feature"],
                         component_names=meta["component_names"], groups=[z[f"groups_{i}"] for i in range(C)],
                         concentration=[z[f"conc_{i}"] for i in range(C)], weights=z["weights"], source=z["source"])
    wl.na_values = ~wl.features.any(axis=-1)
    wl.shape = wl.features.shape
    wl.n_components = C
    wl.clusters = wl.groups[0]
    return wl


def setup_engine(wl, batch, device, kernel="packed", log_mode="product", n_slots=None):
    """Engine with `batch` distinct resident states.  Slot 0 is the workload's own state (the parity gate checks it
    against the oracle); the others get random clusters and weights from the host (a few KB each) and their source
    assignment drawn ON THE DEVICE from its prior given those (sbe_sample_source, Philox stream) -- the recipe of
    sbayes_amd.synthetic.make_state without N*F*C host work per state, so a few thousand states are ready in a second or two
    and eight ranks do not spend minutes of start-up in eight contending Python processes."""
    from sbayes_amd.engine import (LOG_PER_OBS, LOG_PRODUCT, MIXTURE_ONEHOT, MIXTURE_ONEHOT_GENERAL, MIXTURE_PACKED,
                                   MIXTURE_PACKED_GENERAL, MIXTURE_PACKED_TUPLE, MIXTURE_PACKED_TUPLE_LDS, MIXTURE_PACKED_TUPLE_MFMA,
                                   MIXTURE_PACKED_V2, Engine)
    n_obj, n_feat, _ = wl.shape
    C = wl.n_components
    K = wl.clusters.shape[0]
    eng = Engine(wl.features, [g.shape[0] for g in wl.groups], n_slots=max(batch, n_slots or 0), device=device)
    eng.set_option(kernel={"onehot": MIXTURE_ONEHOT, "onehot_general": MIXTURE_ONEHOT_GENERAL, "packed": MIXTURE_PACKED,
                           "packed_general": MIXTURE_PACKED_GENERAL, "packed_v2": MIXTURE_PACKED_V2,
                           "packed_tuple": MIXTURE_PACKED_TUPLE, "packed_tuple_lds": MIXTURE_PACKED_TUPLE_LDS,
                           "packed_tuple_mfma": MIXTURE_PACKED_TUPLE_MFMA}[kernel],
                   log_mode=LOG_PRODUCT if log_mode == "product" else LOG_PER_OBS)
    for c in range(C):
        eng.set_concentration(c, wl.concentration[c])
    eng.load_state(0, wl.groups, wl.weights, source=wl.source)      # counts on the device (a9)
    for c in range(C):
        eng.update_probs(0, c)                                      # tables on the device (a4)
    eng.set_option(deferred_checks=True)
    eng.set_rng(2024, 0)
    all_objects = np.arange(n_obj, dtype=np.int32)
    rng = np.random.default_rng(1000)
    for b in range(1, batch):
        eng.copy_slot(b, 0)
        a = rng.integers(0, 2 * K, size=n_obj)
        eng.set_group_ids(b, 0, np.where(a < K, a, -1).astype(np.int32))
        eng.set_weights(b, rng.dirichlet(np.ones(C), size=n_feat).astype(np.float32))
        eng.sample_source(b, b, all_objects, None, from_prior=True)
        eng.recount(b)
        for c in range(C):
            eng.update_probs(b, c)
    eng.set_option(deferred_checks=False)
    return eng


def oracle_eval(wl):
    from oracle import sbayes_oracle as orc
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    args = (wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
    return orc, args


def cpu_baseline(wl, seconds):
    """The CPU oracle (NumPy restatement of the reference path, validated against the reference's
    golden vectors) timed single-threaded on this host.  Checker/baseline only."""
    orc, args = oracle_eval(wl)
    ll = orc.mixture_loglik(*args)          # warm-up call
    n, t0 = 0, time.perf_counter()
    while True:
        orc.mixture_loglik(*args)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 100000:
            break
    return ll, n / el, n, el


def oracle_from_device(eng, wl, slot):
    """Oracle value of the mixture log-likelihood of the state a slot HOLDS: group ids, source rows and weights are read back
    from the device; the oracle recounts (sbayes/sampling/counts.py:10-32) and the counts must equal the device's bit for bit
    (they feed the tables the timed kernel read); then the NumPy restatement of SURVEY.md 8(d)'s expression.  Checker only."""
    from oracle import sbayes_oracle as orc
    n_obj = wl.shape[0]
    groups = []
    for c, g in enumerate(wl.groups):
        ids = eng.get_group_ids(slot, c)
        groups.append(ids[None, :] == np.arange(g.shape[0], dtype=np.int32)[:, None])
    source = eng.get_source_rows(slot, np.arange(n_obj, dtype=np.int32)).astype(bool)
    weights = eng.get_weights(slot)
    counts = orc.recalculate_feature_counts(wl.features, groups, source)
    for c in range(wl.n_components):
        dev = eng.get_counts(slot, c)
        if not np.array_equal(np.asarray(counts[c], dtype=np.float32), dev):
            raise RuntimeError(f"slot {slot}: counts of component {c} on the device differ from the oracle's recount of the device's own source")
    return float(orc.mixture_loglik(wl.features, wl.na_values, groups, counts, wl.concentration, weights))


def verify_results(eng, wl, slots, values, what, want0=None, tol=1e-10):
    """values[i] = what the device returned for slots[i]; -> (max relative error, the slots).  Raises beyond `tol`."""
    worst = 0.0
    for slot, got in zip(slots, values):
        want = oracle_from_device(eng, wl, int(slot))
        if int(slot) == 0 and want0 is not None:
            # slot 0 holds the workload's own state: the oracle value from the host arrays must be the same number
            assert abs(want - want0) <= 1e-13 * abs(want0), (want, want0)
        err = abs(float(got) - want) / abs(want)
        worst = max(worst, err)
        if not err <= tol:
            raise RuntimeError(f"parity gate failed ({what}): slot {int(slot)} device {float(got)!r} oracle {want!r} rel.err {err:.3e} > {tol}")
    return worst, [int(s) for s in slots]


def _median_rep(times):
    """Index of the median repetition (upper median for an even count): value, ms_per_step and the kernel span all come
    from this ONE repetition."""
    order = np.argsort(times)
    return int(order[len(times) // 2])


def timed_loop(eng, launch, fetch, steps, min_time=0.05, max_reps=200):
    """Secondary legs: `steps` launches + the result fetch between stream syncs, repeated to >= min_time; the median
    repetition's wall time per step and event span per launch (both in seconds)."""
    times, spans = [], []
    total = 0.0
    while len(times) < 3 or (total < min_time and len(times) < max_reps):
        eng.sync()
        t0 = time.perf_counter()
        eng.timer_start()
        for _ in range(steps):
            launch()
        eng.timer_mark()
        res = fetch()
        dt = time.perf_counter() - t0
        spans.append(eng.timer_elapsed() * 1e-3)
        times.append(dt)
        total += dt
    k = _median_rep(times)
    return times[k] / steps, spans[k] / steps, len(times), res


def hbm_regime_leg(eng, wl, n_states, steps, b_eval, unique_fn):
    """The headline launch over n_states = 2 x B resident states: their tables no longer stay in the Infinity Cache between
    launches, every launch streams them from HBM."""
    per_step, span, n_reps, res = timed_loop(eng, lambda: eng.mixture_loglik_batch_async(0, n_states),
                                             lambda: eng.fetch_results(0, n_states), steps)
    assert np.all(np.isfinite(res))
    rng = np.random.default_rng(77)
    slots = np.unique(np.concatenate([[0, n_states - 1], rng.integers(0, n_states, size=6)]))
    worst, checked = verify_results(eng, wl, slots, res[slots], "hbm_regime")
    unique = unique_fn(n_states)
    assert span <= per_step * 1.0005, (span, per_step)
    return {"evals_per_launch": n_states, "evals_per_s": round(n_states / per_step, 1), "ms_per_step": round(per_step * 1e3, 5),
            "kernel_avg_us": round(span * 1e6, 3), "kernel": eng.last_mixture_kernel(),
            "unique_bytes_per_launch": int(unique), "frac": round(unique / span / 1e9 / HBM_PEAK_GBS, 5),
            "frac_contract": round(b_eval * n_states / span / 1e9 / HBM_PEAK_GBS, 5),
            "per_state_bytes": int(unique_fn(2) - unique_fn(1)),
            "steps": steps, "n_reps": n_reps, "parity_max_rel_err": worst, "parity_slots": checked,
            "note": "frac = unique bytes per launch / kernel time / 8 TB/s (the same definition as roofline.frac); the states' tables "
                    f"({(unique_fn(2) - unique_fn(1)) * n_states / 1e6:.0f} MB) exceed the 256 MB Infinity Cache, so no launch finds them cached"}


def changing_tables_leg(eng, wl, n_chains, sweeps, warm=3):
    """Every step CHANGES every state before evaluating it: chain i (slots i and n_chains + i: current and candidate, swapped
    after every step like an accepted proposal) moves one object to another cluster -- the candidate is patched on the device,
    its counts follow by the delta rule (sbayes/sampling/counts.py:55-95), the touched clusters' probability tables are
    rebuilt (conditionals.py:171-188, util.py:990-1007), then ONE launch of the fused kernel evaluates the n_chains candidates
    (plus their collapsed per-group likelihoods: Likelihood.__call__).  One engine call per sweep (sbe_step_batch_delta; the
    per-chain payload is packed by the library's host pool)."""
    n_obj = wl.shape[0]
    K = wl.clusters.shape[0]
    cur = np.arange(n_chains, dtype=np.int32)
    cand = cur + n_chains
    ids = np.stack([eng.get_group_ids(int(s), 0) for s in cur]).astype(np.int64)       # host mirror [n_chains][N], -1 = none
    rng = np.random.default_rng(4242)
    mptr = np.arange(n_chains + 1, dtype=np.int32)
    rows = np.arange(n_chains)
    plan = []
    for _ in range(warm + sweeps):
        obj = rng.integers(0, n_obj, size=n_chains)
        new = (ids[rows, obj] + 1 + 1 + rng.integers(0, K, size=n_chains)) % (K + 1) - 1     # always another cluster (or none)
        ids[rows, obj] = new
        plan.append((obj.astype(np.int32), new.astype(np.int32)))
    state = {"cur": cur, "cand": cand, "k": 0, "mix": None}

    def sweep():
        obj, new = plan[state["k"]]
        state["k"] += 1
        _glh, mix, _changed = eng.step_batch_delta(state["cur"], state["cand"], mptr, obj, new)
        state["mix"] = mix
        state["cur"], state["cand"] = state["cand"], state["cur"]       # every proposal accepted: the candidate is the state now

    for _ in range(warm):
        sweep()
    eng.sync()
    eng.kernel_timing_start()                      # (one event pair per sweep around the fused kernel inside the step)
    t0 = time.perf_counter()
    for _ in range(sweeps):
        sweep()
    eng.sync()
    per_sweep = (time.perf_counter() - t0) / sweeps
    n_pairs, mix_ms = eng.kernel_timing_stop()
    kernel = eng.last_mixture_kernel()
    # the last sweep's results against the oracle, from the state the device holds now; and the host mirror of the moves
    rng2 = np.random.default_rng(78)
    chains = np.unique(np.concatenate([[0, n_chains - 1], rng2.integers(0, n_chains, size=4)]))
    slots = state["cur"][chains]
    worst, checked = verify_results(eng, wl, slots, state["mix"][chains], "changing_tables")
    for ch, slot in zip(chains, slots):
        assert np.array_equal(eng.get_group_ids(int(slot), 0), ids[ch]), f"chain {ch}: device ids differ from the applied moves"
    return {"states": n_chains, "evals_per_s": round(n_chains / per_sweep, 1), "ms_per_step": round(per_sweep * 1e3, 4),
            "us_per_state_step": round(per_sweep / n_chains * 1e6, 4), "steps": sweeps, "kernel": kernel,
            "fused_kernel_us_inside_the_step": round(mix_ms * 1e3, 2) if n_pairs else None,
            "fused_kernel_evals_per_s_on_changed_tables": round(n_chains / (mix_ms * 1e-3), 1) if n_pairs and mix_ms > 0 else None,
            "parity_max_rel_err": worst, "parity_slots": checked,
            "note": "one object of EVERY state moved to another cluster before every evaluation (device-side patch, count delta, "
                    "table rebuild, collapsed per-group likelihoods, fused mixture kernel over all candidates; one engine call and "
                    "one synchronisation per step, host wall clock incl. packing the per-chain deltas).  evals_per_s is the WHOLE "
                    "step: the per-chain candidate kernels (patch, count delta, tables, collapsed likelihood) and the host packing of "
                    "n deltas bound it, not the fused kernel -- fused_kernel_us_inside_the_step is that kernel alone (HIP event pair) "
                    "on tables rewritten a moment earlier"}


def cpu_baseline_compiled(wl, seconds):
    """The COMPILED single-thread CPU baseline: oracle/sbayes_oracle_c.c (plain C, gcc -O3; == the NumPy oracle at 1e-12,
    tests/test_oracle_c_cpu.py) evaluating the same expression.  The real reference compiles two functions of this path with
    numba when numba is installed (not in this image) and leaves the rest to NumPy: the whole eval in C is an upper bound on that
    path's single-core rate.  Checker / baseline only."""
    from oracle import sbayes_oracle as orc
    from oracle import sbayes_oracle_c as orc_c
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    args = (wl.features, wl.na_values, wl.groups, counts, wl.concentration, wl.weights)
    ll = orc_c.mixture_loglik(*args)
    n, t0 = 0, time.perf_counter()
    while True:
        orc_c.mixture_loglik(*args)
        n += 1
        el = time.perf_counter() - t0
        if el >= seconds or n >= 100000:
            break
    return ll, n / el, n, el


NUMBA_CAVEAT = ("the real reference JIT-compiles compute_component_likelihood and dirichlet_categorical_logpdf with numba when "
                "numba is installed; the baseline north_star names is the NumPy path, and that is what is timed here "
                "(BASELINE.md section 3)")


def cpu_call_surface_legs(wl, seconds):
    """BASELINE.md section 3 figures (2) and (3), and a3, from the NumPy oracle ON THIS HOST in this run: the literal
    compute_component_likelihood with all groups changed, likelihood_per_component(caching=False), and
    Likelihood.__call__(caching=False) = recalculate_feature_counts + collapsed log-likelihood.  Checker/baseline only."""
    from oracle import sbayes_oracle as orc
    counts = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
    probs0 = orc.component_probs(counts[0], wl.concentration[0])
    n_obj, n_feat, _ = wl.shape
    buf = np.empty((n_obj, n_feat, wl.n_components))
    all_groups = np.arange(wl.groups[0].shape[0])
    each = max(0.5, seconds / 3.0)
    a1 = _rate(lambda: orc.compute_component_likelihood(wl.features, probs0, wl.groups[0], all_groups, buf[..., 0]), each, 3)
    a3 = _rate(lambda: orc.likelihood_per_component(wl.features, wl.na_values, wl.groups, counts, wl.concentration), each, 3)

    def collapsed():
        cnt = orc.recalculate_feature_counts(wl.features, wl.groups, wl.source)
        return orc.collapsed_loglik(cnt, wl.concentration)
    a7 = _rate(collapsed, each, 3)
    return {"a1_component_lh_calls_per_s": round(a1, 2), "a3_likelihood_per_component_per_s": round(a3, 2),
            "a7_collapsed_uncached_per_s": round(a7, 2), "cores": 1, "kind": "port",
            "sample": f">= {each:.1f} s of single-thread NumPy oracle calls per figure on this host", "caveat": NUMBA_CAVEAT}


def _cpu_worker(args):
    name, seconds = args
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[var] = "1"
    _ll, _rate, n, el = cpu_baseline(load_workload(name), seconds)
    return n, el


def cpu_baseline_processes(name, seconds, n_proc):
    """`min(8, cores)` independent single-thread processes (the 8-chain configs: one chain per
    process, the reference's own MC3 layout, sbayes/mcmc_setup.py:271-282) timed concurrently on this host."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    with ctx.Pool(n_proc) as pool:
        res = pool.map(_cpu_worker, [(name, seconds)] * n_proc)
    return sum(n / el for n, el in res), sum(n for n, _ in res), max(el for _, el in res)


def _rate(fn, min_time=0.3, min_calls=5):
    fn()
    n, t0 = 0, time.perf_counter()
    while True:
        fn()
        n += 1
        el = time.perf_counter() - t0
        if n >= min_calls and el >= min_time:
            return n / el


def batch_rate(eng, b, reps=100):
    eng.mixture_loglik_batch(0, b)            # (the first launch of a kernel instantiation loads its code object: not timed)
    eng.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        eng.mixture_loglik_batch_async(0, b)
    eng.fetch_results(0, b)
    dt = time.perf_counter() - t1
    _t, kb = eng.profile_mixture(0, b, max(10, reps // 2))
    return reps * b / dt, kb * 1e3


def per_config_block(device, headline_eng, cpu_seconds):
    """Every 1-GPU BASELINE config (cfg1 = configs[0] shape, south_america = configs[1], headline = configs[2],
    stress = configs[4] shape) at B in {1, 8, 64}: evals/s of the async batch loop and the dominant kernel's HIP-event
    duration; the oracle's single-thread rate beside it (bounded sample)."""
    from sbayes_amd.synthetic import algorithmic_bytes, unique_bytes_per_launch
    out = {}
    for name in ("cfg1", "south_america", "headline", "stress"):
        wl = load_workload(name)
        eng = headline_eng if name == "headline" else setup_engine(wl, 64, device)
        try:
            orc, args = oracle_eval(wl)
            want = orc.mixture_loglik(*args)
            got = eng.mixture_loglik(0)
            assert abs(got - want) <= 1e-10 * abs(want), (name, got, want)
            has_comp = np.stack([g.any(axis=0) for g in wl.groups], axis=1)
            n_pat = len(np.unique(has_comp, axis=0))
            b_eval = algorithmic_bytes(*wl.shape, [g.shape[0] for g in wl.groups], n_pat, packed=True)
            entry = {"shape": list(wl.shape), "groups": [int(g.shape[0]) for g in wl.groups],
                     "algorithmic_bytes_per_eval_packed": b_eval, "parity_rel_err": abs(got - want) / abs(want)}
            for b in (1, 8, 64):
                r, k_us = batch_rate(eng, b, reps=60 if name == "stress" else 100)
                entry[f"b{b}"] = {"evals_per_s": round(r), "kernel_us": round(k_us, 2),
                                  "frac_of_hbm_peak": round(b_eval * b / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)}
            entry["single_eval_sync_us"] = round(1e6 / _rate(lambda: eng.mixture_loglik(0), 0.2, 100), 2)
            if cpu_seconds > 0:
                _ll, rate, n, el = cpu_baseline(wl, min(cpu_seconds, 3.0 if name == "stress" else 1.0))
                entry["cpu_oracle_evals_per_s"] = round(rate, 3)
            out[name] = entry
        finally:
            if eng is not headline_eng:
                eng.close()
    return out


def secondary_figures(eng, wl, B, args):
    """Figures SURVEY.md 8(d) asks to report beside the headline: single-eval latency, batch sweep,
    the literal a1 / a3 / a7 call rates and the PCIe-inclusive eval (state re-uploaded per eval).
    Rank 0, N=1 only; bounded to a few seconds."""
    out = {}
    for _ in range(20):
        eng.mixture_loglik(0)
    single_us = 1e6 / _rate(lambda: eng.mixture_loglik(0), 0.2, 200)
    _t, k1 = eng.profile_mixture(0, 1, 100)
    out["single_chain"] = {"evals_per_s": round(1e6 / single_us, 1), "us_per_eval": round(single_us, 2),
                           "kernel_us": round(k1 * 1e3, 3),
                           "note": "one chain, one state per call, host-synchronous (BASELINE configs[2] '1 chain')"}
    sweep = {}
    for b in (1, 8, 64, 256, 512, 1024, 2048, 4096):
        if b > B:
            break
        r, kb = batch_rate(eng, b)
        sweep[str(b)] = {"evals_per_s": round(r), "kernel_us": round(kb, 2)}
        if args.explore:
            log(f"[explore] batch {b:4d}: {r:12.0f} evals/s  main kernel {kb:9.2f} us")
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
        return eng.collapsed_loglik_all(0).sum()
    out["a7_collapsed_uncached_per_s"] = round(_rate(collapsed), 1)
    if not args.no_cpu_baseline:
        legs = cpu_call_surface_legs(wl, min(args.cpu_seconds, 6.0))
        out["cpu_call_surfaces"] = legs
        out["call_surface_speedups_same_host"] = {
            k: round(out[k] / legs[k], 1) for k in ("a1_component_lh_calls_per_s", "a3_likelihood_per_component_per_s",
                                                    "a7_collapsed_uncached_per_s")}
    # PCIe-inclusive eval: groups + counts + weights re-uploaded, tables rebuilt, one scalar back
    counts = [eng.get_counts(0, c) for c in range(wl.n_components)]

    def pcie_eval():
        for c in range(wl.n_components):
            eng.set_groups(0, c, wl.groups[c])
            eng.set_counts(0, c, counts[c])
            eng.update_probs(0, c)
        eng.set_weights(0, wl.weights)
        return eng.mixture_loglik(0)
    out["pcie_inclusive_evals_per_s_immediate_checks"] = round(_rate(pcie_eval), 1)
    # the way the drop-in layer runs its engines (registry.get_engine): data checks of the state-setting calls are
    # reported by the result fetch that follows, so the seven uploads / table kernels of an eval never stall the stream
    eng.set_option(deferred_checks=True)
    out["pcie_inclusive_evals_per_s"] = round(_rate(pcie_eval), 1)
    eng.set_option(deferred_checks=False)
    # cluster-membership marginals of all available objects of one cluster
    available = np.flatnonzero((~wl.clusters.any(axis=0)) | wl.clusters[0])
    table = probs0[0]
    out["f1_cluster_marginals_calls_per_s"] = round(_rate(lambda: eng.cluster_marginals(0, table, available)), 1)
    out["f1_cluster_marginals_objects"] = int(available.size)
    # the floor every host-synchronous engine call pays (an empty kernel with the completion flag; + one word read out of
    # the host-mapped input block and one double stored to the mapped result block; the same with 64 / 1000 blocks of
    # completion tickets) next to the one-launch calls of the drop-in path on resident state: microseconds per call
    lib, h = eng._lib, eng._h
    us = lambda fn: round(1e6 / _rate(fn, min_time=0.1), 2)                                    # noqa: E731
    eng.set_uniform_counts(np.asarray(wl.states_per_feature, dtype=np.float64))
    members = np.flatnonzero(wl.clusters[0])
    few = np.arange(min(12, wl.features.shape[0]), dtype=np.int32)
    out["sync_call_us"] = {
        "floor_empty_kernel": us(lambda: lib.sbe_test_roundtrip(h, 1, 0)),
        "floor_mapped_word_in_double_out": us(lambda: lib.sbe_test_roundtrip(h, 1, 3)),
        "floor_64_blocks": us(lambda: lib.sbe_test_roundtrip(h, 64, 3)),
        "floor_1000_blocks": us(lambda: lib.sbe_test_roundtrip(h, 1000, 3)),
        "source_prior": us(lambda: eng.source_prior(0)),
        "collapsed_loglik_all": us(lambda: eng.collapsed_loglik_all(0)),
        "collapsed_and_source_prior": us(lambda: eng.collapsed_and_source_prior(0)),
        "cluster_posterior_marginals": us(lambda: eng.cluster_posterior_marginals(0, 0, available, 1.0, 1.0)),
        "jump_lh_resident": us(lambda: eng.jump_lh_resident(0, 0, 1 % wl.clusters.shape[0], members, 1.0, 1.0)) if members.size else None,
        "given_unchanged_lh_12_objects": us(lambda: eng.given_unchanged_lh(0, 0, few, 1.0, 1.0)),
        "source_posterior_12_objects": us(lambda: eng.source_posterior(0, few, 1.0, 1.0)),
        "note": "one launch and one completion flag per call (DESIGN.md 7.3); tools/call_floor.py prints the same next to the ctypes-only times",
    }
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

    # batched multi-chain step (sbe_step_batch): 64 chains' deltas (a cluster move + 20 changed source rows each, as in
    # the one-call step above) in ONE engine call per sweep; the deltas are prepared outside the timed loop (proposal
    # logic belongs to the sampler), four different sweeps cycled
    from sbayes_amd.resident import ResidentChainBatch
    n_chains = 64
    batch = ResidentChainBatch(model, [sample] * n_chains, device=eng.device)
    sweeps = []
    for _ in range(4):
        cl = np.broadcast_to(wl.clusters, (n_chains,) + wl.clusters.shape).copy()
        objs_all, ptr = [], [0]
        for i in range(n_chains):
            n = int(rng.integers(0, n_obj))
            cl[i][:, n] = False
            cl[i][int(rng.integers(0, cl.shape[1])), n] = True
            objs = np.unique(np.append(rng.integers(0, n_obj, size=19), n)).astype(np.int32)
            objs_all.append(objs)
            ptr.append(ptr[-1] + objs.size)
        objs_cat = np.concatenate(objs_all)
        sweeps.append((cl, np.array(ptr, dtype=np.int32), objs_cat, np.ascontiguousarray(wl.source[objs_cat])))
    state = {"k": 0}

    def batched_sweep():
        cl, ptr, objs_cat, rows = sweeps[state["k"] % 4]
        state["k"] += 1
        res = batch.step_arrays(cl, None, ptr, objs_cat, rows)
        batch.accept()
        return res
    sweeps_per_s = _rate(batched_sweep, 0.5, 20)
    out["f2_batched_matrix_form_steps_per_s"] = round(sweeps_per_s * n_chains, 1)
    out["f2_batched_matrix_form_sweep_us"] = round(1e6 / sweeps_per_s, 1)
    # the same sweeps in DELTA form (sbe_step_batch_delta, round 3): what an operator produces -- the moved objects with
    # their new cluster, the changed source rows -- goes in as it is; the candidates are patched in O(delta)
    delta_sweeps = []
    for cl, ptr, objs_cat, rows in sweeps:
        mptr, mobj, mcl = [0], [], []
        for i in range(n_chains):
            moved = np.flatnonzero((cl[i] != wl.clusters).any(axis=0)).astype(np.int32)
            mobj.append(moved)
            mcl.append(np.where(cl[i][:, moved].any(axis=0), cl[i][:, moved].argmax(axis=0), -1).astype(np.int32))
            mptr.append(mptr[-1] + moved.size)
        delta_sweeps.append((np.array(mptr, dtype=np.int32), np.concatenate(mobj), np.concatenate(mcl), ptr, objs_cat, rows))

    def batched_delta_sweep():
        mptr, mobj, mcl, ptr, objs_cat, rows = delta_sweeps[state["k"] % 4]
        state["k"] += 1
        res = batch.step_delta(mptr, mobj, mcl, ptr, objs_cat, rows)
        batch.accept()                                      # (like the matrix-form loop: every proposal is accepted)
        return res
    sweeps_per_s = _rate(batched_delta_sweep, 0.5, 20)
    out["f2_batched_steps_per_s"] = round(sweeps_per_s * n_chains, 1)
    out["f2_batched_chains"] = n_chains
    out["f2_batched_sweep_us"] = round(1e6 / sweeps_per_s, 1)
    out["f2_batched_form"] = "delta (sbe_step_batch_delta); matrix form: f2_batched_matrix_form_*"
    batch.close()
    return out


def sampler_replay_block(device, with_cpu):
    """Device cost of the UNCHANGED reference sampler (north_star: "the existing Python MCMC sampler ... drops onto it
    unchanged").  tests/golden/<tag>_calls.npz hold the engine-level call sequence of the REAL sampler -- the reference's
    initialiser, operator schedule and MH loop (sbayes/sampling/mcmc_chain.py:186-238) running on the drop-in layer under
    patch.install(operators=True), recorded in the build container with step markers.  Here the recorded MCMC steps are
    replayed against the real Engine (timing only; tests/test_gpu_call_log.py checks every result) and -- the CPU
    baseline leg -- against the oracle-backed double on this host: wall time per MCMC step, engine calls per step and the
    bytes that crossed the C ABI per step (counted in Engine)."""
    from tools.replay_bench import run
    out = {}
    residual = {}
    try:                                                  # tools/host_residual.py, build container (static, labelled)
        residual = json.loads((REPO / "tests" / "golden" / "host_residual.json").read_text())
    except Exception:
        pass
    survey_steps_per_s = {"cfg1": 417, "south_america": 300, "headline": 26}       # SURVEY.md section 6 (survey container)
    for tag in ("cfg1", "south_america", "headline"):
        try:
            out[tag] = run(tag, cpu=with_cpu, repeats=3, device=device)
        except FileNotFoundError as exc:                 # a fixture that was not shipped
            out[tag] = {"error": str(exc)}
            continue
        r = residual.get("shapes", {}).get(tag)
        if r:
            host_us = r["host_python_us_per_step"]["mean"]
            gpu_us = out[tag]["gpu_us_per_step"]
            out[tag].update(
                host_python_us_per_step=host_us,
                host_layer_us_per_step=r["of_which_this_packages_host_layer_us_per_step"]["mean"],
                end_to_end_steps_per_s_bound=round(1e6 / (host_us + gpu_us), 1),
                limiter="reference's Python" if host_us - r["of_which_this_packages_host_layer_us_per_step"]["mean"] > gpu_us else "engine",
                reference_steps_per_s={"same_container_as_host_python": r["plain_reference_steps_per_s"],
                                       "survey_section_6": survey_steps_per_s[tag]},
                end_to_end_speedup_bound=round(1e6 / (host_us + gpu_us) / r["plain_reference_steps_per_s"], 2))
            # the same run with GibbsSampleSource._propose's body on the device (patch.install(gibbs_source=True)): its own
            # recorded call log where one is shipped, its own host residual
            g = r.get("with_gibbs_source_on_device")
            if g and (REPO / "tests" / "golden" / f"{tag}_gibbs_calls.npz").exists():
                try:
                    gr = run(f"{tag}_gibbs", cpu=False, repeats=3, device=device)
                    g_host = g["host_python_us_per_step"]["mean"]
                    out[tag]["gibbs_source_on_device"] = {
                        "calls_per_step": gr["calls_per_step"], "gpu_us_per_step": gr["gpu_us_per_step"],
                        "h2d_bytes_per_step": gr.get("h2d_bytes_per_step"), "d2h_bytes_per_step": gr.get("d2h_bytes_per_step"),
                        "host_python_us_per_step": g_host,
                        "end_to_end_steps_per_s_bound": round(1e6 / (g_host + gr["gpu_us_per_step"]), 1),
                        "end_to_end_speedup_bound": round(1e6 / (g_host + gr["gpu_us_per_step"]) / r["plain_reference_steps_per_s"], 2)}
                except Exception as exc:             # noqa: BLE001  (a secondary figure never takes the bench line down)
                    out[tag]["gibbs_source_on_device"] = {"error": repr(exc)}
    out["note"] = ("per recorded MCMC step of the real sampler.  gpu_us_per_step: engine-side cost measured in this run (the "
                   "recorded engine calls replayed against the device); cpu_us_per_step: the same call sequence served by the "
                   "NumPy oracle on this host, single thread.  host_python_us_per_step (STATIC, tests/golden/host_residual.json, "
                   "measured in the build container by tools/host_residual.py -- the reference cannot run on the GPU box): wall "
                   "time of the real sampler's MCMC step when every engine call returns a recorded result in O(1) = the "
                   "reference's own Python (proposal logic, RNG, cache bookkeeping, priors) plus host_layer_us_per_step of this "
                   "package's host layer.  end_to_end_steps_per_s_bound = 1e6 / (host_python_us_per_step + gpu_us_per_step): "
                   "what the patched sampler can reach; reference_steps_per_s: the unpatched reference (NumPy path).  "
                   "gibbs_source_on_device: the same with patch.install(gibbs_source=True) -- GibbsSampleSource._propose's body "
                   "(posterior, draw with the reference's own uniforms, count delta, transition probabilities) as engine calls; "
                   "same Markov chain (tests/test_reference_sampler_cpu.py), its own call log and host residual.")
    return out


def static_profile_figures(workload, kernel, B, kern_us, kernel_name=None, results_sha1=None):
    """STATIC figures from the committed rocprofv3 PMC passes (never measured in this run): the HBM traffic per launch
    and the VALU roofline of the dominant kernel.  Each carries the file it was read from AND is self-checking: the
    profile records the dominant kernel's name and the digest of the profiled run's results (tools/summarize_profiles.py);
    a figure whose identity differs from THIS run's (`sbe_last_mixture_kernel`, `results_sha1`) -- a kernel was changed
    and tools/profile_gpu.sh not re-run -- is withheld (None) and the reason returned as `stale`."""
    traffic, valu, stale = None, None, {}

    def identity_ok(what, rec_kernel, rec_sha):
        if rec_kernel is None or rec_sha is None:
            stale[what] = "the committed profile carries no kernel identity (written before round 4): re-run tools/profile_gpu.sh"
            return False
        if kernel_name is not None and rec_kernel != kernel_name:
            stale[what] = f"profiled kernel {rec_kernel!r} != this run's {kernel_name!r}"
            return False
        if results_sha1 is not None and rec_sha != results_sha1:
            stale[what] = f"profiled run's results digest {rec_sha} != this run's {results_sha1}"
            return False
        return True

    tf = REPO / TRAFFIC_FILE
    if tf.exists():
        try:
            tr = json.loads(tf.read_text())
            key = f"{workload}:{kernel}:{B}"
            if key in tr:
                rec = tr[key] if isinstance(tr[key], dict) else {"bytes": tr[key]}
                if identity_ok("traffic", rec.get("kernel"), rec.get("results_sha1")):
                    traffic = {"bytes_per_launch": rec["bytes"],
                               "source": f"{TRAFFIC_FILE} (static: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE passes of an earlier run "
                                         f"of this command, profile {rec.get('profile')}; kernel name and results digest match this run)"}
        except Exception:
            pass
    for cand in (PMC_FILE, "profiles/r5/pmc_summary.json", "profiles/r4/pmc_summary.json", "profiles/r3/pmc_summary.json", "profiles/r2/pmc_summary.json"):
        pf = REPO / cand
        if not pf.exists():
            continue
        try:
            sq = json.loads(pf.read_text()).get(f"sq_counters_{workload}_{kernel}_b{B}")
        except Exception:
            sq = None
        if sq and "SQ_INSTS_VALU" in sq:
            if not identity_ok("roofline_valu", sq.get("_kernel"), sq.get("_results_sha1")):
                break
            # a wave64 VALU instruction occupies its SIMD-32 for 2 passes x 2 cycles = 4 cycles when issued back to
            # back by one wave (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"); peak = every SIMD issuing always
            busy = sq["SQ_INSTS_VALU"] * 4.0
            avail = N_SIMDS * NOMINAL_CLOCK_GHZ * 1e9 * kern_us * 1e-6
            valu = {"bound": "valu", "unit": "SIMD issue cycles per launch", "achieved": round(busy), "peak": round(avail),
                    "frac": round(busy / avail, 4), "valu_instructions_per_launch": round(sq["SQ_INSTS_VALU"]),
                    "clock_ghz": NOMINAL_CLOCK_GHZ, "kernel_us": round(kern_us, 3),
                    "source": f"{cand} (static PMC pass: SQ_INSTS_VALU; kernel name and results digest match this run) x 4 cycles "
                              "/ (1024 SIMDs x nominal clock x the kernel time measured in this run)"}
            break
    return traffic, valu, stale


def roofline_block(b_eval, unique_bytes, B, kern_ms, traffic, valu, kernel_name, packed, kernel_avg_source, shape=None):
    """The `roofline` object of the output line (pure arithmetic: tests/test_bench_roofline_cpu.py).
    Three HBM figures for the dominant kernel, all over the SAME measured kernel time:
      frac           UNIQUE bytes per launch -- the feature block the B states of a launch share counted ONCE, plus every
                     state's tables, ids and result: what the launch must move at least -- / 8 TB/s.  Cannot exceed 1.
                     `achieved` is this figure in GB/s.
      frac_contract  SURVEY.md 8(d)'s per-eval bytes x evals per launch (the shared block counted once PER EVAL): exceeds 1
                     for a batched kernel, so it is no roofline fraction; kept for the contract, never as `frac`.
      frac_traffic   what the memory counters saw (static PMC passes; None when no pass of this build is committed).
    `bound` is the roofline these figures are priced against -- "hbm" (GB/s against 8 TB/s), one of the two the bench contract
    names; `limiter` names the pipe that actually limits the kernel: "valu" when the vector-issue fraction (roofline_valu) exceeds
    both memory fractions, else "hbm"."""
    kern_s = kern_ms * 1e-3
    achieved = unique_bytes / kern_s / 1e9
    frac = achieved / HBM_PEAK_GBS
    contract = b_eval * B / kern_s / 1e9
    frac_traffic = traffic["bytes_per_launch"] / kern_s / 1e9 / HBM_PEAK_GBS if traffic else None
    limiter = "hbm"
    if valu and valu.get("frac") is not None and (frac_traffic is None or valu["frac"] > frac_traffic) and valu["frac"] > frac:
        limiter = "valu"
    out = {
        "bound": "hbm", "limiter": limiter, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(frac, 5),
        "traffic": traffic["bytes_per_launch"] if traffic else None,
        "traffic_source": traffic["source"] if traffic else None,
        "frac_traffic": round(frac_traffic, 5) if frac_traffic is not None else None,
        "unique_bytes_per_launch": int(unique_bytes),
        "frac_contract": round(contract / HBM_PEAK_GBS, 5), "achieved_contract": round(contract, 2),
        "kernel": kernel_name,
        "kernel_avg_us": round(kern_ms * 1e3, 3),
        "kernel_avg_source": kernel_avg_source,
        "algorithmic_bytes_per_eval": b_eval, "evals_per_launch": B,
        "representation": "packed state index (N*F bytes)" if packed else "one-hot (N*F*S bytes)",
        "note": "frac = unique bytes per launch (the shared feature block once + every state's tables, ids, result) / kernel time / "
                "8 TB/s; frac_contract = SURVEY.md 8(d) bytes per eval x evals per launch / the same: it counts the shared block once "
                "per eval and exceeds 1 for a kernel that reads it once per launch -- not a roofline fraction; frac_traffic = "
                "what the memory counters saw.  Up to ~4096 headline states the per-state tables stay in the 256 MB Infinity Cache "
                "between launches (hbm_regime: the same kernel with the tables streaming from HBM) and the kernel is vector-issue "
                "bound: see roofline_valu",
    }
    if "k_mixture_tuple_mfma" in kernel_name and shape is not None:
        # counts on the matrix pipe: rows (16 slots x 2 tuples per 32-row tile) x columns (F*S padded to 32) x objects (padded to 128)
        import re
        m = re.search(r"M tiles (\d+)", kernel_name)
        mt = int(m.group(1)) if m else 1
        m = re.search(r"(\d+) slots x", kernel_name)
        sl = int(m.group(1)) if m else 16                    # slots per block (16 / 4 / 2)
        fp4 = "fp4" in kernel_name
        n_obj, n_feat, n_states = shape
        rows = -(-B // sl) * mt * 32
        cols = -(-(n_feat * n_states) // 32) * 32
        depth = -(-n_obj // 256) * 256 if fp4 else -(-n_obj // 128) * 128      # k-blocks of 64 (FP4) / 32 (i8) objects, four at a time
        ops = 2.0 * rows * cols * depth
        peak = FP4_MFMA_PEAK_TOPS if fp4 else I8_MFMA_PEAK_TOPS
        out["matrix_pipe"] = {("fp4_ops_per_launch" if fp4 else "int8_ops_per_launch"): ops, "achieved_tops": round(ops / kern_s / 1e12, 1),
                              "peak_tops": peak, "frac": round(ops / kern_s / 1e12 / peak, 4),
                              "note": ("v_mfma_f32_32x32x64_f8f6f4 (FP4 x FP4, exact 0/1 operands)" if fp4 else "v_mfma_i32_32x32x32_i8")
                                      + " work of the count contraction (padded tile sizes) over the measured "
                                      "kernel time; the kernel's vector epilogue (one log per table entry), not the matrix pipe, is its limiter"}
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
    from sbayes_amd.synthetic import algorithmic_bytes, unique_bytes_per_launch

    n_dev = device_count()
    device = chains.device_for(local_rank, n_dev)
    wl = load_workload(args.workload)
    n_obj, n_feat, n_states = wl.shape
    B = args.batch if args.batch else (64 if args.workload == "stress" else HEADLINE_BATCH)
    # the two legs beside the headline figure need a second set of B slots (candidates of the changing-tables leg, then the
    # upper half of the 2 B states of the HBM-regime launch): rank 0 of a 1-GPU run only
    legs = rank == 0 and n_gpus == 1 and not args.no_legs and (args.legs or not args.no_secondary) and args.kernel == "packed" and 2 * B <= 16384
    t_setup = time.perf_counter()
    eng = setup_engine(wl, B, device, args.kernel, args.log_mode, n_slots=2 * B if legs else B)
    t_setup = time.perf_counter() - t_setup
    info = eng.info()
    if n_dev >= world > 1:                          # one GPU per rank: every rank must sit on its own device
        devices = chains.gather_chain_values([rank], [info["device"]], world, dist)
        assert len(set(int(d) for d in devices)) == world, f"ranks share devices: {devices}"

    # ---- oracle value of the workload's own state (slot 0); the gate itself runs on the TIMED results, below ------
    want = None
    if rank == 0:
        orc, oargs = oracle_eval(wl)
        want = float(orc.mixture_loglik(*oargs))
        log(f"[bench] {B} states ready in {t_setup:.1f} s; oracle value of slot 0: {want!r}")

    def barrier():
        eng.sync()
        chains.barrier(dist)
        eng.sync()

    def step():
        eng.mixture_loglik_batch_async(0, B)

    for _ in range(args.warmup):
        step()
    eng.fetch_results(0, B)

    stride = args.event_stride
    eng.profile_mixture(0, B, 64)                  # (grows the engine's event pool outside the timed region)
    eng.kernel_timing_start()                      # (events are bracketing nothing until resumed inside the loop)
    eng.kernel_timing_pause()

    own_times, spans = [], []

    def timed_rep():
        """EXACTLY K steps, barrier + stream sync on both sides, max over ranks.  One event pair on the engine's stream
        spans the K launches (first event before launch 1, second behind launch K, read after the fetch)."""
        barrier()
        t0 = time.perf_counter()
        eng.timer_start()
        for i in range(args.steps):
            if stride > 0 and i % stride == 0:
                eng.kernel_timing_resume()         # (secondary figure: event pair around the dominant kernel of this launch)
                step()
                eng.kernel_timing_pause()
            else:
                step()
        eng.timer_mark()
        res = eng.fetch_results(0, B)              # D2H of the B scalars + stream sync, inside the timed region
        own = time.perf_counter() - t0             # this rank's own K steps (before it waits for the others)
        barrier()
        dt = time.perf_counter() - t0
        spans.append(eng.timer_elapsed() * 1e-3)   # seconds; the second event is long done
        own_times.append(own)
        return chains.max_over_ranks(dt, dist), res

    # A K-step loop is ~1 ms at the driver's K = 20: one stray interrupt moves it by percents.  The loop is therefore
    # repeated -- every repetition is the contract's region (exactly K steps, barrier + sync both sides, max over
    # ranks; the repetition count follows from the first one's all-reduced time, so every rank runs the same number)
    # -- and the MEDIAN repetition is reported: value, ms_per_step and the kernel span all come from that ONE repetition.
    first, results = timed_rep()
    n_reps = args.reps if args.reps > 0 else int(min(200, max(5, np.ceil(args.min_time / max(first, 1e-9)))))
    n_reps += 1 - n_reps % 2                           # an odd count: the median is ONE repetition, not the upper of two middle ones
    rep_times = [first]
    for _ in range(n_reps - 1):
        dt, results = timed_rep()
        rep_times.append(dt)
    k_med = _median_rep(rep_times)
    elapsed = float(rep_times[k_med])
    kern_ms = spans[k_med] / args.steps * 1e3      # this rank's span of the median repetition / K
    pair_n, pair_ms = 0, None
    if stride <= 0 and rank == 0:                  # the same launches again, each bracketed by an event pair (secondary figure)
        eng.kernel_timing_start()
        for _ in range(max(args.steps, 20)):
            step()
        eng.fetch_results(0, B)
    pair_n, pair_ms = eng.kernel_timing_stop()
    assert np.all(np.isfinite(results))

    evals = args.steps * B * n_gpus
    value = evals / elapsed
    ms_per_step = elapsed / args.steps * 1e3
    # a kernel cannot take longer than the step that contains it; the span lies inside this rank's own
    # host-timed region, which is at most the max over ranks
    assert kern_ms <= ms_per_step * 1.0005, (kern_ms, ms_per_step)

    # ---- parity gate ON THE TIMED RESULTS (rank 0; outside the timed region): the last repetition's B values -------------------
    parity = parity_slots = None
    if rank == 0:
        rng = np.random.default_rng(2025)
        slots = np.unique(np.concatenate([[0, B - 1], rng.integers(0, B, size=15)])) if B > 1 else np.array([0])
        parity, parity_slots = verify_results(eng, wl, slots, results[slots], "timed launches", want0=want)
        log(f"[bench] parity of the timed kernel ({eng.last_mixture_kernel()}): {len(parity_slots)} slots incl. slot 0 = the workload's state, "
            f"oracle from device readback, max rel.err {parity:.3e}")

    # ---- roofline of the dominant kernel ---------------------------------------------------------------------------
    has_comp = np.stack([g.any(axis=0) for g in wl.groups], axis=1)
    n_pat = len(np.unique(has_comp, axis=0))          # distinct has_components rows (likelihood.py:183)
    packed = not args.kernel.startswith("onehot")
    groups_per_comp = [g.shape[0] for g in wl.groups]
    b_eval = algorithmic_bytes(n_obj, n_feat, n_states, groups_per_comp, n_pat, packed=packed)
    results_digest = __import__("hashlib").sha1(np.ascontiguousarray(results).tobytes()).hexdigest()[:16]
    timed_kernel = eng.last_mixture_kernel()
    traffic, valu, stale = static_profile_figures(args.workload, args.kernel, B, kern_ms * 1e3, timed_kernel, results_digest)

    def unique_fn(n):
        return unique_bytes_per_launch(n_obj, n_feat, n_states, groups_per_comp, n_pat, n, packed=packed)
    roofline = roofline_block(b_eval, unique_fn(B), B, kern_ms, traffic, valu, timed_kernel, packed,
                              f"one HIP event pair on the engine's stream around the {args.steps} back-to-back launches of the median "
                              f"repetition / {args.steps}: the kernel plus the dispatch gap to the next one; it lies inside the "
                              "host-timed region, so kernel_avg_us <= ms_per_step x 1000 (asserted); rocprofv3's kernel trace of the same "
                              "command (profiles/) reads the kernel alone",
                              shape=(n_obj, n_feat, n_states))
    if pair_n:
        roofline["kernel_event_pairs_us"] = round(pair_ms * 1e3, 3)
        roofline["kernel_event_pairs_note"] = (f"secondary: HIP event pairs around {pair_n} single launches "
                                               + ("issued right after the timed loops" if stride <= 0 else f"(every {stride}-th of the timed loops)")
                                               + "; a pair also holds the dispatch gap behind the PREVIOUS kernel")

    # ---- what every rank (= every GPU's chains) saw by itself: north_star asks for PER-CHAIN throughput at each N ------
    # (chains are independent in the reference: sbayes/sampling/mcmc.py:239-241, one OS process per chain in MC3:
    #  sbayes/mcmc_setup.py:271-299).  Each rank reports its own batched rate (its own K steps, not the max over ranks),
    #  its own kernel time and what ONE host-synchronous chain gets on its GPU while the other ranks do the same.
    per_rank = None
    if n_gpus > 1:
        own_rate = args.steps * B / float(np.median(own_times))
        barrier()
        for _ in range(10):
            eng.mixture_loglik(0)
        single_rate = _rate(lambda: eng.mixture_loglik(0), 0.2, 200)
        barrier()
        rows = [chains.gather_chain_values([rank], [v], world, dist)
                for v in (float(info["device"]), own_rate, kern_ms * 1e3, single_rate)]
        per_rank = [{"rank": r, "device": int(rows[0][r]), "evals_per_s": round(float(rows[1][r]), 1),
                     "kernel_avg_us": round(float(rows[2][r]), 3), "single_chain_evals_per_s": round(float(rows[3][r]), 1)}
                    for r in range(world)]

    extra = {}
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        extra.update(secondary_figures(eng, wl, B, args))
        extra["per_config"] = per_config_block(device, eng if args.workload == "headline" and B >= 64 else None,
                                               0.0 if args.no_cpu_baseline else args.cpu_seconds)
        extra["sampler_replay"] = sampler_replay_block(device, not args.no_cpu_baseline)
    if legs:
        # (after every figure that takes slot 0 for the workload's own state: the legs change what the slots hold)
        # (order matters: the changing-tables leg builds a valid, distinct state in each of the upper B slots -- the candidates of
        #  its last sweeps -- which the 2 B-state launch of the HBM-regime leg then evaluates next to the lower B)
        for name, fn in (("changing_tables", lambda: changing_tables_leg(eng, wl, B, sweeps=max(5, min(args.steps, 20)))),
                         ("hbm_regime", lambda: hbm_regime_leg(eng, wl, 2 * B, args.steps, b_eval, unique_fn))):
            try:
                extra[name] = fn()
                log(f"[bench] {name}: {extra[name]['evals_per_s']:.0f} evals/s")
            except RuntimeError as exc:
                if "parity gate failed" in str(exc) or "differ from" in str(exc):
                    raise                                   # wrong numbers take the whole line down
                extra[name] = {"error": repr(exc)}
        if "evals_per_s" in extra.get("hbm_regime", {}):
            extra["value_hbm_regime"] = extra["hbm_regime"]["evals_per_s"]
        if "evals_per_s" in extra.get("changing_tables", {}):
            extra["value_changing_tables"] = extra["changing_tables"]["evals_per_s"]

    cpu = None
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        ll_cpu, cpu_rate, n_cpu, cpu_el = cpu_baseline(wl, args.cpu_seconds)
        cpu = {"value": round(cpu_rate, 3), "unit": "evals/s", "cores": 1, "kind": "port",
               "sample": f"{n_cpu} uncached mixture-LL evals of the {args.workload} workload in {cpu_el:.1f} s, "
                         f"single-thread NumPy oracle (oracle/sbayes_oracle.py), host has {os.cpu_count()} cores",
               "caveat": NUMBA_CAVEAT}
        try:
            ll_c, rate_c, n_c, el_c = cpu_baseline_compiled(wl, min(args.cpu_seconds, 3.0))
            assert abs(ll_c - ll_cpu) <= 1e-10 * abs(ll_cpu), (ll_c, ll_cpu)
            extra["cpu_baseline_compiled"] = {
                "value": round(rate_c, 2), "unit": "evals/s", "cores": 1, "kind": "port",
                "sample": f"{n_c} evals in {el_c:.1f} s by oracle/sbayes_oracle_c.c (plain C, gcc -O3, one thread; the whole eval compiled)",
                "note": "numba is not installed here, so the reference's JIT path cannot be timed; it compiles compute_component_likelihood "
                        "and dirichlet_categorical_logpdf only and leaves the rest of this eval to NumPy: the whole expression in C is an "
                        "UPPER bound on that path's single-core rate",
                "speedup_of_value": round(value / rate_c, 1)}
        except Exception as exc:                                  # noqa: BLE001  (no C compiler on the box: the figure is skipped)
            extra["cpu_baseline_compiled"] = {"error": repr(exc)}
        n_proc = min(8, os.cpu_count() or 1)
        rate8, n8, el8 = cpu_baseline_processes(args.workload, args.cpu_seconds, n_proc)
        extra["cpu_baseline_processes"] = {
            "value": round(rate8, 3), "unit": "evals/s", "cores": n_proc, "kind": "port",
            "sample": f"{n8} evals by {n_proc} independent single-thread processes in {el8:.1f} s "
                      f"(BASELINE.md section 3: min(8, cores) processes for the 8-chain configs)"}

    if rank == 0:
        line = {
            "metric": "log-likelihood evals/sec at 1000 sites x 200 feats x 10 states; 1/2/4/8-GPU chains",
            "value": round(value, 2), "unit": "evals/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "n_reps": n_reps,
            "timed_region": {"what": f"{n_reps} repetitions of the {args.steps}-step loop, each bracketed by barrier + stream sync "
                                     "on both sides and max-reduced over ranks; value / ms_per_step are the MEDIAN repetition",
                             "rep_ms_per_step_min": round(min(rep_times) / args.steps * 1e3, 5),
                             "rep_ms_per_step_median": round(ms_per_step, 5),
                             "rep_ms_per_step_max": round(max(rep_times) / args.steps * 1e3, 5),
                             "timed_s_total": round(float(sum(rep_times)), 4)},
            "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic" if args.workload != "south_america" else "south_america fixture (real data)",
            "config": {"workload": {"headline": f"headline synthetic {n_obj}x{n_feat}x{n_states}, K={wl.clusters.shape[0]}, C={wl.n_components} (BASELINE.json configs[2])",
                                    "south_america": f"experiments/south_america {n_obj}x{n_feat}x{n_states} (BASELINE.json configs[1])",
                                    "stress": f"stress synthetic {n_obj}x{n_feat}x{n_states}, K={wl.clusters.shape[0]}, C={wl.n_components} (BASELINE.json configs[4] shape)",
                                    "cfg1": f"cfg1 synthetic {n_obj}x{n_feat}x{n_states} (BASELINE.json configs[0] shape)"}[args.workload],
                       "evals_per_step": B, "chains_per_gpu": B, "kernel": args.kernel, "log_mode": args.log_mode,
                       "single_chain_evals_per_s": extra.get("single_chain", {}).get("evals_per_s"),
                       "parallelism": f"{n_gpus} independent engine(s), one per GPU, no collectives"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity_timed_kernel_max_rel_err": parity,
            "parity_timed_kernel": {"slots": parity_slots, "tolerance": 1e-10, "kernel": timed_kernel,
                                    "what": "the B results of the last timed repetition; slot 0 = the workload's own state (its oracle value "
                                            "from the host arrays and from the device readback agree), the others random: group ids, source "
                                            "rows and weights read back from the device, counts recounted by the oracle and compared bit for "
                                            "bit with the device's, then oracle/sbayes_oracle.py mixture_loglik"},
            "results_sha1": results_digest,   # (rank 0's B results of the last repetition: A/B runs of engine builds must agree on it)
            "device": info["device_name"],
            "dist_backend": chains.backend_name(dist),
            "setup_s": round(t_setup, 2),
        }
        if per_rank:
            line["per_rank"] = per_rank
            line["per_rank_evals_per_s_min"] = min(r["evals_per_s"] for r in per_rank)
            line["per_rank_evals_per_s_max"] = max(r["evals_per_s"] for r in per_rank)
            line["per_chain_evals_per_s_min"] = min(r["single_chain_evals_per_s"] for r in per_rank)
            line["per_chain_evals_per_s_max"] = max(r["single_chain_evals_per_s"] for r in per_rank)
            line["per_rank_note"] = ("per_rank[i].evals_per_s: rank i's own K-step loops (median repetition, its own clock, B resident states per "
                                     "launch); single_chain_evals_per_s: one host-synchronous chain on rank i's GPU, all ranks measuring at "
                                     "once; per_chain_* = min / max of the latter over the ranks.  `value` stays the whole-job aggregate over "
                                     "the max-over-ranks time.")
        if valu:
            line["roofline_valu"] = valu
        if stale:
            line["static_profile_stale"] = stale          # a static figure was withheld: its profile is not of this build
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
