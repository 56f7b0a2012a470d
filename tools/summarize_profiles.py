#!/usr/bin/env python3
"""Summarise gpurun_out/prof_<TAG>/ (written by tools/profile_gpu.sh on the GPU box) into the
tracked profiles/<TAG>/ directory: rocprofv3 kernel stats, per-kernel PMC means, the FETCH_SIZE
calibration and profiles/traffic_latest.json (per-launch HBM bytes of the dominant kernel, read by
bench.py for `roofline.traffic`)."""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
TAG = sys.argv[1] if len(sys.argv) > 1 else "r2"
SRC = REPO / "gpurun_out" / f"prof_{TAG}"
DST = REPO / "profiles" / TAG
DST.mkdir(parents=True, exist_ok=True)


def newest(pattern):
    """The most recently written match (a directory may hold the files of earlier runs of the same pass)."""
    files = sorted(glob.glob(pattern), key=lambda f: Path(f).stat().st_mtime)
    return files[-1:]


def counters(dirname):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in newest(str(SRC / dirname / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            out[name][(r["Counter_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return out


def dominant(rows, prefix):
    """Means for the kernel whose name starts with `prefix`, at its largest grid (the batched launches)."""
    res = {}
    for name, d in rows.items():
        if not name.startswith(prefix):
            continue
        grid = max(g for (_, g) in d)
        for (cname, g), v in d.items():
            if g == grid:
                res[cname] = {"mean": sum(v) / len(v), "dispatches": len(v), "grid_size": g, "kernel": name}
    return res


summary = {}
# ---- FETCH_SIZE calibration ---------------------------------------------------------------------
calib = {}
rows = counters("pmc_fetch_calib")
for name, d in rows.items():
    for (cname, g), v in d.items():
        if cname == "FETCH_SIZE" and name.startswith("read_"):
            kb = sum(v) / len(v)
            calib[name] = {"streamed_bytes": 1 << 30, "FETCH_SIZE_KB": kb, "reported_fraction": kb * 1024 / (1 << 30)}
summary["fetch_calibration"] = calib
corr4 = 1.0 / calib["read_dword"]["reported_fraction"] if "read_dword" in calib else None
corr16 = 1.0 / calib["read_dwordx4"]["reported_fraction"] if "read_dwordx4" in calib else None

traffic = {}
sys.path.insert(0, str(REPO))
from bench import HEADLINE_BATCH      # noqa: E402  (the default bench command's states per launch: what profile_gpu.sh ran)
CASES = [("headline", HEADLINE_BATCH, k) for k in ("packed", "packed_tuple", "packed_tuple_lds", "packed_general", "packed_v2", "onehot", "onehot_general")] + \
        [("stress", b, k) for k in ("packed", "packed_v2", "onehot") for b in (8, 64)]
for wl, batch, kern in CASES:
        suffix = kern if wl == "headline" else f"stress_{kern}_b{batch}"
        prefix = {"packed": "sbe::k_mixture_tuple_mfma" if wl == "headline" else "sbe::k_mixture_rows",
                  "packed_tuple": "sbe::k_mixture_tuple64",
                  "packed_tuple_lds": "sbe::k_mixture_combo",
                  "packed_general": "sbe::k_mixture_rows",
                  "packed_v2": "sbe::k_mixture_v2",
                  "onehot": "sbe::k_mixture_combo" if wl == "headline" else "sbe::k_mixture_onehot_v2",
                  "onehot_general": "sbe::k_mixture_onehot_v2"}[kern]
        entry = {}
        fetch = dominant(counters(f"pmc_fetch_{suffix}"), prefix)
        write = dominant(counters(f"pmc_write_{suffix}"), prefix) if wl == "headline" else {}
        # (k_mixture_tuple_mfma's HBM stream is 4-byte lane loads of the per-slot tables, like the packed forms)
        if "FETCH_SIZE" in fetch:
            raw = fetch["FETCH_SIZE"]["mean"] * 1024
            # packed streams 4-byte lane loads + 16-byte table staging, onehot 16-byte lane loads:
            # apply the calibrated factor of the dominant access width
            factor = (corr16 if kern.startswith("onehot") else corr4) or 2.0     # (gfx950: FETCH_SIZE reports 1/2)
            entry.update(fetch_raw_bytes=raw, fetch_correction=factor, fetch_bytes=raw * factor)
        if "WRITE_SIZE" in write:
            entry["write_bytes"] = write["WRITE_SIZE"]["mean"] * 1024
        if entry:
            entry["hbm_bytes_per_launch"] = entry.get("fetch_bytes", 0.0) + entry.get("write_bytes", 0.0)
            entry["evals_per_launch"] = batch
            traffic[f"{wl}:{kern}:{batch}"] = entry
        sq = {}
        for part in ("sq1", "sq2", "sq3"):
            sq.update({k: v["mean"] for k, v in dominant(counters(f"pmc_{part}_{suffix}"), prefix).items()})
        if sq:
            summary[f"sq_counters_{wl}_{kern}_b{batch}"] = sq
        for f in newest(str(SRC / f"trace_{suffix}" / "*" / "*_kernel_stats.csv")):
            shutil.copy(f, DST / f"kernel_stats_{wl}_{kern}_b{batch}.csv")
        b = SRC / f"bench_{suffix}.json"
        if b.exists() and b.stat().st_size:
            shutil.copy(b, DST / f"bench_line_{wl}_{kern}_b{batch}_under_rocprof.json")
            # identity of what was profiled: the dominant kernel's name as the engine reports it and the digest of the
            # run's results -- bench.py refuses a static figure whose identity differs from the run it is quoted beside
            try:
                line = json.loads([ln for ln in b.read_text().splitlines() if ln.startswith("{")][-1])
                ident = {"kernel": line["roofline"]["kernel"], "results_sha1": line["results_sha1"]}
            except Exception:
                ident = None
            if ident:
                if f"{wl}:{kern}:{batch}" in traffic:
                    traffic[f"{wl}:{kern}:{batch}"].update(ident)
                if sq:
                    summary[f"sq_counters_{wl}_{kern}_b{batch}"].update({"_kernel": ident["kernel"], "_results_sha1": ident["results_sha1"]})
for f in newest(str(SRC / "trace_stress_packed_unsorted_b64" / "*" / "*_kernel_stats.csv")):
    shutil.copy(f, DST / "kernel_stats_stress_packed_unsorted_b64.csv")
# entries of earlier sessions of this round that this session did not re-measure (another batch size, a pass that was
# skipped) are kept: every entry names the kernel and the results digest it belongs to
old = json.loads((DST / "pmc_summary.json").read_text()) if (DST / "pmc_summary.json").exists() else {}
old_traffic = old.pop("traffic", {})
summary = {**old, **{k: v for k, v in summary.items() if v}}
summary["traffic"] = {**old_traffic, **traffic}
(DST / "pmc_summary.json").write_text(json.dumps(summary, indent=1))
latest_path = REPO / "profiles" / "traffic_latest.json"
latest = json.loads(latest_path.read_text()) if latest_path.exists() else {}
latest.update({k: {"bytes": round(v["hbm_bytes_per_launch"]), "kernel": v.get("kernel"), "results_sha1": v.get("results_sha1"), "profile": TAG}
               for k, v in traffic.items()})
latest_path.write_text(json.dumps(latest, indent=1))
print(json.dumps(summary, indent=1)[:6000])
