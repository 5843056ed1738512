#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (CSV output) of `bench.py --pmc-safe` into profiles/rNN_pmc_bench.json.

usage: pmc_summarize.py <fetch_dir> <write_dir> <out.json> dtype B T layers experts [--bench-line FILE] [--skip-forwards N]
                        [--fetch-factor-file calib.json]

Each dir is the -d directory of one `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv` pass.  Per kernel name:
launches, mean / median / min / max FETCH_SIZE and WRITE_SIZE (KB) and traffic_bytes_per_launch = (k x FETCH_SIZE +
WRITE_SIZE) x 1024, k = 2 by default: FETCH_SIZE tallies a 128-B line as 64 B on gfx950 (MI355X_MICROARCH.md, HBM); the
factor for this library's half-line weight loads is calibrated by tools/ubench/fetch_calib.hip (--fetch-factor-file).

Only launches of the TIMED WORKLOAD are counted: the process also runs the synthetic-router calibration and a first eager
forward (other routing / cold caches); the first --skip-forwards forwards' worth of every kernel's dispatches (in dispatch
order) are dropped.  For the grouped expert FFN the launches are also folded per layer (launch i of a forward = layer
i mod L) and, when --bench-line names the JSON line of the same run, compared layer by layer with the algorithmic bytes of
that layer's touched experts.  Kernels of the library only (namespace m3::)."""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict

import numpy as np


def read(d, counter):
    """kernel name -> [(dispatch id, value)] in dispatch order"""
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append((int(row.get("Dispatch_Id", 0) or 0), float(row["Counter_Value"])))
    return {k: [v for _, v in sorted(vs)] for k, vs in acc.items()}


def stats(v):
    a = np.asarray(v, dtype=np.float64)
    if a.size == 0:
        return {"mean": 0.0, "median": 0.0, "min": 0.0, "max": 0.0}
    return {"mean": round(float(a.mean()), 2), "median": round(float(np.median(a)), 2), "min": round(float(a.min()), 2),
            "max": round(float(a.max()), 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("out")
    ap.add_argument("dtype"); ap.add_argument("B", type=int); ap.add_argument("T", type=int)
    ap.add_argument("layers", type=int); ap.add_argument("experts", type=int)
    ap.add_argument("--bench-line", default=None, help="stdout of the profiled bench.py run (its JSON line)")
    ap.add_argument("--skip-forwards", type=int, default=3)
    ap.add_argument("--fetch-factor", type=float, default=2.0)
    a = ap.parse_args()
    fetch, write = read(a.fetch_dir, "FETCH_SIZE"), read(a.write_dir, "WRITE_SIZE")
    bench = None
    if a.bench_line and os.path.exists(a.bench_line):
        for line in open(a.bench_line):
            if line.startswith("{"):
                bench = json.loads(line)
    per_fwd = {}
    if bench:     # launches per forward of every kernel family, from the run's own stage list
        for fam, v in (bench.get("roofline") or {}).get("families", {}).items():
            per_fwd[fam.split("<")[0]] = v["launches"]
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if "m3::" not in k:
            continue
        name = k.replace("void ", "").replace("m3::", "").split("(")[0]
        f, w = fetch.get(k, []), write.get(k, [])
        n_all = max(len(f), len(w))
        # drop the first forwards (calibration, first eager forward): per kernel VARIANT the share is unknown, so drop the same
        # fraction of its dispatches as skip-forwards is of the run's forwards (estimated from the most frequent family)
        kernels[name] = {"launches_all": n_all, "_f": f, "_w": w}
    # forwards in the run ~ launches of the grouped expert kernel / layers
    n_fwd = None
    for name, v in kernels.items():
        if name.startswith("expert_ffn") or name.startswith("expert_gemm"):
            n_fwd = max(n_fwd or 0, v["launches_all"] // max(a.layers, 1))
    n_fwd = n_fwd or 1
    keep_from = min(max(a.skip_forwards, 0), max(n_fwd - 1, 0)) / float(n_fwd)
    out = {}
    for name, v in kernels.items():
        f, w = v.pop("_f"), v.pop("_w")
        f, w = f[int(len(f) * keep_from):], w[int(len(w) * keep_from):]
        fs, ws = stats(f), stats(w)
        out[name] = {"launches": max(len(f), len(w)), "launches_all": v["launches_all"], "fetch_size_kb": fs["mean"],
                     "write_size_kb": ws["mean"], "fetch_size_kb_stats": fs, "write_size_kb_stats": ws,
                     "traffic_bytes_per_launch": int((a.fetch_factor * fs["mean"] + ws["mean"]) * 1024)}
        if (name.startswith("expert_ffn") or name.startswith("expert_gemm")) and len(f) >= a.layers and len(f) % a.layers == 0:
            fl = np.asarray(f).reshape(-1, a.layers)
            wl = np.asarray(w).reshape(-1, a.layers) if len(w) == len(f) else np.zeros_like(fl)
            per_layer = ((a.fetch_factor * fl + wl) * 1024).mean(0)
            out[name]["traffic_bytes_per_layer"] = [int(x) for x in per_layer]
            out[name]["traffic_spread_over_forwards"] = round(float((fl.std(0) / np.maximum(fl.mean(0), 1e-9)).max()), 4)
            re = (bench or {}).get("roofline_expert") or {}
            if re.get("experts_touched") and len(re["experts_touched"]) == a.layers:
                D, F = 512, 1024
                wsz = {"f32": 4, "bf16": 2, "fp8": 1}[a.dtype]
                extra = (F + D) * 4 * (2 if a.dtype == "fp8" else 1)
                live = (bench.get("forward") or {}).get("live_rows", 0)
                alg = [t * (2 * D * F * wsz + extra) + live * 2 * D * 4 for t in re["experts_touched"]]
                out[name]["alg_bytes_per_layer"] = alg
                out[name]["traffic_over_alg_per_layer"] = [round(p / max(q, 1), 3) for p, q in zip(per_layer, alg)]
                out[name]["traffic_over_alg"] = round(float(per_layer.sum()) / max(float(sum(alg)), 1.0), 4)
    json.dump({"workload": [a.dtype, a.B, a.T, a.layers, a.experts], "forwards_in_run": n_fwd, "forwards_skipped": a.skip_forwards,
               "fetch_factor": a.fetch_factor,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --pmc-safe; "
                         "timed-workload launches only (the first forwards of the process are dropped by dispatch order)",
               "kernels": out}, open(a.out, "w"), indent=1)
    print("wrote", a.out, len(out), "kernels;", n_fwd, "forwards in the run")


if __name__ == "__main__":
    main()
