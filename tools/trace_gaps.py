#!/usr/bin/env python3
"""Per-queue timeline of a rocprofv3 --kernel-trace run of bench.py: how much of a forward's wall time is kernels and how
much is the gap between one kernel's end and the next one's start on the same queue.

  rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --streams N --steps 30 --warmup 5 --no-cpu-baseline
  python tools/trace_gaps.py DIR [--launches-per-forward 275]

Only the steady part is looked at: per queue, the last `--tail` dispatches.
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def short(name):
    n = name.split("(")[0].replace("void ", "").replace("m3::", "")
    return n[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--tail", type=int, default=275 * 20)
    ap.add_argument("--launches-per-forward", type=int, default=275)
    a = ap.parse_args()
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel_trace.csv under " + a.dir
    rows = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                rows[int(r["Queue_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    out = {"queues": []}
    for q, rs in sorted(rows.items()):
        rs.sort()
        if len(rs) < 2 * a.launches_per_forward:
            continue
        rs = rs[-a.tail:]
        dur = [e - s for s, e, _ in rs]
        gap = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        wall = rs[-1][1] - rs[0][0]
        n = len(rs)
        gs = sorted(gap)
        by = defaultdict(lambda: [0, 0, 0])       # name -> [count, dur, gap AFTER it]
        for i, (s, e, nm) in enumerate(rs[:-1]):
            b = by[short(nm)]
            b[0] += 1
            b[1] += e - s
            b[2] += gap[i]
        top = sorted(by.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:8]
        out["queues"].append({
            "queue": q, "dispatches": n, "wall_us_per_launch": wall / n / 1e3,
            "kernel_us_per_launch": sum(dur) / n / 1e3, "gap_us_per_launch": sum(gap) / max(len(gap), 1) / 1e3,
            "gap_p10_p50_p90_us": [gs[len(gs) // 10] / 1e3, gs[len(gs) // 2] / 1e3, gs[len(gs) * 9 // 10] / 1e3],
            "negative_gaps": sum(1 for g in gap if g < 0),
            "per_forward_ms": {"wall": wall / n * a.launches_per_forward / 1e6, "kernels": sum(dur) / n * a.launches_per_forward / 1e6,
                               "gaps": sum(gap) / n * a.launches_per_forward / 1e6},
            "top": [{"kernel": k, "n": v[0], "dur_us": v[1] / v[0] / 1e3, "gap_after_us": v[2] / v[0] / 1e3} for k, v in top],
        })
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
