#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (CSV output) of `bench.py --pmc-safe` into profiles/r02_pmc_bench.json.

usage: pmc_summarize.py <fetch_dir> <write_dir> <out.json> dtype B T layers experts
Each dir is the -d directory of one `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv` pass.  Per kernel name:
launches, mean FETCH_SIZE / WRITE_SIZE (KB) and traffic_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE
tallies a 128-B line as 64 B on gfx950 (MI355X_MICROARCH.md, HBM).  Kernels of the library only (namespace m3::)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    dtype, B, T, L, E = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
    fetch, write = read(fd, "FETCH_SIZE"), read(wd, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if "m3::" not in k:
            continue
        f, w = fetch.get(k, []), write.get(k, [])
        name = k.replace("void ", "").replace("m3::", "").split("(")[0]
        fm = sum(f) / len(f) if f else 0.0
        wm = sum(w) / len(w) if w else 0.0
        kernels[name] = {"launches": max(len(f), len(w)), "fetch_size_kb": round(fm, 2), "write_size_kb": round(wm, 2),
                         "traffic_bytes_per_launch": int((2 * fm + wm) * 1024)}
    json.dump({"workload": [dtype, B, T, L, E], "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) "
               "-- python3 bench.py --pmc-safe", "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
