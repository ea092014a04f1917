#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes written by tools/pmc_k1.sh (gpurun_out/pmc/*_counter_collection.csv) into
profiles/r01_pmc_bench_c2.json: per kernel, the mean of every counter over its dispatches (a3:: kernels only).
bench.py reads the K1 entry for roofline.traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch."""
import csv
import json
import pathlib
import sys
from collections import defaultdict

ROOT = pathlib.Path(__file__).resolve().parent.parent
src = pathlib.Path(sys.argv[1]) if len(sys.argv) > 1 else ROOT / "gpurun_out" / "pmc"
dst = pathlib.Path(sys.argv[2]) if len(sys.argv) > 2 else ROOT / "profiles" / "r02_pmc_bench_c2.json"

acc = defaultdict(lambda: defaultdict(list))
for path in sorted(src.glob("*_counter_collection.csv")):
    per_dispatch = defaultdict(float)           # a counter may be reported per XCD/instance: sum within a dispatch
    names = {}
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            k = row["Kernel_Name"]
            if "a3::" not in k:
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            per_dispatch[key] += float(row["Counter_Value"])
            names[row["Dispatch_Id"]] = k
    for (disp, counter), v in per_dispatch.items():
        acc[names[disp]][counter].append(v)

out = {}
for k in sorted(acc):
    short = k.split("(")[0]
    out[short] = {c: sum(v) / len(v) for c, v in sorted(acc[k].items())}
    out[short]["dispatches"] = max(len(v) for v in acc[k].values())
dst.write_text(json.dumps(out, indent=1) + "\n")
k1 = [k for k in out if "k_grey_threshold7" in k]
for k in k1:
    f, w = out[k].get("FETCH_SIZE", 0), out[k].get("WRITE_SIZE", 0)
    print(f"{k}: FETCH_SIZE {f:.0f} KiB (x2 = {2 * f * 1024 / 1e9:.3f} GB)  WRITE_SIZE {w:.0f} KiB ({w * 1024 / 1e9:.3f} GB)  "
          f"traffic {(2 * f + w) * 1024 / 1e9:.3f} GB/launch")
