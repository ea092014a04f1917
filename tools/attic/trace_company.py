#!/usr/bin/env python3
"""Per-kernel durations in the steady state of a rocprofv3 kernel trace (the last `frac` of its time span): count, mean, median,
and the share of the span each kernel's instances cover.  Usage: python tools/trace_company.py trace.csv [frac]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
cut = t1 - (t1 - t0) * frac
per = {}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < cut: continue
    per.setdefault(r["Kernel_Name"].replace("a3::", "").split("(")[0][:40], []).append((e - s) / 1e3)
span = (t1 - cut) / 1e3
k1 = len(per.get(next((k for k in per if "k_grey_threshold7" in k), ""), [])) or 1
print(f"steady-state span {span / 1e3:.2f} ms, {k1} threshold launches -> {span / k1:.1f} us per step")
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:42s} n {len(v):5d}  mean {statistics.mean(v):8.1f} us  median {statistics.median(v):8.1f}  sum/step {sum(v) / k1:8.1f} us")
