#!/usr/bin/env python3
"""Timeline of the last steady-state step in a rocprofv3 kernel trace (gpurun_out/quick/stats_kernel_trace.csv): start offset,
idle gap before each kernel, duration.  Usage: python tools/trace_gaps.py [trace.csv]"""
import csv, sys
path = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/quick/stats_kernel_trace.csv"
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "k_grey_threshold7" in r["Kernel_Name"]]
a, b = k1[-2], k1[-1]
t0 = int(rows[a]["Start_Timestamp"]); prev = None; gaps = 0.0; busy = 0.0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max((s - prev) / 1e3, 0.0) if prev else 0.0
    gaps += gap
    if r is not rows[b]: busy += (e - s) / 1e3
    print(f"{(s - t0) / 1e3:9.1f} +{gap:6.1f} {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:60]}")
    prev = e
print(f"step {((int(rows[b]['Start_Timestamp']) - t0) / 1e3):.1f} us, busy {busy:.1f} us, idle {gaps:.1f} us")
