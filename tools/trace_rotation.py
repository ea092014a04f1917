#!/usr/bin/env python3
"""One rotation of the burst stepping, from a rocprofv3 kernel trace (csv): for the rotations of the steady state (the last
`frac` of the trace), per kernel name the wall-clock union of its four instances -- first start and last end relative to the
rotation's first threshold kernel, time covered -- i.e. where the 2.3 ms of a rotation go.  A rotation = four consecutive
threshold-kernel launches.  Usage: python tools/trace_rotation.py trace.csv [frac]"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("a3::", "").replace("void ", "").split("(")[0].split("<")[0][:28]) for r in rows))
t0, t1 = ev[0][0], max(e for _, e, _ in ev)
cut = t1 - (t1 - t0) * frac
k1 = [(s, e) for s, e, n in ev if "k_grey_threshold" in n and s >= cut]
rots = [k1[i:i + 4] for i in range(0, len(k1) - 4, 4)]
# align rotations: a rotation starts where the gap to the previous threshold kernel's end is largest among 4 consecutive
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
best = None
for shift in range(4):
    rr = [k1[i:i + 4] for i in range(shift, len(k1) - 4, 4)]
    span = statistics.mean(r[3][1] - r[0][0] for r in rr)
    if best is None or span < best[0]: best = (span, shift, rr)
_, shift, rots = best
per = {}
lens = []
for i in range(len(rots) - 1):
    a, b = rots[i][0][0], rots[i + 1][0][0]
    lens.append((b - a) / 1e3)
    names = {}
    for s, e, n in ev:
        if a <= s < b: names.setdefault(n, []).append((s, e))
    for n, iv in names.items():
        per.setdefault(n, []).append(((min(s for s, _ in iv) - a) / 1e3, (max(e for _, e in iv) - a) / 1e3, union(iv) / 1e3, len(iv)))
print(f"{len(lens)} rotations, median length {statistics.median(lens):.1f} us (= {statistics.median(lens) / 4:.1f} us per step)")
print(f"{'kernel':30s} {'n':>3s} {'first start':>12s} {'last end':>10s} {'covered':>9s}   (medians, us from the rotation's first threshold kernel)")
for n, v in sorted(per.items(), key=lambda kv: statistics.median(x[0] for x in kv[1])):
    print(f"{n:30s} {statistics.median(x[3] for x in v):3.0f} {statistics.median(x[0] for x in v):12.1f} {statistics.median(x[1] for x in v):10.1f} {statistics.median(x[2] for x in v):9.1f}")
