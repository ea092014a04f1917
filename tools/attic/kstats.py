"""print the a3:: rows of the kernel-stats csv files tools/refresh_profiles.sh stats wrote (gpurun_out/refresh/)"""
import csv
import sys
from pathlib import Path

root = Path(__file__).resolve().parent.parent / "gpurun_out" / "refresh"
for f in sys.argv[1:] or ("isolated", "overlapped"):
    print("==", f)
    for r in csv.DictReader(open(root / f"{f}_kernel_stats.csv")):
        if "a3::" in r["Name"] and float(r["Percentage"]) > 0.2:
            print(f"{r['Name'][:50]:52s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
