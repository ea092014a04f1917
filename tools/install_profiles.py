#!/usr/bin/env python3
"""Copy the artefacts of tools/refresh_profiles.sh (gpurun_out/refresh, gpurun_out/pmc) into profiles/: the previous set moves to
profiles/history/ with the next version number; the bench line's roofline.traffic is patched from the PMC summary of the
same run (bench.py read the older summary when it ran).  Usage: python tools/install_profiles.py"""
import csv, json, pathlib, re, shutil, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
prof, hist, new = ROOT / "profiles", ROOT / "profiles" / "history", ROOT / "gpurun_out" / "refresh"
hist.mkdir(parents=True, exist_ok=True)
ver = 1 + max([int(m.group(1)) for p in hist.iterdir() if (m := re.search(r"_v(\d+)\.", p.name))] or [0])
for name, dst in (("r01_bench_c2_kernel_stats.csv", f"r01_bench_c2_kernel_stats_v{ver}.csv"), ("r01_bench_c2.json.log", f"r01_bench_c2_v{ver}.json.log"),
                  ("r01_pmc_bench_c2.json", f"r01_pmc_bench_c2_v{ver}.json")):
    if (prof / name).exists():
        shutil.move(str(prof / name), str(hist / dst))
shutil.copy(new / "stats_kernel_stats.csv", prof / "r01_bench_c2_kernel_stats.csv")
subprocess.check_call([sys.executable, str(ROOT / "tools" / "pmc_aggregate.py")])
d = json.loads((new / "bench.json.log").read_text().strip().splitlines()[-1])
pmc = json.loads((prof / "r01_pmc_bench_c2.json").read_text())
k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
d["roofline"]["traffic"] = int((2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
(prof / "r01_bench_c2.json.log").write_text(json.dumps(d) + "\n")
print(d["value"], d["ms_per_step"], d["roofline"], d["stage_ms_per_step"], d["cpu_baseline"])
for r in csv.DictReader(open(prof / "r01_bench_c2_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.35:
        print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
