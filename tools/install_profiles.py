#!/usr/bin/env python3
"""Copy the artefacts of tools/refresh_profiles.sh (gpurun_out/refresh, gpurun_out/pmc) into profiles/ under this round's
tag: bench line, rocprofv3 kernel stats of the same command, the PMC summary (tools/pmc_aggregate.py), the read
microbenchmark and the 2-rank rehearsal line.  The bench line's roofline.traffic is patched from the PMC summary of the same
run (bench.py read the previously committed summary when it ran).  Usage: python tools/install_profiles.py [tag]"""
import csv, json, pathlib, shutil, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
prof, new = ROOT / "profiles", ROOT / "gpurun_out" / "refresh"
shutil.copy(new / "stats_kernel_stats.csv", prof / f"{tag}_bench_c2_kernel_stats.csv")
subprocess.check_call([sys.executable, str(ROOT / "tools" / "pmc_aggregate.py"), str(ROOT / "gpurun_out" / "pmc"), str(prof / f"{tag}_pmc_bench_c2.json")])
d = json.loads((new / "bench.json.log").read_text().strip().splitlines()[-1])
pmc = json.loads((prof / f"{tag}_pmc_bench_c2.json").read_text())
k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
d["roofline"]["traffic"] = int((2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
(prof / f"{tag}_bench_c2.json.log").write_text(json.dumps(d) + "\n")
for name in ("readbench.txt", "scatterbench.txt", "ab_overlap.txt", "pmc_issue.txt"):
    if (new / name).exists():
        shutil.copy(new / name, prof / f"{tag}_{name}")
for line in (new / "rehearsal_n2_gloo.log").read_text().splitlines():
    if line.startswith("{"):
        (prof / f"{tag}_rehearsal_n2_gloo.json.log").write_text(line + "\n")
print(d["value"], d["ms_per_step"], d["roofline"], d["stage_ms_per_step"], d.get("cpu_baseline"))
for r in csv.DictReader(open(prof / f"{tag}_bench_c2_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.35:
        print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
