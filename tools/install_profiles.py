#!/usr/bin/env python3
"""Copy the artefacts of tools/refresh_profiles.sh (gpurun_out/refresh, gpurun_out/pmc) into profiles/ under this round's tag:
bench line, rocprofv3 kernel stats of the isolated and of the overlapped stepping, the PMC summary (tools/pmc_aggregate.py), the
microbenchmarks, the stepping A/B, the spinner probe, the two-rank rehearsals.  The bench line's roofline.traffic is patched from
the PMC summary of the same run (bench.py read the previously committed summary when it ran).  Usage: python tools/install_profiles.py [tag]"""
import csv, json, pathlib, shutil, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
prof, new = ROOT / "profiles", ROOT / "gpurun_out" / "refresh"


def line(path):
    return json.loads([ln for ln in path.read_text().splitlines() if ln.startswith("{")][-1])


for src, dst in (("isolated_kernel_stats.csv", "bench_c2_kernel_stats.csv"), ("overlapped_kernel_stats.csv", "bench_c2_overlapped_kernel_stats.csv")):
    if (new / src).exists():
        shutil.copy(new / src, prof / f"{tag}_{dst}")
if any((ROOT / "gpurun_out" / "pmc").glob("*_counter_collection.csv")):
    subprocess.check_call([sys.executable, str(ROOT / "tools" / "pmc_aggregate.py"), str(ROOT / "gpurun_out" / "pmc"), str(prof / f"{tag}_pmc_bench_c2.json")])
if (new / "bench.json.log").exists():
    d = line(new / "bench.json.log")
    try:
        pmc = json.loads((prof / f"{tag}_pmc_bench_c2.json").read_text())
        k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
        d["roofline"]["traffic"] = int((2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
    except Exception as e:
        print("traffic not patched:", e)
    (prof / f"{tag}_bench_c2.json.log").write_text(json.dumps(d) + "\n")
    print(d["value"], d["ms_per_step"], d["roofline"], d["stage_ms_per_step"], d.get("cpu_baseline", {}).get("value"), d.get("parity_in_run", {}).get("summary"))
for name in ("readbench.txt", "scatterbench.txt", "valubench.txt", "k1_concurrency.txt", "ab_streams.txt", "spin_probe.txt", "pmc_issue.txt", "pmc_decode.txt",
             "rotation.txt", "pmc_chain.txt", "pmc_chain.json", "noise_prof.txt", "noise_c0_kernel_stats.csv", "noise_c4_kernel_stats.csv", "queue_probe_nccl16.txt"):
    if (new / name).exists() and (new / name).stat().st_size > 0:
        shutil.copy(new / name, prof / f"{tag}_{name}")
for src, dst in (("rehearsal_n2_gloo.log", "rehearsal_n2_gloo.json.log"), ("rehearsal_c5_n2_gloo.log", "rehearsal_c5_n2_gloo.json.log"),
                 ("bench_c5.json.log", "bench_c5.json.log"), ("force_dist_nccl_1rank.log", "force_dist_nccl_1rank.json.log"),
                 ("isolated.json.log", "bench_c2_isolated.json.log"), ("overlapped.json.log", "bench_c2_overlapped_profiled.json.log"),
                 ("gather_guard_on.log", "gather_guard_on.json.log"), ("gather_guard_off.log", "gather_guard_off.json.log")):
    if (new / src).exists():
        try:
            (prof / f"{tag}_{dst}").write_text(json.dumps(line(new / src)) + "\n")
        except Exception as e:
            print(src, "skipped:", e)
for which in ("bench_c2_kernel_stats.csv", "bench_c2_overlapped_kernel_stats.csv"):
    p = prof / f"{tag}_{which}"
    if p.exists():
        print("--", which)
        for r in csv.DictReader(open(p)):
            if float(r["Percentage"]) > 0.35:
                print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
