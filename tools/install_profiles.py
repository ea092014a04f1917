#!/usr/bin/env python3
"""Copy the artefacts of tools/refresh_r06.sh (gpurun_out/refresh, gpurun_out/pmc) into profiles/ under the round's tag: every bench run's
compact line (<name>.json.log) and detail (<name>.detail.json), the rocprofv3 kernel stats of the isolated and of the overlapped
stepping, the PMC summary (tools/pmc_aggregate.py), the microbenchmarks, the stepping A/B, the rehearsals.  The headline's
roofline.traffic is patched from the PMC summary of the same refresh (bench.py read the previously committed summary when it ran) and
traffic_source says so.  Usage: python tools/install_profiles.py [tag]"""
import csv, json, pathlib, shutil, subprocess, sys
ROOT = pathlib.Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
prof, new = ROOT / "profiles", ROOT / "gpurun_out" / "refresh"


def line(path):
    return json.loads([ln for ln in path.read_text().splitlines() if ln.startswith("{")][-1])


for src, dst in (("isolated_kernel_stats.csv", "bench_c2_kernel_stats.csv"), ("overlapped_kernel_stats.csv", "bench_c2_overlapped_kernel_stats.csv")):
    if (new / src).exists():
        shutil.copy(new / src, prof / f"{tag}_{dst}")
if any((ROOT / "gpurun_out" / "pmc").glob("*_counter_collection.csv")):
    subprocess.check_call([sys.executable, str(ROOT / "tools" / "pmc_aggregate.py"), str(ROOT / "gpurun_out" / "pmc"), str(prof / f"{tag}_pmc_bench_c2.json")])
traffic = None
try:
    pmc = json.loads((prof / f"{tag}_pmc_bench_c2.json").read_text())
    k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
    traffic = int((2 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
except Exception as e:
    print("no PMC summary of this tag:", e)
for src, dst in (("bench", "bench_c2"), ("isolated", "bench_c2_isolated"), ("overlapped", "bench_c2_overlapped_profiled"), ("rehearsal_n2_gloo", "rehearsal_n2_gloo"),
                 ("rehearsal_n5_gloo", "rehearsal_n5_gloo"), ("rehearsal_c5_n2_gloo", "rehearsal_c5_n2_gloo"), ("bench_c5", "bench_c5"),
                 ("force_dist_nccl_1rank", "force_dist_nccl_1rank"), ("gather_guard_on", "gather_guard_on"), ("gather_guard_off", "gather_guard_off")):
    if not (new / f"{src}.json.log").exists():
        continue
    try:
        ln = line(new / f"{src}.json.log")
        det = json.loads((new / f"{src}.detail.json").read_text()) if (new / f"{src}.detail.json").exists() else None
        if src == "bench" and traffic is not None:
            for d in (ln, det):
                if d is not None:
                    d["roofline"]["traffic"] = traffic
                    d["roofline"]["traffic_source"] = f"profiles/{tag}_pmc_bench_c2.json (PMC passes of the same refresh, tools/pmc_k1.sh; not this run)"
        ln["detail"] = f"profiles/{tag}_{dst}.detail.json"
        (prof / f"{tag}_{dst}.json.log").write_text(json.dumps(ln) + "\n")
        if det is not None:
            (prof / f"{tag}_{dst}.detail.json").write_text(json.dumps(det, indent=1) + "\n")
        if src == "bench":
            print(len(json.dumps(ln)), "bytes:", ln["value"], ln["ms_per_step"], ln["roofline"], ln["stage_ms_per_step"], ln.get("cpu_baseline", {}).get("value"), ln.get("parity_in_run"))
    except Exception as e:
        print(src, "skipped:", e)
for name in ("readbench.txt", "scatterbench.txt", "valubench.txt", "k1_concurrency.txt", "ab_streams.txt", "spin_probe.txt", "pmc_issue.txt", "pmc_decode.txt",
             "rotation.txt", "pmc_chain.txt", "pmc_chain.json", "noise_prof.txt", "noise_c0_kernel_stats.csv", "noise_c4_kernel_stats.csv", "queue_probe_nccl16.txt"):
    if (new / name).exists() and (new / name).stat().st_size > 0:
        shutil.copy(new / name, prof / f"{tag}_{name}")
for which in ("bench_c2_kernel_stats.csv", "bench_c2_overlapped_kernel_stats.csv"):
    p = prof / f"{tag}_{which}"
    if p.exists():
        print("--", which)
        for r in csv.DictReader(open(p)):
            if float(r["Percentage"]) > 0.35:
                print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
