#!/usr/bin/env python3
"""One bench run, condensed: python tools/show_bench.py <name>.json.log   (the compact stdout line; <name>.detail.json beside it --
A3_BENCH_DETAIL, tools/refresh_r06.sh -- is read for everything the line does not carry; a bench_detail.json may be given directly)"""
import json, signal, sys
from pathlib import Path

signal.signal(signal.SIGPIPE, signal.SIG_DFL)   # `| head` is not an error

p = Path(sys.argv[1])
text = p.read_text().strip()
if text.startswith("{") and "\n" in text:       # a bench_detail.json (indented)
    line, d = None, json.loads(text)
else:
    line = json.loads([ln for ln in text.splitlines() if ln.startswith("{")][-1])
    det = p.with_name(p.name.replace(".json.log", "").replace(".log", "") + ".detail.json")
    d = json.loads(det.read_text()) if det.exists() else line
if line is not None:
    print(f"line: {len(json.dumps(line))} bytes, detail {line.get('detail')}")
r = d["roofline"]
print(f"value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  n_gpus {d['n_gpus']}  contexts {d.get('contexts')}  gates {d.get('gates')}  ids {d.get('frames_with_all_ids_correct')}")
print(f"roofline frac {r['frac']} in burst {(r.get('in_burst') or {}).get('frac')} (K1 alone {r['avg_launch_ms']} ms over {r['launches_timed']} launches, in company {r.get('avg_launch_ms_in_company')})  "
      f"e2e_frac {d.get('e2e_frac')}  stages {d.get('stage_ms_per_step')}  outliers {(d.get('outlier_regions') or {}).get('count')}")
for k in ("roofline_warp", "ab_shared_stream", "ab_burst_gates", "ab_r04_library_default", "ab_fps", "cpu_baseline", "parity_in_run", "gathered", "dist", "library"):
    if k in d:
        print(f"{k}: {json.dumps(d[k])[:400]}")
for k, v in (d.get("other_workloads") or {}).items():
    if isinstance(v, dict):
        print(f"  {k}: value {v.get('value')} pipelined {(v.get('pipelined') or {}).get('value')} parity {(v.get('parity_in_run') or {}).get('summary')} {v.get('skipped', '')}")
        for kk, vv in v.items():
            if isinstance(vv, dict) and ("parity_in_run" in vv or "median_ms" in vv):
                print(f"      {kk}: {json.dumps(vv)[:200]}")
    else:
        print(f"  {k}: {v}")
