#!/usr/bin/env python3
"""One bench JSON line, condensed: python tools/show_bench.py file"""
import json, sys
d = json.loads([ln for ln in open(sys.argv[1]).read().strip().splitlines() if ln.startswith("{")][-1])
r = d["roofline"]
print(f"value {d['value']} {d['unit']}  ms/step {d['ms_per_step']}  n_gpus {d['n_gpus']}  contexts {d.get('contexts')}  streams {d.get('streams')}  ids {d['frames_with_all_ids_correct']}")
print(f"roofline frac {r['frac']} in burst {(r.get('in_burst') or {}).get('frac')} (K1 alone {r['avg_launch_ms']} ms over {r['launches_timed']} launches, in company {r.get('avg_launch_ms_in_company')})  e2e_frac {d.get('e2e_frac')}  stages {d['stage_ms_per_step']}  outliers {d['outlier_regions']['count']}")
for k in ("ab_shared_stream", "cpu_baseline", "parity_in_run", "gathered", "dist", "library"):
    if k in d:
        print(f"{k}: {json.dumps(d[k])[:400]}")
for k, v in d.get("other_workloads", {}).items():
    if isinstance(v, dict):
        print(f"  {k}: value {v.get('value')} pipelined {(v.get('pipelined') or {}).get('value')} parity {(v.get('parity_in_run') or {}).get('summary')} {v.get('skipped', '')}")
        for kk, vv in v.items():
            if isinstance(vv, dict) and "parity_in_run" in vv:
                print(f"      {kk}: parity {vv['parity_in_run']['summary']}")
    else:
        print(f"  {k}: {v}")
