import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["stage_ms_per_step"], d["outlier_regions"]["count"])
for k,v in d["other_workloads"].items():
    print(k, v.get("value"), v.get("pipelined"), v.get("skipped"))
