#!/bin/bash
# On the GPU box: per-kernel times of the reference bench recipe (uniform noise, 1920x1080, batch of 16).
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/noise; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o noise -- python3 "$ROOT/tools/noise_prof.py" > "$OUT/noise.log" 2>&1
tail -2 "$OUT/noise.log"
python3 - "$OUT/noise_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.5:
        print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
PY
