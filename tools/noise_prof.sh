#!/bin/bash
# On the GPU box: per-kernel times of the reference bench recipe (c0: uniform noise, 1920x1080, batch of 32) and of BASELINE config 4 (c4).
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd /tmp
for WL in ${1:-c0 c4}; do
  OUT=$ROOT/gpurun_out/noise_$WL; mkdir -p "$OUT"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o noise -- python3 "$ROOT/tools/noise_prof.py" $WL > "$OUT/noise.log" 2>&1
  grep "^$WL" "$OUT/noise.log"
  python3 - "$OUT/noise_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.3:
        print(f"{r['Name'][:60]:62s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} {float(r['Percentage']):5.1f}%")
PY
done
