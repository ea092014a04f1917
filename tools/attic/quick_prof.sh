#!/bin/bash
# On the GPU box: (optionally the GPU tests, then) the bench under rocprofv3 --kernel-trace --stats; prints the per-kernel table.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/quick; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
if [ "${SKIP_TESTS:-1}" != "1" ]; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > "$OUT/test.log" 2>&1 || { tail -30 "$OUT/test.log"; exit 1; }
  tail -1 "$OUT/test.log"
fi
ARGS="--device-synth --no-cpu-baseline --no-other-workloads --repeats 5 ${BENCH_ARGS:-}"
timeout -k 10 300 python3 bench.py $ARGS > "$OUT/bench.log" 2>&1 || { tail -5 "$OUT/bench.log"; exit 1; }
grep '^{' "$OUT/bench.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['stage_ms_per_step'], d['roofline']['frac'], d['stats'])"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$ROOT/bench.py" $ARGS > "$OUT/stats.log" 2>&1
grep '^{' "$OUT/stats.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('profiled', d['value'], d['stage_ms_per_step'], d['roofline']['frac'])"
python3 - "$OUT/stats_kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f"{r['Name'][:44]:46s} calls {r['Calls']:>4s} avg_us {float(r['AverageNs'])/1e3:8.1f} tot_us {float(r['TotalDurationNs'])/1e3:9.1f} {float(r['Percentage']):5.1f}%")
PY
