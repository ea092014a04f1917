#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): bench line, rocprofv3 kernel stats and the PMC passes of the
# same command; everything lands in gpurun_out/refresh/.  Copy into profiles/ afterwards (tools/pmc_aggregate.py for the PMC).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT" "$ROOT/gpurun_out/pmc"; export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 600 python3 bench.py --frames-cache /tmp/c2frames > "$OUT/bench.json.log" 2> "$OUT/bench.err" || exit 1
cat "$OUT/bench.json.log"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$ROOT/bench.py" --frames-cache /tmp/c2frames --no-cpu-baseline > "$OUT/stats.log" 2>&1 || exit 2
tail -1 "$OUT/stats.log"
bash "$ROOT/tools/pmc_k1.sh" > "$OUT/pmc.log" 2>&1 || exit 3
ls "$OUT" "$ROOT/gpurun_out/pmc"
