#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"; OUT=$ROOT/gpurun_out/r4/trace; mkdir -p $OUT
timeout -k 5 200 ./tools/micro/valubench
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o own4 -- python3 "$ROOT/tools/ab_streams.py" 256 48 2 own:4:2:0 > "$OUT/own4.log" 2>&1
grep median $OUT/own4.log
python3 "$ROOT/tools/trace_company.py" "$OUT/own4_kernel_trace.csv" 0.4
