#!/bin/bash
# On the GPU box: kernel traces of stepping arrangements (tools/ab_streams.py specs), per-kernel durations in company.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r4/trace; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for spec in ${SPECS:-own:2:2:0 own:2:1:0 shared:2:2:2}; do
  tag=$(echo $spec | tr ':' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o $tag -- python3 "$ROOT/tools/ab_streams.py" 256 40 2 $spec > "$OUT/$tag.log" 2>&1 || { tail -5 "$OUT/$tag.log"; exit 1; }
  echo "== $spec"; grep median "$OUT/$tag.log"
  python3 "$ROOT/tools/trace_company.py" "$OUT/${tag}_kernel_trace.csv" 0.4
done
