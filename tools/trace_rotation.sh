#!/bin/bash
# On the GPU box: kernel trace of the BURST stepping (--gates burst; the bench default is the free-running rotation since round 5) and where a rotation's time goes (tools/trace_rotation.py)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/${OUTDIR:-r06}/rot; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o rot -- python3 "$ROOT/bench.py" --device-synth --no-cpu-baseline --no-other-workloads --gates burst --repeats 3 --steps 20 --warmup 3 --isolated-launches 2 > "$OUT/rot.log" 2>&1 || { tail -5 "$OUT/rot.log"; exit 1; }
grep '^{' "$OUT/rot.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('profiled', d['value'], d['ms_per_step'])"
python3 "$ROOT/tools/trace_rotation.py" "$(ls $OUT/*kernel_trace.csv | head -1)" 0.3 | tee "$OUT/rotation.txt"
rm -f $OUT/*kernel_trace.csv   # (tens of MB)
