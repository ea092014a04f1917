#!/bin/bash
# On the GPU box: kernel trace of the pipelined bench (two contexts, deferred decode) and the timeline of its last steady-state step
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/trace; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o pipe -- python3 "$ROOT/bench.py" --device-synth --no-cpu-baseline --no-other-workloads --repeats 3 --steps 20 --warmup 3 > "$OUT/pipe.log" 2>&1
grep '^{' "$OUT/pipe.log" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('profiled', d['value'], d['ms_per_step'])"
python3 "$ROOT/tools/trace_gaps.py" "$OUT/pipe_kernel_trace.csv"
