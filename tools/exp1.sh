#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"; OUT=$ROOT/gpurun_out/r4/trace; mkdir -p $OUT
for v in rc2:128 rc2:24 rc2b:24; do
  name=${v%%:*}; fl=${v#*:}
  echo "== $name A3_K1_FLUSH=$fl"
  A3_K1_FLUSH=$fl A3_HIP_LIB=$ROOT/build/$name/libaruco3_hip.so timeout -k 10 300 python3 tools/ab_streams.py 256 40 3 own:2:2:0,own:3:2:0 2>&1 | grep -v amdgpu.ids
done
cd /tmp
A3_K1_FLUSH=24 A3_HIP_LIB=$ROOT/build/rc2b/libaruco3_hip.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o rc2b -- python3 "$ROOT/tools/ab_streams.py" 256 40 2 own:2:2:0 > "$OUT/rc2b.log" 2>&1
python3 "$ROOT/tools/trace_company.py" "$OUT/rc2b_kernel_trace.csv" 0.4
