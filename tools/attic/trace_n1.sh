#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/n1; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
python3 "$ROOT/tools/trace_n1.py" $1
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o n1 -- python3 "$ROOT/tools/trace_n1.py" $1 > "$OUT/n1.log" 2>&1; tail -1 "$OUT/n1.log"
python3 - "$OUT/n1_kernel_trace.csv" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
k1 = [i for i, r in enumerate(rows) if "k_grey_threshold7" in r["Kernel_Name"]]
a, b = k1[-3], k1[-2]
t0 = int(rows[a]["Start_Timestamp"]); busy = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  {r['Kernel_Name'][:50]}")
print(f"call-to-call {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us, first kernel to last kernel end {(int(rows[b - 1]['End_Timestamp']) - t0) / 1e3:.1f} us")
PY
