#!/bin/bash
# Two PMC passes over the bench (GPU box): per kernel, how busy the VALU is and what the waves wait for (tuning aid).
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_issue; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="--device-synth --no-cpu-baseline --no-other-workloads --no-pipeline --isolated-launches 2 --repeats 1 --steps 3 --warmup 1"
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT" -o p1 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p1.log" 2>&1 || { tail -5 "$OUT/p1.log"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT" -o p2 -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p2.log" 2>&1 || { tail -5 "$OUT/p2.log"; exit 2; }
python3 "$ROOT/tools/pmc_aggregate.py" "$OUT" "$OUT/issue.json" > /dev/null
python3 - "$OUT/issue.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{'kernel':30s} {'waves':>7s} {'waveMcyc':>9s} {'busyMcyc':>9s} {'VALU M':>7s} {'valu_act/busy':>13s} {'SALU M':>7s} {'LDS M':>6s} {'wait_any/wave':>13s} {'wait_inst/wave':>14s} {'vmem rd/wr M':>13s} {'bankconf':>8s}")
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    g = lambda n: v.get(n, 0.0)
    wc, bc = max(g("SQ_WAVE_CYCLES"), 1), max(g("SQ_BUSY_CYCLES"), 1)
    print(f"{k.replace('a3::','')[:30]:30s} {g('SQ_WAVES'):7.0f} {wc/1e6:9.1f} {bc/1e6:9.2f} {g('SQ_INSTS_VALU')/1e6:7.1f} {g('SQ_ACTIVE_INST_VALU')/bc:13.2f} {g('SQ_INSTS_SALU')/1e6:7.1f} {g('SQ_INSTS_LDS')/1e6:6.1f} "
          f"{g('SQ_WAIT_ANY')/wc:13.2f} {g('SQ_WAIT_INST_ANY')/wc:14.2f} {g('SQ_INSTS_VMEM_RD')/1e6:6.2f}/{g('SQ_INSTS_VMEM_WR')/1e6:5.2f} {g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_LDS_IDX_ACTIVE'),1):8.2f}")
PY
