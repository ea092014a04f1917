#!/bin/bash
# One PMC pass over the bench for LDS behaviour of every kernel (GPU box): bank conflicts, LDS issue stalls, instruction mix.
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_lds; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"; python3 bench.py --frames-cache /tmp/c2frames --no-cpu-baseline --steps 3 > "$OUT/b.log" 2>&1 || exit 1
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT" -o lds -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --frames-cache /tmp/c2frames --no-cpu-baseline > "$OUT/lds.log" 2>&1
python3 "$ROOT/tools/pmc_aggregate.py" "$OUT" "$OUT/lds.json" > /dev/null
python3 - "$OUT/lds.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
    print(f"{k[:34]:36s} waveMcyc {wc/1e6:8.1f}  valu {v.get('SQ_INSTS_VALU',0)/1e6:7.1f}M lds {v.get('SQ_INSTS_LDS',0)/1e6:6.1f}M  "
          f"bank_conflict/lds_active {v.get('SQ_LDS_BANK_CONFLICT',0)/max(v.get('SQ_LDS_IDX_ACTIVE',1),1):5.2f}  wait_lds/wave {v.get('SQ_WAIT_INST_LDS',0)/wc:5.2f}")
PY
