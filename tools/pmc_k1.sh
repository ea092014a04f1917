#!/bin/bash
# PMC passes for the bench (run on the GPU box): SQ activity, FETCH_SIZE, WRITE_SIZE -- each in its own pass.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 3 --warmup 1 --frames-cache /tmp/c2frames --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o write -- python3 $ARGS > $OUT/write.log 2>&1
ls $OUT
