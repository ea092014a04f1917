#!/bin/bash
# PMC passes for the bench (run on the GPU box): SQ activity, FETCH_SIZE, WRITE_SIZE -- each in its own pass, counters only
# (no --stats, no tracing domains besides --kernel-trace).  Output: gpurun_out/pmc/*_counter_collection.csv -> tools/pmc_aggregate.py
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --steps 3 --warmup 1 --repeats 1 --no-pipeline --isolated-launches 2 --device-synth --no-cpu-baseline --no-other-workloads"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o sq -- python3 $ARGS > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT -o fetch -- python3 $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o write -- python3 $ARGS > $OUT/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $OUT -o mem -- python3 $ARGS > $OUT/mem.log 2>&1
ls $OUT
