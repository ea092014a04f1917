#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"
for part in ${PARTS:-0 128:0 128:1 160:1 96:1 192:1}; do
  echo "== A3_PARTITION=$part"
  A3_PARTITION=$part timeout -k 10 200 python3 tools/ab_streams.py 256 40 3 ${SPECS:-own:2:2:0,own:3:2:0,own:4:2:0} 2>&1 | grep -v amdgpu.ids | cut -c1-140
done
