#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"
echo "== 256 frames per batch"; timeout -k 10 300 python3 tools/ab_streams.py 256 48 3 own:2:2:0,own:3:2:0,own:4:2:0,own:6:2:0 2>&1 | grep -v amdgpu.ids
echo "== 128 frames per batch"; timeout -k 10 300 python3 tools/ab_streams.py 128 96 3 own:2:2:0,own:4:2:0,own:6:2:0,own:8:2:0 2>&1 | grep -v amdgpu.ids
echo "== 64 frames per batch"; timeout -k 10 300 python3 tools/ab_streams.py 64 192 3 own:4:2:0,own:8:2:0 2>&1 | grep -v amdgpu.ids
