#!/usr/bin/env python3
"""Where does the time of the contour-stage kernels go?  Runs BASELINE config 2 once, then re-times single kernels and
truncated variants of them through a3_debug_kernel_time (GPU box only).  Usage: python tools/kernel_probe.py [frames]"""
import sys
import pathlib

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
d = ARDictionary.new_from_named_dict("ARUCO")
spec, _ = synth.config_spec(2)
dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(n)])
det = Detector(DetectorConfig(), d)
for _ in range(2):
    det.detect_batch_raw(dev)
ctx = det._context()
print("stats", ctx.stats())
# contract first (it needs a linked successor array), then the assign variants, the full one (5, relinks) last
for m in (0, -1, -2, -3, -4, -5):   # decode first: it only reads what the batch left behind (frames, work list)
    print(f"decode          dbg={m:3d}  {ctx.debug_kernel_time(3, m, 5) * 1e3:8.1f} us", flush=True)
if len(sys.argv) > 2 and sys.argv[2] == "decode":
    sys.exit(0)
for name, k, modes in (("local_contract", 2, (-1, 1, 2, 4, 6, 8, 11)), ("dart_count", 0, (0,)), ("dart_assign", 1, (1, 2, 3, 4, 5))):
    for m in modes:
        print(f"{name:15s} dbg={m:3d}  {ctx.debug_kernel_time(k, m, 5) * 1e3:8.1f} us", flush=True)
