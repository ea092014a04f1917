#!/usr/bin/env python3
"""sweep an environment knob of a -DA3_TUNING build over one kernel re-run alone (a3_debug_kernel_time):
  A3_HIP_LIB=build/tuning/libaruco3_hip.so python tools/r6_knob.py <kernel 0..4> <dbg> <KNOB> v1,v2,...  [frames]"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

kernel, dbg, knob, vals = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4].split(",")
n = int(sys.argv[5]) if len(sys.argv) > 5 else 256
d = ARDictionary.new_from_named_dict("ARUCO")
spec, _ = synth.config_spec(2)
dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(n)])
det = Detector(DetectorConfig(), d)
for _ in range(2):
    det.detect_batch_raw(dev)
ctx = det._context()
for rep in range(2):
    for v in vals:
        os.environ[knob] = v
        print(f"{knob}={v:>6s}  kernel {kernel} dbg {dbg}: {ctx.debug_kernel_time(kernel, dbg, 8) * 1e3:8.1f} us", flush=True)
