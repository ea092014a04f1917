#!/usr/bin/env python3
"""k_decode<256,64> against k_decode_pipe (round 6) in ONE process, on a -DA3_TUNING build (A3_DECODE_PIPE = workgroups of the
sampler / finisher form, 0 = the product kernel): markers of a synchronous batch (crc), the kernel alone warm / cold
(a3_debug_kernel_time 3 / 4), the decode stage inside isolated batches (events).
  A3_HIP_LIB=build/tuning/libaruco3_hip.so python tools/r6_decode_ab.py [values]"""
import os, sys, zlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

vals = (sys.argv[1] if len(sys.argv) > 1 else "0,640,1024,1280,2048,4096").split(",")
d = ARDictionary.new_from_named_dict("ARUCO")
spec, _ = synth.config_spec(2)
dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(256)])
n, h, w, c = dev.shape
args = (dev.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
ctx = Detector(DetectorConfig.default(), d)._context()
crc0 = None
for rep in range(2):
    for v in vals:
        os.environ["A3_DECODE_PIPE"] = v
        for _ in range(2):
            m, per = ctx.detect_batch(*args, out_cap=n * 64)
        crc = zlib.crc32(np.ascontiguousarray(m[["frame", "id", "code", "corners", "hamming_distance", "rotation", "candidate_index"]]).tobytes())
        crc0 = crc if crc0 is None else crc0
        ctx.set_profiling(True)
        for st in range(3):
            ctx.profile(st, reset=True)
        for _ in range(8):
            ctx.detect_batch(*args, out_cap=n * 64)
        t = [ctx.profile(st, reset=True) for st in range(3)]
        ctx.set_profiling(0)
        warm = ctx.debug_kernel_time(3, -5, 10) * 1e3
        cold = ctx.debug_kernel_time(4, -5, 10) * 1e3
        samp = ctx.debug_kernel_time(4, -2, 10) * 1e3
        print(f"A3_DECODE_PIPE={v:>5s}  markers {len(m)} {'same' if crc == crc0 else 'DIFFERENT'}  decode stage in isolated batches {t[2][0] / t[2][1] * 1e3:7.1f} us   "
              f"kernel alone warm {warm:6.1f}  cold {cold:6.1f} (cut after the sampling {samp:6.1f}) us", flush=True)
