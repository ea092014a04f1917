#!/usr/bin/env python3
"""n = 1 latency: one 640x480 frame (or, with the argument `noise`, one 1920x1080 uniform-noise frame) from pinned host memory through a3_detect_batch, 60 calls (run under rocprofv3 --kernel-trace
to see how much of a call's wall time the GPU is busy: tools/trace_n1.sh)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
if len(sys.argv) > 1 and sys.argv[1] == "noise":   # the reference bench's recipe as it is called: one 1080p noise frame per call
    frames = np.random.default_rng(29).integers(0, 256, size=(1, 1080, 1920, 3), dtype=np.uint8)
    d = ARDictionary.new_from_named_dict("ARUCO")
else:
    frames, _ = synth.config_frames(1, 1)
    d = ARDictionary.new_from_named_dict("ARUCO_DEFAULT")
ctx = Detector(DetectorConfig.default(), d)._context()
n, h, w, c = frames.shape
pin = _lib.PinnedBuffer(frames.nbytes); pin.array[:] = frames.reshape(-1)
a = (pin.ptr, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, 1)
for _ in range(10): ctx.detect_batch(*a, out_cap=64)
ts = []
for _ in range(60):
    t0 = time.perf_counter(); ctx.detect_batch(*a, out_cap=64); ts.append(time.perf_counter() - t0)
ts.sort(); print("median call %.1f us" % (ts[30] * 1e6))
