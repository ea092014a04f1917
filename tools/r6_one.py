#!/usr/bin/env python3
"""One frame per call (BASELINE config 1: 640x480, 4 markers; and the reference bench's 1080p noise frame): median ms per synchronous call
from pinned host memory, stage times of the same calls (events), and the kernel launches of one call counted from the stats.
  A3_HIP_LIB=... python tools/r6_one.py [calls]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for wl in ("c1", "noise1080"):
    if wl == "c1":
        frames, _ = synth.config_frames(1, 1)
        d = ARDictionary.new_from_named_dict("ARUCO_DEFAULT")
    else:
        frames = np.random.default_rng(5).integers(0, 256, (1, 1080, 1920, 3), dtype=np.uint8)
        d = ARDictionary.new_from_named_dict("ARUCO")
    n, h, w, c = frames.shape
    pin = _lib.PinnedBuffer(frames.nbytes)
    pin.array[:] = frames.reshape(-1)
    ctx = Detector(DetectorConfig.default(), d)._context()
    args = (pin.ptr, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, 1)
    for _ in range(20):
        m, per = ctx.detect_batch(*args)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter(); m, per = ctx.detect_batch(*args); ts.append(time.perf_counter() - t0)
    ts.sort()
    ctx.set_profiling(True)
    for st in range(3):
        ctx.profile(st, reset=True)
    for _ in range(50):
        ctx.detect_batch(*args)
    t = [ctx.profile(st, reset=True) for st in range(3)]
    ctx.set_profiling(0)
    print(f"{wl:10s} lib {_lib.library_info()['path'][-34:]}  median {ts[len(ts) // 2] * 1e3:.4f} ms  p10 {ts[len(ts) // 10] * 1e3:.4f}  markers {len(m)}  "
          f"stages (with events between them) threshold {t[0][0] / t[0][1] * 1e3:.1f}  contour {t[1][0] / t[1][1] * 1e3:.1f}  decode {t[2][0] / t[2][1] * 1e3:.1f} us", flush=True)
    pin.close()
