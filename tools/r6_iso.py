#!/usr/bin/env python3
"""BASELINE config 2, one context, synchronous batches (every kernel runs alone): the workload tools/r6_ab.sh puts under
rocprofv3 --kernel-trace --stats for each library build.  Prints the stage times (HIP events) and a checksum of the markers.
  python tools/r6_iso.py [frames] [batches] [workload: c2 | noise | c4 | one]"""
import sys
import zlib
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 12
wl = sys.argv[3] if len(sys.argv) > 3 else "c2"
if wl == "noise":     # benches/detect_markers.rs:29-51: uniform noise, 1920x1080
    d = ARDictionary.new_from_named_dict("ARUCO")
    g = torch.Generator(device="cuda").manual_seed(7)
    dev = torch.randint(0, 256, (frames, 1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
elif wl == "c4":
    spec, name = synth.config_spec(4)
    d = ARDictionary.new_from_named_dict(name)
    dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(4, i) for i in range(frames)])
elif wl == "one":     # BASELINE config 1: one 640x480 frame per call
    spec, name = synth.config_spec(1)
    d = ARDictionary.new_from_named_dict(name)
    dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(1, i) for i in range(frames)])
else:
    spec, name = synth.config_spec(2)
    d = ARDictionary.new_from_named_dict(name)
    dev, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
n, h, w, c = dev.shape
args = (dev.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
ctx = Detector(DetectorConfig.default(), d)._context()
for _ in range(3):
    m, per = ctx.detect_batch(*args, out_cap=n * 64)
ctx.set_profiling(True)
for st in range(3):
    ctx.profile(st, reset=True)
for _ in range(batches):
    m, per = ctx.detect_batch(*args, out_cap=n * 64)
t = [ctx.profile(st, reset=True) for st in range(3)]
ctx.set_profiling(0)
crc = zlib.crc32(np.ascontiguousarray(m[["frame", "id", "code", "corners", "hamming_distance", "rotation", "candidate_index"]]).tobytes()) if len(m) else 0
print(f"{wl} {n}x{w}x{h} lib {_lib.library_info()['path'][-40:]}  threshold {t[0][0] / t[0][1]:.4f}  contour {t[1][0] / t[1][1]:.4f}  decode {t[2][0] / t[2][1]:.4f} ms"
      f"  markers {len(m)} crc {crc:08x}  stats {ctx.stats()}")
