#!/usr/bin/env python3
"""On the GPU box: the reference bench's noise recipe (32 x 1920x1080) and BASELINE config 2 (256 frames), synchronous calls, median
of 7 -- for A/B runs of two builds of the library (A3_HIP_LIB selects the copy)."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
d = ARDictionary.new_from_named_dict('ARUCO')
g = torch.Generator(device='cuda'); g.manual_seed(1)
noise = torch.randint(0, 256, (32, 1080, 1920, 3), dtype=torch.uint8, device='cuda', generator=g)
spec, _ = synth.config_spec(2)
c2, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(256)])
for name, t in (("noise32", noise), ("config2", c2)):
    ctx = Detector(DetectorConfig(), d)._context()
    n, h, w, c = t.shape
    a = (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    for _ in range(3): ctx.detect_batch(*a, out_cap=n * 64)
    ts = []
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter(); ctx.detect_batch(*a, out_cap=n * 64); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, f"{sorted(ts)[3] * 1e3:.3f} ms per batch", ctx.stats()["contours_materialised"], end="   ")
print()
