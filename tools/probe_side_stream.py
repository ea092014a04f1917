#!/usr/bin/env python3
"""Where do 60 ms per step go when the gather's host copy waits on the side stream (gloo rehearsal)?  Times, under the pipelined
detection loop, a tiny kernel + D2H on (a) a torch pool stream, (b) a stream made with hipStreamCreateWithFlags via a context,
(c) the detection stream itself."""
import sys, time
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aruco3_amd import _lib, synth, shard
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
d = ARDictionary.new_from_named_dict('ARUCO')
spec, _ = synth.config_spec(2)
fr, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(256)])
n, h, w, c = fr.shape
a = (fr.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
ctxs = [Detector(DetectorConfig(), d)._context() for _ in range(2)]
stream = torch.cuda.Stream(); side = torch.cuda.Stream(); side2 = torch.cuda.Stream(priority=-1)
for cx in ctxs:
    cx.set_stream(stream.cuda_stream)
    for _ in range(3): cx.detect_batch(*a, out_cap=n * 64)
dev = torch.device("cuda", 0)
pin = torch.empty((n, shard.record_bytes()), dtype=torch.uint8, pin_memory=True)
small = torch.zeros(1024, dtype=torch.uint8, device=dev)
pin_small = torch.empty(1024, dtype=torch.uint8, pin_memory=True)

def loop(variant, k=30):
    for i in range(2): ctxs[i].submit(*a, out_cap=n * 64)
    waits = []
    t_all = time.perf_counter()
    for i in range(k):
        cx = ctxs[i % 2]
        cx.collect()
        t0 = time.perf_counter()
        if variant == "pack on side, sync side":
            cx.set_stream(side.cuda_stream)
            with torch.cuda.stream(side):
                rec = shard.pack_detections_device(cx, n, 0, dev); pin.copy_(rec, non_blocking=True)
            cx.set_stream(stream.cuda_stream); side.synchronize()
        elif variant == "pack on high-priority side, sync":
            cx.set_stream(side2.cuda_stream)
            with torch.cuda.stream(side2):
                rec = shard.pack_detections_device(cx, n, 0, dev); pin.copy_(rec, non_blocking=True)
            cx.set_stream(stream.cuda_stream); side2.synchronize()
        elif variant == "tiny copy on side, sync side":
            with torch.cuda.stream(side):
                pin_small.copy_(small, non_blocking=True)
            side.synchronize()
        elif variant == "pack on detection stream, sync it":
            with torch.cuda.stream(stream):
                rec = shard.pack_detections_device(cx, n, 0, dev); pin.copy_(rec, non_blocking=True)
            stream.synchronize()
        waits.append(time.perf_counter() - t0)
        if i + 2 < k: cx.submit(*a, out_cap=n * 64)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t_all) / k * 1e3
    waits.sort()
    print(f"{variant:40s} {tot:8.3f} ms/step   host wait in the variant: median {waits[len(waits) // 2] * 1e3:8.3f} ms  max {waits[-1] * 1e3:8.3f} ms")

for v in ("none", "tiny copy on side, sync side", "pack on side, sync side", "pack on high-priority side, sync", "pack on detection stream, sync it"):
    loop(v)
