#!/usr/bin/env python3
"""How do several threshold-kernel launches in flight at once share the chip?  N launches (BASELINE config 2's batch each) enqueued
on ONE stream (they run one after the other) against the same N launches on N streams (the dispatcher refills every slot a
finishing wave frees with a wave of the next launch): wall time per launch, bytes that must move per second -- each, since round 5,
once with every launch reading ITS OWN batch and once with all of them reading ONE batch (round 4's set-up: launches that read the
same 1.59 GB at about the same time serve one another out of the Infinity Cache; the "0.71-0.77 of the peak in a burst" came from that).
  python tools/k1_concurrency.py [frames] [reps]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def main():
    import ctypes as C

    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    bufs = [synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, j * frames + i) for i in range(frames)])[0] for j in range(6)]
    d_frames = bufs[0]
    n, h, w, c = d_frames.shape
    L = _lib.load()
    ctxs = [Detector(DetectorConfig.default(), d)._context() for _ in range(6)]
    shared = torch.cuda.Stream()
    gb = 3.125 * w * h * n / 1e9

    def run(nl, own, distinct):
        for cx in ctxs[:nl]:
            cx.set_stream(0 if own else shared.cuda_stream)
        ts = []
        for _ in range(reps + 2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k, cx in enumerate(ctxs[:nl]):
                assert L.a3_debug_launch_threshold(cx.handle, C.c_void_p((bufs[k] if distinct else d_frames).data_ptr()), _lib.FMT_RGB8, w, h, n) == 0
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ts = sorted(ts[2:])
        return ts[len(ts) // 2] / nl * 1e3

    print(f"k_grey_threshold7, {n} x {w}x{h} RGB per launch ({gb:.3f} GB must move); ms per launch (median of {reps}), launch + sync overhead included")
    for nl in (1, 2, 3, 4, 6):
        for distinct in (True, False):
            a, b = run(nl, False, distinct), run(nl, True, distinct)
            print(f"{nl} launches, {'a batch each' if distinct else 'ONE batch   '}: one stream {a:.4f} ms ({gb / a:.2f} TB/s, {gb / a / 8:.3f} of 8 TB/s)   {nl} streams {b:.4f} ms "
                  f"({gb / b:.2f} TB/s, {gb / b / 8:.3f})", flush=True)


if __name__ == "__main__":
    main()
