#!/usr/bin/env python3
"""A/B inside ONE process (boxes of the pool differ by more than the effect): BASELINE config 2, two contexts in submit / collect,
with the decode stage of a submitted batch (0) not deferred, (1) released behind the next batch's threshold kernel, (2) behind its
k_local_contract.  Modes are interleaved ROUNDS times; prints the median ms per step of every mode and round.
  python tools/ab_overlap.py [frames] [steps] [rounds] [contexts]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    n_ctx = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    own_streams = len(sys.argv) > 5 and sys.argv[5] == "own"
    MODES = tuple(int(x) for x in sys.argv[6].split(",")) if len(sys.argv) > 6 else (0, 1, 2)   # every context on a stream of its own: batches overlap freely
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
    n, h, w, c = d_frames.shape
    a = (d_frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    L = _lib.load()
    stream = torch.cuda.Stream()
    # two sets of contexts, created one after the other: their buffers lie elsewhere, and that alone moves a step by ~2 %
    sets = {}
    for label in ("first set of contexts", "second set of contexts"):
        sets[label] = [Detector(DetectorConfig.default(), d)._context() for _ in range(n_ctx)]
        for cx in sets[label]:
            if not own_streams:
                cx.set_stream(stream.cuda_stream)
            for _ in range(3):
                ref = cx.detect_batch(*a, out_cap=n * 64)

    def run(k, ctxs):
        nc = len(ctxs)       # as bench.py steps: every context has a batch out; the one just collected submits the next at once
        for i in range(min(nc, k)):
            ctxs[i % nc].submit(*a, out_cap=n * 64)
        for i in range(k):
            m, per = ctxs[i % nc].collect()
            if i + nc < k:
                ctxs[i % nc].submit(*a, out_cap=n * 64)
        return m, per

    modes = [(m, lab) for lab in sets for m in MODES]
    res = {m: [] for m in modes}
    for r in range(rounds):
        for mode in modes:
            assert L.a3_debug_set_overlap(mode[0]) == 0
            run(6, sets[mode[1]])
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m, per = run(steps, sets[mode[1]])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            assert len(m) == len(ref[0]) and np.array_equal(per, ref[1])
            res[mode].append(dt / steps * 1e3)
    for mode in modes:
        v = sorted(res[mode])
        print(f"overlap mode {mode[0]}, {mode[1]}: median {v[len(v) // 2]:.4f} ms/step  ({n / v[len(v) // 2] * 1e3:.0f} frames/s)  all {[round(x, 4) for x in res[mode]]}")
    L.a3_debug_set_overlap(-1)


if __name__ == "__main__":
    main()
