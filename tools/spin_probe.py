#!/usr/bin/env python3
"""What does a collective's company cost?  (VERDICT r03 #3a; one GPU at hand, so RCCL's channel kernels are played by a3_debug_spin:
W workgroups of 512 threads resident for T microseconds on a side stream, launched once per step where bench.py enqueues its
all-gather -- right after the next batch has been submitted.)  BASELINE config 2, the bench's stepping (four contexts on streams of
their own, burst gates).  For every (W, T): ms per step, and the threshold kernel's launch duration with such company
(events around the kernel).  Arrangements are interleaved ROUNDS times in one process.
  python tools/spin_probe.py [frames] [steps] [rounds]"""
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

CASES = [(0, 0, 0), (0, 0, -1), (8, 50, 0), (16, 50, 0), (16, 50, 1), (32, 50, 0), (32, 50, 1), (32, 200, 0), (32, 200, 1), (16, 600, 0), (16, 600, 1)]   # (W, T, mode): mode 0 = the
# spinner is released at once (the product: no event behind the threshold kernel); 1 = it waits for the threshold kernel of the batch just submitted
# (a3_debug_set_mark_threshold + a3_debug_stream_wait_threshold); -1 = no spinner, but the event is recorded behind every threshold kernel: its own cost


def main():
    import numpy as np
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else len(CASES)
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
    n, h, w, c = d_frames.shape
    a = (d_frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    L = _lib.load()
    assert L.a3_debug_set_overlap(0) == 0
    nc = 4
    ctxs = [Detector(DetectorConfig.default(), d)._context() for _ in range(nc)]
    for cx in ctxs:
        for _ in range(3):
            ref = cx.detect_batch(*a, out_cap=n * 64)
    side = torch.cuda.Stream()

    def run(k, wg, usec, profile, mode=0):
        assert L.a3_debug_set_mark_threshold(0 if mode == 0 else 1) == 0
        for cx in ctxs:
            cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY if profile else 0)
            cx.profile(_lib.STAGE_THRESHOLD, reset=True)

        def sub(i):
            for m in range(i % nc + 1, nc):
                ctxs[i % nc].order_after(ctxs[m])
            ctxs[i % nc].submit(*a, out_cap=n * 64)

        for i in range(min(nc, k)):
            sub(i)
        for i in range(k):
            m, per = ctxs[i % nc].collect()
            if i + nc < k:
                sub(i + nc)
            if wg:
                if mode == 1 and i + nc < k:
                    assert L.a3_debug_stream_wait_threshold(ctxs[i % nc].handle, side.cuda_stream) == 0
                assert L.a3_debug_spin(side.cuda_stream, wg, 512, usec) == 0
        tot = [cx.profile(_lib.STAGE_THRESHOLD, reset=True) for cx in ctxs]
        return m, per, sum(t for t, _ in tot) / max(sum(b for _, b in tot), 1)

    res = {cs: [] for cs in CASES}
    k1 = {cs: [] for cs in CASES}
    for r in range(rounds):
        order = CASES[r % len(CASES):] + CASES[:r % len(CASES)]      # every case takes every position in turn: what runs before a case moves it by 1-2 %
        run(2 * nc, 0, 0, False, 0)
        for wg, usec, mode in order:
            run(8, wg, usec, False, mode)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m, per, _ = run(steps, wg, usec, False, mode)
            torch.cuda.synchronize(); res[(wg, usec, mode)].append((time.perf_counter() - t0) / steps * 1e3)
            assert len(m) == len(ref[0]) and np.array_equal(per, ref[1])
            _, _, k = run(16, wg, usec, True, mode)
            torch.cuda.synchronize()
            k1[(wg, usec, mode)].append(k)
    L.a3_debug_set_mark_threshold(0)
    base = sorted(res[(0, 0, 0)])[len(res[(0, 0, 0)]) // 2]
    print(f"BASELINE config 2, {frames} frames per batch, {nc} contexts on their own streams with burst gates; spinner: W workgroups x 512 threads for T us per step, side stream")
    for wg, usec, mode in CASES:
        v = sorted(res[(wg, usec, mode)]); med = v[len(v) // 2]
        kk = sorted(k1[(wg, usec, mode)]); kmed = kk[len(kk) // 2]
        print(f"W {wg:3d}  T {usec:4d} us {('released behind the threshold kernel' if mode == 1 else ('event behind every threshold kernel' if mode < 0 else 'released at once')):38s}: {med:.4f} ms/step ({n / med * 1e3:8.0f} frames/s, {100 * (med / base - 1):+5.1f} %)   threshold kernel in company {kmed:.4f} ms   "
              f"all {[round(x, 4) for x in res[(wg, usec, mode)]]}", flush=True)


if __name__ == "__main__":
    main()
