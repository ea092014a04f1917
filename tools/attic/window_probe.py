#!/usr/bin/env python3
"""The threshold stage alone for every threshold_window (radius): BASELINE config 2's batch, synchronous calls, the stage between
events (nothing else on the GPU).  Radii 1..7 run the register-resident kernel templated on the radius, 8..31 the fused ring kernel
(k_threshold_big.hip: one wave per SIMD from 27 on), larger ones the separable path (grey plane + row sums + column sums).   python tools/window_probe.py [frames] [radii...]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    radii = [int(x) for x in sys.argv[2:]] or [7, 1, 2, 3, 4, 5, 6, 8, 11]
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
    n, h, w, c = d_frames.shape
    a = (d_frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    base = None
    print(f"threshold stage alone, {n} x {w}x{h} RGB frames, ms per batch (mean of 12 synchronous batches)")
    for r in radii:
        ctx = Detector(DetectorConfig(threshold_window=r), d)._context()
        for _ in range(3):
            m, per = ctx.detect_batch(*a, out_cap=n * 64)
        ctx.set_profiling(True)
        ctx.profile(_lib.STAGE_THRESHOLD, reset=True)
        for _ in range(12):
            ctx.detect_batch(*a, out_cap=n * 64)
        ms, cnt = ctx.profile(_lib.STAGE_THRESHOLD, reset=True)
        t = ms / cnt
        base = base or t
        print(f"threshold_window {r:2d} ({2 * r + 1:2d} x {2 * r + 1:2d}): {t:.4f} ms  ({t / base:.2f} x window {radii[0]})  markers {len(m)}", flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
