#!/usr/bin/env python3
"""Does the threshold kernel's time alone depend on how its strips' addresses fall on the memory channels?  The same 256 frames laid
out with other frame and row strides (padding bytes between frames / rows; the library takes any), one synchronous batch at a time,
the threshold stage between HIP events (a3_set_profiling threshold-only).  python tools/k1_stride_probe.py [frames] [reps]"""
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
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
    n, h, w, c = d_frames.shape
    cx = Detector(DetectorConfig.default(), d)._context()
    ref = None
    gb = 3.125 * w * h * n / 1e9
    print(f"k_grey_threshold7 alone, {n} x {w}x{h} RGB; ms per launch (mean of {reps} synchronous batches, HIP events around the stage)")
    cases = [(0, 0), (0, 1024), (0, 4096), (256, 256), (0, 4096 + 256)] * (int(sys.argv[3]) if len(sys.argv) > 3 else 3)   # (cycled: the time alone drifts within a process)
    for row_pad, frame_pad in cases:
        rs = w * c + row_pad
        fs = rs * h + frame_pad
        buf = torch.zeros(n * fs + 64, dtype=torch.uint8, device=d_frames.device)
        view = torch.as_strided(buf, (n, h, w * c), (fs, rs, 1))
        view.copy_(d_frames.reshape(n, h, w * c))
        a = (buf.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, rs, fs, n)
        for _ in range(2):
            m, _pf = cx.detect_batch(*a, out_cap=n * 64)
        if ref is None:
            ref = m
        assert len(m) == len(ref) and (m["id"] == ref["id"]).all()
        cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY)
        cx.profile(0, reset=True)
        for _ in range(reps):
            cx.detect_batch(*a, out_cap=n * 64)
        ms, cnt = cx.profile(0, reset=True)
        cx.set_profiling(0)
        per = ms / max(cnt, 1)
        print(f"row stride {rs:5d} (+{row_pad:3d})  frame stride {fs:8d} (+{frame_pad:7d})  {per:.4f} ms  {gb / per:.2f} TB/s  {gb / per / 8:.3f} of 8 TB/s", flush=True)
        del buf, view
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
