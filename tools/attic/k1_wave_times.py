#!/usr/bin/env python3
"""When do the waves of ONE threshold-kernel launch start and end, and where do they run?  A -DA3_TUNING build stamps every wave
(s_memrealtime at its first and last instruction, HW_ID, XCC_ID); this prints, for a lone launch of BASELINE config 2's batch: the launch's
span, the spread of the waves' starts, durations and ends, and the mean duration per XCD.
  A3_HIP_LIB=build/tuning/libaruco3_hip.so python tools/k1_wave_times.py [frames] [launches]"""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    import numpy as np
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    launches = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, i) for i in range(frames)])
    n, h, w, c = d_frames.shape
    L = _lib.load()
    assert hasattr(L, "a3_debug_set_k1_stamps"), "needs a -DA3_TUNING build (A3_HIP_LIB)"
    L.a3_debug_set_k1_stamps.restype = C.c_int; L.a3_debug_set_k1_stamps.argtypes = [C.c_void_p]
    cx = Detector(DetectorConfig.default(), d)._context()
    n_wg = 8 * ((n + 7) // 8) * 2 * 8   # (more than any strip plan for these frames launches)
    buf = torch.zeros(n_wg * 4, dtype=torch.int64, device=d_frames.device)
    for _ in range(3):
        assert L.a3_debug_launch_threshold(cx.handle, C.c_void_p(d_frames.data_ptr()), _lib.FMT_RGB8, w, h, n) == 0
    torch.cuda.synchronize()
    assert L.a3_debug_set_k1_stamps(C.c_void_p(buf.data_ptr())) == 0
    for it in range(launches):
        buf.zero_(); torch.cuda.synchronize()
        assert L.a3_debug_launch_threshold(cx.handle, C.c_void_p(d_frames.data_ptr()), _lib.FMT_RGB8, w, h, n) == 0
        torch.cuda.synchronize()
        a = buf.cpu().numpy().reshape(-1, 4)
        blk = np.nonzero(a[:, 1] != 0)[0]
        a = a[a[:, 1] != 0]
        t0 = a[:, 0].min()
        beg, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0   # us
        dur = end - beg
        xcc = a[:, 3] & 0xF
        q = lambda v, p: float(np.percentile(v, p))
        print(f"launch {it}: {len(a)} waves, span {end.max():.1f} us | starts: median {q(beg, 50):.1f} p95 {q(beg, 95):.1f} max {beg.max():.1f} | "
              f"durations: min {dur.min():.1f} median {q(dur, 50):.1f} p95 {q(dur, 95):.1f} max {dur.max():.1f} | ends: p5 {q(end, 5):.1f} median {q(end, 50):.1f} p95 {q(end, 95):.1f}")
        per = [f"{x}: {dur[xcc == x].mean():.1f} ({(xcc == x).sum()})" for x in sorted(set(xcc.tolist()))]
        print("   mean duration per XCD (waves): " + "  ".join(per))
        # by strip: BASELINE config 2's plan is 2 column strips x 4 row strips per frame, workgroup b -> idx = (b >> 3) % 8, sx = idx // 4, sy = idx % 4
        if n == 256 and w == 1920 and h == 1080:
            idx = (blk >> 3) % 8
            print("   mean duration per (column strip, row strip): " + "  ".join(f"({i // 4},{i % 4}): {dur[idx == i].mean():.1f}" for i in range(8)))
        half = blk >= (blk.max() + 1) // 2
        print(f"   mean duration, first half of the grid {dur[~half].mean():.1f} (ends median {np.median(end[~half]):.1f}), second half {dur[half].mean():.1f} (ends median {np.median(end[half]):.1f})")
        wslot = a[:, 2] & 0xF
        print("   mean duration per wave slot (HW_ID.wave_id): " + "  ".join(f"{x}: {dur[wslot == x].mean():.1f} ({(wslot == x).sum()})" for x in sorted(set(wslot.tolist()))))
        simd = (a[:, 2] >> 4) & 0x3
        print("   mean duration per SIMD: " + "  ".join(f"{x}: {dur[simd == x].mean():.1f} ({(simd == x).sum()})" for x in range(4)))
        # how busy is the chip over time: waves alive per 10-us bin
        bins = np.arange(0, end.max() + 10, 10.0)
        alive = [int(((beg < b + 10) & (end > b)).sum()) for b in bins]
        print("   waves alive per 10 us: " + " ".join(str(v) for v in alive))
    L.a3_debug_set_k1_stamps(None)


if __name__ == "__main__":
    main()
