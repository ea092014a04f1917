"""Frames/s when the frames start in host memory (the PCIe-inclusive figure, never reported as `value`): pageable
numpy buffer handed to a3_detect_batch (MEM_HOST), and a pinned buffer copied with torch then detected in place (GPU box)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from aruco3_amd import _lib
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

z = np.load('/tmp/c2frames.n256.r0.npz', allow_pickle=True)['frames']
n, h, w, c = z.shape
d = ARDictionary.new_from_named_dict('ARUCO')
ctx = Detector(DetectorConfig(), d)._context()
for _ in range(2): ctx.detect_batch(z.ctypes.data, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
t0 = time.perf_counter()
for _ in range(5): m, p = ctx.detect_batch(z.ctypes.data, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
dt = (time.perf_counter() - t0) / 5
print(f"pageable host frames, a3_detect_batch(MEM_HOST): {n / dt:,.0f} frames/s ({dt * 1e3:.1f} ms per {n} frames, {z.nbytes / dt / 1e9:.1f} GB/s of input)")
pin = torch.from_numpy(z).pin_memory()
dev = torch.empty_like(pin, device='cuda')
for _ in range(2):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    ctx.detect_batch(dev.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
    m, p = ctx.detect_batch(dev.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
dt = (time.perf_counter() - t0) / 5
print(f"pinned host frames, copy then detect (not overlapped): {n / dt:,.0f} frames/s ({dt * 1e3:.1f} ms per {n} frames)")
t0 = time.perf_counter()
for _ in range(5):
    dev.copy_(pin, non_blocking=True); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"pinned H2D alone: {z.nbytes / dt / 1e9:.1f} GB/s = {n / dt:,.0f} frames/s")
