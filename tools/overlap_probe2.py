"""Diagnose concurrency: one context at 128/256 frames, two contexts run back to back from one thread, two from two threads."""
import sys, time, threading
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from aruco3_amd import _lib
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

z = np.load('/tmp/c2frames.n256.r0.npz', allow_pickle=True)['frames']
n, h, w, c = z.shape
t = torch.from_numpy(z).cuda(); torch.cuda.synchronize()
d = ARDictionary.new_from_named_dict('ARUCO')

def run(ctx, ptr, frames, iters):
    for _ in range(iters):
        ctx.detect_batch(ptr, _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, frames, out_cap=frames * 64)

dets = [Detector(DetectorConfig(), d) for _ in range(2)]
ctxs = [x._context() for x in dets]
half = n // 2
ptrs = [t.data_ptr(), t.data_ptr() + half * h * w * c]
for nn in (256, 128):
    run(ctxs[0], ptrs[0], nn, 3)
    t0 = time.perf_counter(); run(ctxs[0], ptrs[0], nn, 10); dt = time.perf_counter() - t0
    print(f"one context, {nn} frames: {dt / 10 * 1e3:.3f} ms per batch, stats {ctxs[0].stats()}")
run(ctxs[1], ptrs[1], half, 3)
t0 = time.perf_counter()
for _ in range(10):
    run(ctxs[0], ptrs[0], half, 1); run(ctxs[1], ptrs[1], half, 1)
dt = time.perf_counter() - t0
print(f"two contexts back to back, one thread: {dt / 10 * 1e3:.3f} ms per 256 frames")
th = [threading.Thread(target=run, args=(ctxs[i], ptrs[i], half, 10)) for i in range(2)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
dt = time.perf_counter() - t0
print(f"two contexts, two threads: {dt / 10 * 1e3:.3f} ms per 256 frames")
