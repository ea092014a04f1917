"""Does running two half-batches on two contexts/streams concurrently beat one full batch? (run on the GPU box)"""
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

for parts in (1, 2, 4):
    dets = [Detector(DetectorConfig(), d) for _ in range(parts)]
    ctxs = [x._context() for x in dets]
    per = n // parts
    ptrs = [t.data_ptr() + i * per * h * w * c for i in range(parts)]
    for i in range(parts): run(ctxs[i], ptrs[i], per, 2)
    iters = 10
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(ctxs[i], ptrs[i], per, iters)) for i in range(parts)]
    for x in th: x.start()
    for x in th: x.join()
    dt = time.perf_counter() - t0
    print(f'{parts} concurrent contexts x {per} frames: {n * iters / dt:,.0f} frames/s ({dt / iters * 1e3:.3f} ms per {n} frames)')
