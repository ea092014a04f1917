"""Potential of running two half-batches concurrently (submit/collect on two streams) vs one full batch (GPU box)."""
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
t = torch.from_numpy(z).cuda(); torch.cuda.synchronize()
d = ARDictionary.new_from_named_dict('ARUCO')
args = lambda first, cnt: (t.data_ptr() + first * h * w * c, _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, cnt)

def bench(parts, same_stream, iters=20):
    ctxs = [Detector(DetectorConfig(), d)._context() for _ in range(2 * parts)]   # two generations in flight
    streams = [torch.cuda.Stream() for _ in range(parts)]
    per = n // parts
    for i, cx in enumerate(ctxs):
        cx.set_stream(streams[0 if same_stream else i % parts].cuda_stream)
        for _ in range(3): cx.detect_batch(*args((i % parts) * per, per), out_cap=per * 64)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for p in range(parts): ctxs[p].submit(*args(p * per, per), out_cap=per * 64)
    for it in range(iters):
        g, g2 = (it % 2) * parts, ((it + 1) % 2) * parts
        if it + 1 < iters:
            for p in range(parts): ctxs[g2 + p].submit(*args(p * per, per), out_cap=per * 64)
        tot = 0
        for p in range(parts): tot += len(ctxs[g + p].collect()[0])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
    print(f"{parts} part(s) x {per} frames, {'one stream' if same_stream else 'own streams'}: {dt * 1e3:.3f} ms per {n} frames = {n / dt:,.0f} frames/s, markers {tot}", flush=True)

bench(1, True)
bench(2, True)
bench(2, False)
bench(4, False)

import threading
def bench_threads(parts, iters=20):
    """one host thread per part, each ping-ponging two contexts on its own stream"""
    per = n // parts
    groups = []
    for p in range(parts):
        st = torch.cuda.Stream()
        cx = [Detector(DetectorConfig(), d)._context() for _ in range(2)]
        for c_ in cx:
            c_.set_stream(st.cuda_stream)
            for _ in range(3): c_.detect_batch(*args(p * per, per), out_cap=per * 64)
        groups.append(cx)
    bar = threading.Barrier(parts + 1)
    def work(p):
        cx = groups[p]; a = args(p * per, per)
        bar.wait()
        cx[0].submit(*a, out_cap=per * 64)
        for it in range(iters):
            if it + 1 < iters: cx[(it + 1) % 2].submit(*a, out_cap=per * 64)
            cx[it % 2].collect()
        bar.wait()
    th = [threading.Thread(target=work, args=(p,)) for p in range(parts)]
    for x in th: x.start()
    torch.cuda.synchronize(); bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = (time.perf_counter() - t0) / iters
    for x in th: x.join()
    print(f"{parts} thread(s) x {per} frames, own streams: {dt * 1e3:.3f} ms per {n} frames = {n / dt:,.0f} frames/s", flush=True)

bench_threads(1)
bench_threads(2)
bench_threads(3 if n % 3 == 0 else 4)
