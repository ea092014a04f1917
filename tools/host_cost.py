"""Host-side cost of one batch: time spent inside submit() (every launch enqueued) and inside collect() (GPU box)."""
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
st = torch.cuda.Stream()
ctxs = [Detector(DetectorConfig(), d)._context() for _ in range(2)]
args = (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
for cx in ctxs:
    cx.set_stream(st.cuda_stream)
    for _ in range(3): cx.detect_batch(*args, out_cap=n * 64)
ts, tc = [], []
ctxs[0].submit(*args, out_cap=n * 64)
for it in range(40):
    a = time.perf_counter(); ctxs[(it + 1) % 2].submit(*args, out_cap=n * 64); b = time.perf_counter()
    ctxs[it % 2].collect(); c2 = time.perf_counter()
    ts.append(b - a); tc.append(c2 - b)
ctxs[0].collect()   # the batch submitted by the last iteration
print(f"submit: median {np.median(ts) * 1e6:.0f} us (min {min(ts) * 1e6:.0f}), collect (incl. waiting for the GPU): median {np.median(tc) * 1e6:.0f} us")
# the same without a GPU queue ahead: submit on an idle stream
torch.cuda.synchronize()
ts2 = []
for it in range(20):
    torch.cuda.synchronize(); a = time.perf_counter(); ctxs[0].submit(*args, out_cap=n * 64); b = time.perf_counter(); ctxs[0].collect(); ts2.append(b - a)
print(f"submit on an idle stream: median {np.median(ts2) * 1e6:.0f} us")
