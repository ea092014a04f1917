"""Time K1 alone-ish for L8 / RGB8 / RGBA8 inputs and compare with a plain device copy (run on the GPU box)."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from aruco3_amd import _lib
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

n, h, w = 256, 1080, 1920
z = np.load('/tmp/c2frames.n256.r0.npz', allow_pickle=True)['frames']
det = Detector(DetectorConfig(), ARDictionary.new_from_named_dict('ARUCO'))
ctx = det._context(); ctx.set_profiling(True)
for name, c in (('L8', 1), ('RGB8', 3), ('RGBA8', 4)):
    if c == 1: a = np.ascontiguousarray(z[..., 1])[..., None]
    elif c == 3: a = z
    else: a = np.concatenate([z, np.full(z.shape[:3] + (1,), 255, np.uint8)], axis=3)
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    fmt = {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c]
    for _ in range(2): ctx.detect_batch(t.data_ptr(), _lib.MEM_DEVICE, fmt, w, h, w * c, h * w * c, n, out_cap=n * 64)
    ctx.profile(_lib.STAGE_THRESHOLD, reset=True)
    for _ in range(6): ctx.detect_batch(t.data_ptr(), _lib.MEM_DEVICE, fmt, w, h, w * c, h * w * c, n, out_cap=n * 64)
    ms, k = ctx.profile(_lib.STAGE_THRESHOLD)
    px = n * h * w
    print(f'{name}: K1 {ms/k:.3f} ms  real bytes {(c + 1.125) * px / 1e9:.2f} GB -> {(c + 1.125) * px / (ms / k * 1e-3) / 1e12:.2f} TB/s')
    del t
# plain copy reference: read 1.59 GB + write 1.59 GB
src = torch.empty(n * h * w * 3, dtype=torch.uint8, device='cuda'); dst = torch.empty_like(src)
for _ in range(3): dst.copy_(src)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): dst.copy_(src)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'torch copy of {src.numel()/1e9:.2f} GB: {dt*1e3:.3f} ms -> {2*src.numel()/dt/1e12:.2f} TB/s (read+write)')
