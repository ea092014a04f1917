"""One-off soak on the GPU box: the randomised structured-frame parity check of tests/test_gpu_parity.py over many more seeds,
plus random noise / mixed batches of one shape in a row (exercises the device-side plan, its fallback, both entry modes)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_parity as T
from oracle import a3oracle
from aruco3_amd.dictionaries import ARDictionary

class Dicts:
    new_from_named_dict = staticmethod(ARDictionary.new_from_named_dict)

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 150
t0 = time.time(); cases = 0
for seed in range(100, 100 + n_seeds):
    T.test_randomised_structured_frames_full_parity.__wrapped__(Dicts, a3oracle, seed) if hasattr(T.test_randomised_structured_frames_full_parity, "__wrapped__") else T.test_randomised_structured_frames_full_parity(Dicts, a3oracle, seed)
    cases += 5
    if seed % 25 == 0: print(f"seed {seed} ok, {time.time() - t0:.0f} s", flush=True)
# same-shape sequences with changing content
rng = np.random.default_rng(77)
det = T._detector(Dicts, "ARUCO")
for it in range(60):
    h, w = 240, 320
    kind = rng.integers(0, 3)
    if kind == 0: frames = rng.integers(0, 256, size=(3, h, w, 3), dtype=np.uint8)
    elif kind == 1: frames = np.stack([np.repeat(T._fuzz_frame(rng, h, w, "quads")[..., None], 3, axis=2) for _ in range(3)])
    else: frames = np.stack([np.repeat(T._fuzz_frame(rng, h, w, "rects")[..., None], 3, axis=2), rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8), np.full((h, w, 3), 200, np.uint8)])
    T._check(det, a3oracle, frames, check_patches=False)
    cases += 1
print(f"soak ok: {cases} cases in {time.time() - t0:.0f} s")

# large frames of random sizes (many tiles, long borders crossing tile and word edges)
t1 = time.time()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    h, w = int(rng.integers(300, 1300)), int(rng.integers(300, 2100))
    kind = ["rects", "quads", "strokes"][int(rng.integers(0, 3))]
    c = int(rng.choice([1, 3, 4]))
    fr = np.stack([np.repeat(T._fuzz_frame(rng, h, w, kind)[..., None], c, axis=2) for _ in range(2)])
    T._check(det, a3oracle, fr if c > 1 else fr[..., 0][..., None], check_patches=False)
print(f"large frames ok in {time.time() - t1:.0f} s")

# round 3: the same kinds of frames through submit / collect on two contexts (host frames on the copy stream, the decode stage of
# one batch deferred beside the contour stage of the next), against the synchronous call on a third context; shapes change from
# batch to batch, so plans, pools and the deferred halves are re-made all the time
from aruco3_amd import _lib
from tests.util import marker_tuples
t2 = time.time()
dets = [T._detector(Dicts, "ARUCO") for _ in range(3)]
ctxs = [x._context() for x in dets]
def batch():
    h, w = int(rng.integers(120, 700)), int(rng.integers(160, 1000))
    kind = ["rects", "quads", "strokes", "noise"][int(rng.integers(0, 4))]
    n = int(rng.integers(1, 5))
    if kind == "noise": fr = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    else: fr = np.stack([np.repeat(T._fuzz_frame(rng, h, w, kind)[..., None], 3, axis=2) for _ in range(n)])
    fr = np.ascontiguousarray(fr)
    return fr, (fr.ctypes.data, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * 3, h * w * 3, n)
n_pipe = int(sys.argv[3]) if len(sys.argv) > 3 else 200
pending = []
for it in range(n_pipe):
    fr, a = batch()
    want = ctxs[2].detect_batch(*a)
    cx = ctxs[it % 2]
    if len(pending) == 2:
        ocx, ofr, owant = pending.pop(0)
        got = ocx.collect()
        assert marker_tuples(got[0]) == marker_tuples(owant[0]) and np.array_equal(got[1], owant[1]), it
    cx.submit(*a)
    pending.append((cx, fr, want))
for ocx, ofr, owant in pending:
    got = ocx.collect()
    assert marker_tuples(got[0]) == marker_tuples(owant[0]) and np.array_equal(got[1], owant[1])
print(f"pipelined batches ok: {n_pipe} in {time.time() - t2:.0f} s")
