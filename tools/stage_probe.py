"""stage times of isolated synchronous batches: the same batch again and again vs four different batches in turn (do the library's
per-context hints -- plan, pass counts, marker guess -- cost anything when consecutive batches differ?)"""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

spec, name = synth.config_spec(2)
d = ARDictionary.new_from_named_dict(name)
bufs = []
for j in range(4):
    df, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, 256 * j + i) for i in range(256)], device=0)
    bufs.append(df)
n, h, w, c = bufs[0].shape
args = [(b.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n) for b in bufs]
ctx = Detector(DetectorConfig.default(), d)._context()
for a in args:
    ctx.detect_batch(*a, out_cap=n * 64); ctx.detect_batch(*a, out_cap=n * 64)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"))
for label, seq in (("same batch", [0] * 16), ("four batches in turn", [0, 1, 2, 3] * 4), ("same batch", [1] * 16)):
    ctx.set_profiling(True)
    for st in range(3):
        ctx.profile(st, reset=True)
    reruns, darts = 0, []
    for j in seq:
        ctx.detect_batch(*args[j], out_cap=n * 64)
        st = ctx.stats(); reruns += st["reruns"]; darts.append(st["darts"])
    t = [ctx.profile(st, reset=True) for st in range(3)]
    ctx.set_profiling(0)
    print(f"{label:22s} threshold {t[0][0] / t[0][1]:.4f}  contour {t[1][0] / t[1][1]:.4f}  decode {t[2][0] / t[2][1]:.4f} ms  launches {t[0][1]}  reruns {reruns}  darts {sorted(set(darts))}")
