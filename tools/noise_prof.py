import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
noise = np.stack([synth.noise_frame(1920, 1080, 100 + i) for i in range(32)])
det = Detector(DetectorConfig(), ARDictionary.new_from_named_dict('ARUCO')); ctx = det._context()
t = torch.from_numpy(noise).cuda(); n, h, w, c = noise.shape
for _ in range(4): ctx.detect_batch(t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
print(ctx.stats())
