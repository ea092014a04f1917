"""one synchronous batch workload under the profiler: `c0` = the reference's bench recipe (uniform noise 1920x1080 x 32),
`c4` = BASELINE config 4 (APRILTAG_36H11, 1280x720, sigma 8, x 32).  Prints the stats and the stage times."""
import sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
wl = sys.argv[1] if len(sys.argv) > 1 else "c0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
if wl == "c0":
    g = torch.Generator(device="cuda"); g.manual_seed(20261004)
    t = torch.randint(0, 256, (32, 1080, 1920, 3), dtype=torch.uint8, device="cuda", generator=g)
    d = ARDictionary.new_from_named_dict("ARUCO")
else:
    spec, name = synth.config_spec(4)
    d = ARDictionary.new_from_named_dict(name)
    t, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(4, i) for i in range(32)])
ctx = Detector(DetectorConfig(), d)._context()
n, h, w, c = t.shape
a = (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
for _ in range(2): ctx.detect_batch(*a, out_cap=n * 64)
ctx.set_profiling(True)
for _ in range(reps): ctx.detect_batch(*a, out_cap=n * 64)
print(wl, ctx.stats(), {k: round(ctx.profile(i)[0] / max(ctx.profile(i)[1], 1), 4) for i, k in enumerate(("threshold", "contour", "decode"))})
