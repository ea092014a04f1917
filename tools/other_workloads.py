"""Throughput of the non-headline workloads (run on the GPU box): the reference bench's noise recipe (C0), config 4 and config 5."""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from aruco3_amd import _lib, synth, pose
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary
from oracle import a3oracle

def timeit(fn, iters):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters

def gpu_vs_cpu(name, frames, dname, iters=5, extra=None, pose_mm=None):
    d = ARDictionary.new_from_named_dict(dname)
    det = Detector(DetectorConfig(), d); ctx = det._context(); ctx.set_profiling(True)
    t = torch.from_numpy(frames).cuda(); n, h, w, c = frames.shape
    res = {}
    def step():
        if pose_mm:   # detect + IPPE pose of every marker in one device pass (a3_detect_batch_pose)
            res['m'], res['p'], res['poses'] = ctx.detect_batch_pose(t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, pose_mm, None, n * 64)
        else:
            res['m'], res['p'] = ctx.detect_batch(t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, out_cap=n * 64)
        if extra: extra(res['m'])
    dt = timeit(step, iters)
    k = min(n, 8); t0 = time.perf_counter()
    for f in range(k): a3oracle.detect_markers_only(frames[f], d.code_list, d.num_bits, d._tau)
    cpu = k / (time.perf_counter() - t0)
    st = ctx.stats()
    print(f'{name}: GPU {n / dt:,.0f} frames/s ({dt * 1e3:.2f} ms / {n} frames), CPU oracle 1 thread {cpu:,.1f} frames/s; '
          f'darts/frame {st["darts"] // n:,}, borders/frame {st["contours_traced"] // n:,}, chunks {st["chunks"]}, markers {len(res["m"])}', flush=True)

noise = np.stack([synth.noise_frame(1920, 1080, 100 + i) for i in range(32)])
gpu_vs_cpu('C0 noise 1920x1080 x32', noise, 'ARUCO', iters=3)
f4, _ = synth.config_frames(4, 32)
gpu_vs_cpu('C4 apriltag36h11 1280x720 sigma8 x32', f4, 'APRILTAG_36H11', iters=3)
f5, _ = synth.config_frames(5, 8)
gpu_vs_cpu('C5 3840x2160 x8 detect+pose', f5, 'ARUCO', iters=5, pose_mm=40.0)
f5b, _ = synth.config_frames(5, 32)
gpu_vs_cpu('C5 3840x2160 x32 detect+pose', f5b, 'ARUCO', iters=5, pose_mm=40.0)
f1, _ = synth.config_frames(1, 1)
gpu_vs_cpu('C1 640x480 x1 (latency of one small frame)', f1, 'ARUCO_DEFAULT', iters=20)
