#!/usr/bin/env python3
"""bench.py -- frames/sec of the detection hot path on MI355X, with the roofline of the dominant
HBM-bound kernel and a CPU baseline measured beside it.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one pass of Detector::detect over one batch of synthetic frames that already sit in HBM:
BASELINE.json config 2, a batch of 256 x 1920x1080 RGB frames with 4-8 ARUCO markers each (config 3 is the
same batch per GPU on 8 GPUs: weak scaling, frames sharded by rank, dictionary broadcast once, detections
all-gathered per batch over RCCL).  One JSON line is printed by rank 0.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

FRAMES_PER_GPU = 256
WIDTH, HEIGHT = 1920, 1080
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
K1_BYTES_PER_PIXEL = 5             # 3 B RGB read + 1 B grey + 1 B binary written
K1_BYTES_MOVED_PER_PIXEL = 3.125   # what K1 moves now: 3 B read + 1/8 B packed binary written, no grey plane


def _render(args):
    from aruco3_amd import synth
    from aruco3_amd.dictionaries import ARDictionary

    config, idx = args
    spec, name = synth.config_spec(config)
    d = ARDictionary.new_from_named_dict(name)
    img, truth = synth.render_frame(spec, d.code_list, d.num_bits, synth.frame_seed(config, idx))
    return img, [t.id for t in truth]


def make_frames(first, count, workers):
    """Config-2 frames `first .. first+count` (seeded per frame index), rendered by a process pool on the host."""
    jobs = [(2, first + i) for i in range(count)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_render, jobs, chunksize=4)
    else:
        res = [_render(j) for j in jobs]
    frames = np.stack([r[0] for r in res])
    return frames, [r[1] for r in res]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-synth", action="store_true", help="render the frames on the GPU (a3_synth_render) instead of on the host: no "
                                                                  "host rendering, no H2D copy (same layouts and ids; pixels may differ by "
                                                                  "a grey level at cell edges)")
    ap.add_argument("--no-pipeline", action="store_true", help="one context, a3_detect_batch per step (the GPU idles while the host "
                                                                "collects a batch); default: two contexts on one stream, step i+1 is "
                                                                "submitted before step i is collected")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                       "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--synth-workers", type=int, default=0, help="host processes rendering frames (0 = auto)")
    ap.add_argument("--frames-cache", default="", help="npz path: reuse rendered frames between runs (profiling runs use it so that "
                                                        "nothing forks under the profiler)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 through torch.distributed.run")

    # host-side frame synthesis first (forks a pool; nothing has touched the GPU yet)
    workers = args.synth_workers or max(1, min(16, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    cache = Path(f"{args.frames_cache}.n{args.frames}.r{rank}.npz") if args.frames_cache else None
    if args.device_synth:
        frames, truth_ids = None, None        # rendered below, once the device is set up
    elif cache is not None and cache.exists():
        z = np.load(cache, allow_pickle=True)
        frames, truth_ids = z["frames"], [list(t) for t in z["truth"]]
        assert frames.shape[0] == args.frames
    else:
        frames, truth_ids = make_frames(rank * args.frames, args.frames, workers)
        if cache is not None:
            cache.parent.mkdir(parents=True, exist_ok=True)
            np.savez(cache, frames=frames, truth=np.array(truth_ids, dtype=object))
    t_gen = time.time() - t0

    import torch
    import torch.distributed as dist

    from aruco3_amd import _lib, shard
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    if args.backend != "nccl":   # rehearsal: ranks may share a GPU, collectives run on host tensors
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    d = ARDictionary.new_from_named_dict("ARUCO") if rank == 0 or world == 1 else None
    if world > 1:
        d = shard.broadcast_dictionary(d, coll_dev, 0)   # RCCL broadcast, once
    # two contexts on ONE stream: kernels of consecutive steps never overlap (K1 is timed alone), but the host enqueues step
    # i+1 while step i runs, so the GPU does not idle between steps
    dets = [Detector(DetectorConfig.default(), d, device=local_rank) for _ in range(1 if args.no_pipeline else 2)]
    ctxs = [x._context() for x in dets]
    stream = torch.cuda.Stream(device=dev)   # an explicit stream: handle 0 (the default stream) would mean "the context's own"
    for ctx in ctxs:
        ctx.set_stream(stream.cuda_stream)
        ctx.set_profiling(True)
    ctx = ctxs[0]

    if args.device_synth:
        from aruco3_amd import synth
        spec, _ = synth.config_spec(2)
        seeds = [synth.frame_seed(2, rank * args.frames + i) for i in range(args.frames)]
        t0 = time.time()
        d_frames, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds, device=local_rank)
        truth_ids = [[t.id for t in tr] for tr in truths]
        t_gen = time.time() - t0
        frames = d_frames.cpu().numpy() if (rank == 0 and not args.no_cpu_baseline and world == 1) else None   # only the CPU baseline reads them
    else:
        d_frames = torch.from_numpy(frames).to(dev)      # inputs resident in HBM before the timed region
    torch.cuda.synchronize()
    n, h, w, c = d_frames.shape
    first_frame = rank * args.frames

    batch_args = (d_frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)

    def run_steps(k):
        """k steps; a step = one pass of Detector::detect over the rank's batch, results on the host (and all-gathered)."""
        markers, per = None, None
        if args.no_pipeline:
            for _ in range(k):
                markers, per = ctx.detect_batch(*batch_args, out_cap=n * 64)
                if world > 1:
                    shard.gather_detections(markers, per, first_frame, coll_dev)   # RCCL all-gather of the compact records
            return markers, per
        if k > 0:
            ctxs[0].submit(*batch_args, out_cap=n * 64)
        for i in range(k):
            if i + 1 < k:
                ctxs[(i + 1) % 2].submit(*batch_args, out_cap=n * 64)
            markers, per = ctxs[i % 2].collect()
            if world > 1:
                shard.gather_detections(markers, per, first_frame, coll_dev)
        return markers, per

    # set-up, not steps: every context allocates its device buffers on its first batches (hipMalloc is slow and synchronous)
    for cx in ctxs:
        for _ in range(2):
            cx.detect_batch(*batch_args, out_cap=n * 64)
    for cx in ctxs:
        for st_id in (_lib.STAGE_THRESHOLD, _lib.STAGE_CONTOUR, _lib.STAGE_DECODE):
            cx.profile(st_id, reset=True)
    # Warm-up with every stage timed (the breakdown reported as stage_ms_per_step); the timed steps keep only the two event
    # records around the threshold kernel -- the roofline figure must be measured live -- because each record between two
    # kernels costs ~6 us of device time.
    markers, per = run_steps(args.warmup)
    stage_ms = {}
    for name, st_id in (("threshold", _lib.STAGE_THRESHOLD), ("contour", _lib.STAGE_CONTOUR), ("decode", _lib.STAGE_DECODE)):
        tot = cnt = 0
        for cx in ctxs:
            a, b = cx.profile(st_id, reset=True); tot += a; cnt += b
        stage_ms[name] = round(tot / cnt, 3) if cnt else None
    if args.warmup > 0:
        for cx in ctxs:
            cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    markers, per = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: what was rendered is what was read (ids per frame), on this rank's last step
    pos, id_ok = 0, 0
    for f in range(n):
        got = sorted(int(m["id"]) for m in markers[pos: pos + int(per[f])])
        pos += int(per[f])
        id_ok += got == sorted(truth_ids[f])

    k1_ms = k1_n = 0
    for cx in ctxs:
        a, b = cx.profile(_lib.STAGE_THRESHOLD); k1_ms += a; k1_n += b
    if args.warmup == 0:   # no warm-up to take the breakdown from: every stage was timed in the timed steps instead
        for name, st_id in (("contour", _lib.STAGE_CONTOUR), ("decode", _lib.STAGE_DECODE)):
            tot = sum(cx.profile(st_id)[0] for cx in ctxs)
            stage_ms[name] = round(tot / max(k1_n, 1), 3)
    stats = ctx.stats()

    if rank == 0:
        total_frames = args.frames * world * args.steps
        value = total_frames / elapsed
        k1_avg_ms = k1_ms / max(k1_n, 1)
        k1_bytes = K1_BYTES_PER_PIXEL * WIDTH * HEIGHT * args.frames          # algorithmic bytes per launch
        achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9 if k1_avg_ms > 0 else 0.0
        out = {
            "metric": "frames/sec at 1920x1080 ARUCO dict",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" + (" (rendered on the device)" if args.device_synth else ""),
            "config": {
                "workload": "BASELINE config 2: batch of 256 x 1920x1080 synthetic RGB frames per GPU, ARUCO dict, 4-8 markers per frame, "
                            "frames resident in HBM; Detector::detect end to end (grey, threshold, contours, quads, warp+decode, lookup) "
                            "incl. D2H of the marker list",
                "frames_per_gpu": args.frames,
                "resolution": [WIDTH, HEIGHT],
                "dictionary": "ARUCO",
                "sharding": "frames by rank, no data-path collective; dictionary broadcast once, detections all-gathered per batch" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": "k_grey_threshold7 (RGB->grey + 15x15 adaptive threshold)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_bytes(),
                "bytes_per_launch": k1_bytes,
                "avg_launch_ms": round(k1_avg_ms, 4),
                # what this kernel has to move: it reads the frame (3 B/px) and writes only the bit-packed binary image
                # (1/8 B/px) -- the grey plane of SURVEY's 5 B/px is never materialised (the decode stage re-derives the
                # grey levels it samples), so `achieved` counts a write the kernel avoids; this is the rate of real bytes
                "bytes_moved_per_launch": int(K1_BYTES_MOVED_PER_PIXEL * WIDTH * HEIGHT * args.frames),
                "moved_gbs": round(K1_BYTES_MOVED_PER_PIXEL * WIDTH * HEIGHT * args.frames / (k1_avg_ms * 1e-3) / 1e9, 1) if k1_avg_ms > 0 else 0.0,
            },
            # threshold: the timed steps; contour / decode: the warm-up steps (every stage timed there, see above)
            "stage_ms_per_step": {"threshold": round(k1_avg_ms, 3), "contour": stage_ms.get("contour"), "decode": stage_ms.get("decode")},
            "stepping": "one context, synchronous" if args.no_pipeline else "two contexts on one stream: step i+1 submitted before step i is collected",
            "stats": stats,
            "frames_with_all_ids_correct": f"{id_ok}/{n}",
            "frame_synthesis_s": round(t_gen, 1),
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, d)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def pmc_traffic_bytes():
    """HBM bytes per K1 launch from the committed PMC passes of this same command (tools/pmc_k1.sh ->
    profiles/r01_pmc_bench_c2.json): (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for wide coalesced reads on gfx950.  Counters cannot be read inside this process, so this is the profiled value for the
    default 256-frame batch, or None when the summary is missing."""
    path = ROOT / "profiles" / "r01_pmc_bench_c2.json"
    try:
        pmc = json.loads(path.read_text())
        k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
        return int((2.0 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
    except Exception:
        return None


def cpu_baseline(frames, d):
    """The CPU oracle (a restatement of the reference algorithm, NOT the Rust crate, which cannot be built here) on the
    same frames, one thread -- the reference's own execution model -- for about 10 s of CPU work."""
    from oracle import a3oracle

    a3oracle.build()
    codes = np.ascontiguousarray(d.code_list)
    done, t0 = 0, time.perf_counter()
    budget_s, max_frames = 10.0, 4 * len(frames)
    while done < max_frames:
        a3oracle.detect_markers_only(frames[done % len(frames)], codes, d.num_bits, d._tau)
        done += 1
        if done >= 32 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    out = {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} of the same 1920x1080 config-2 frames, single thread, oracle/a3_oracle.c (gcc -O2)",
           "host_cores_available": os.cpu_count()}
    # the same oracle, frame-parallel over the host's cores (one frame per worker; the C call releases the
    # GIL).  Informational: the reference itself is single-threaded.
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(os.cpu_count() or 1, 64))
    one = lambda f: a3oracle.detect_markers_only(frames[f % len(frames)], codes, d.num_bits, d._tau)
    done_mt, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        while time.perf_counter() - t0 < 6.0:
            list(pool.map(one, range(done_mt, done_mt + 4 * workers)))
            done_mt += 4 * workers
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": round(done_mt / dt, 2), "unit": "frames/s", "cores": workers,
                        "sample": f"{done_mt} frames, one frame per worker thread"}
    return out


if __name__ == "__main__":
    main()
