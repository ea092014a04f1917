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
K1_BYTES_PER_PIXEL = 3.125         # what K1 has to move: 3 B RGB read + 1/8 B bit-packed binary written; NO grey plane is written
K1_SURVEY_BYTES_PER_PIXEL = 5      # SURVEY.md section 8d's figure (3 B read + 1 B grey + 1 B byte-wide binary): reported separately
PROFILE_TAG = "r02"                # profiles/<tag>_pmc_bench_c2.json holds the PMC passes of this same command


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
    ap.add_argument("--repeats", type=int, default=25, help="the timed region (exactly --steps steps between barrier + synchronize) is run "
                                                            "this many times back to back; the median is reported, every value listed "
                                                            "(20 steps are 20 ms: one region alone measures clocks ramping)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path (process group, dictionary broadcast, device-packed "
                                                                "records, all-gather) even with one rank: lets a 1-GPU box exercise the RCCL branch")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other_workloads block (reference bench recipe, configs 4 and 5)")
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
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    d = ARDictionary.new_from_named_dict("ARUCO") if rank == 0 or world == 1 else None
    if use_dist:
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

    last_gather = [None]
    side = torch.cuda.Stream(device=dev)   # the collective runs beside the detection stream, not in it

    def gather(cx):
        # Per batch: fixed-capacity records written by a kernel from the device-resident marker list (a3_pack_detections, on the
        # contexts' stream), then all-gathered over RCCL on a side stream that waits for the pack: the detection stream goes on
        # with the next batch instead of waiting for the collective; no host copy in between.
        with torch.cuda.stream(stream):
            rec = shard.pack_detections_device(cx, n, first_frame, dev)
            if coll_dev.type == "cpu":                  # gloo rehearsal: host tensors (the copy is ordered on the same stream)
                rec = rec.cpu()
        if coll_dev.type == "cpu":
            last_gather[0] = shard._all_gather(rec, n)
            return
        side.wait_stream(stream)
        with torch.cuda.stream(side):
            last_gather[0] = shard._all_gather(rec, n)
        rec.record_stream(side)

    def run_steps(k):
        """k steps; a step = one pass of Detector::detect over the rank's batch, results on the host (and all-gathered)."""
        markers, per = None, None
        if args.no_pipeline:
            for _ in range(k):
                markers, per = ctx.detect_batch(*batch_args, out_cap=n * 64)
                if use_dist:
                    gather(ctx)
            return markers, per
        if k > 0:
            ctxs[0].submit(*batch_args, out_cap=n * 64)
        for i in range(k):
            if i + 1 < k:
                ctxs[(i + 1) % 2].submit(*batch_args, out_cap=n * 64)
            markers, per = ctxs[i % 2].collect()
            if use_dist:
                gather(ctxs[i % 2])
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

    # The timed region: EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks.  It is run
    # --repeats times back to back and the median region is the one reported (all are listed in ms_per_step_all).
    regions = []
    for _ in range(max(1, args.repeats)):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        markers, per = run_steps(args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        regions.append(time.perf_counter() - t0)
    if use_dist:
        t = torch.tensor(regions, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in t.tolist()]
    elapsed = sorted(regions)[len(regions) // 2]

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

    gathered = None
    if use_dist:
        # what the last all-gather delivered, checked on rank 0: every rank's frames, global indices in order, ids as rendered
        torch.cuda.synchronize()
        g = last_gather[0].cpu().numpy()
        if rank == 0:
            from aruco3_amd import synth
            spec2, _ = synth.config_spec(2)
            recs = shard.unpack_detections(g.reshape(-1, g.shape[-1]))
            seeds_all = [synth.frame_seed(2, i) for i in range(world * args.frames)]
            truth_all = [sorted(t.id for t in tr) for tr in synth.device_layout(spec2, d.code_list, d.num_bits, seeds_all)[2]]
            gathered = {"frames": len(recs), "global_frame_indices_in_order": [f for f, _ in recs] == list(range(world * args.frames)),
                        "all_ranks_ids_correct": int(sum(sorted(int(x) for x in m["id"]) == truth_all[f] for f, m in recs)),
                        "record_bytes": int(g.shape[-1]), "packed_on": "device (a3_pack_detections)",
                        "collective": f"all_gather_into_tensor over {args.backend}"}

    if rank == 0:
        total_frames = args.frames * world * args.steps
        value = total_frames / elapsed
        k1_avg_ms = k1_ms / max(k1_n, 1)
        k1_bytes = int(K1_BYTES_PER_PIXEL * WIDTH * HEIGHT * args.frames)     # algorithmic bytes per launch = the bytes that move
        achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9 if k1_avg_ms > 0 else 0.0
        survey_gbs = K1_SURVEY_BYTES_PER_PIXEL * WIDTH * HEIGHT * args.frames / (k1_avg_ms * 1e-3) / 1e9 if k1_avg_ms > 0 else 0.0
        out = {
            "metric": "frames/sec at 1920x1080 ARUCO dict",
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "repeats": len(regions),
            "ms_per_step_all": [round(r / args.steps * 1e3, 4) for r in regions],
            "timed_region_s_total": round(sum(regions), 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" + (" (rendered on the device)" if args.device_synth else ""),
            "config": {
                "workload": f"BASELINE config 2: batch of {args.frames} x 1920x1080 synthetic RGB frames per GPU, ARUCO dict, 4-8 markers per frame, "
                            "frames resident in HBM; Detector::detect end to end (grey, threshold, contours, quads, warp+decode, lookup) "
                            "incl. D2H of the marker list",
                "frames_per_gpu": args.frames,
                "resolution": [WIDTH, HEIGHT],
                "dictionary": "ARUCO",
                "sharding": "frames by rank, no data-path collective; dictionary broadcast once, detections all-gathered per batch" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": "k_grey_threshold7 (RGB->grey + 15x15 adaptive threshold; output bit-packed, no grey plane)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_bytes(),
                # 3.125 B/px: the frame is read once (3 B/px) and only the bit-packed binary image (1/8 B/px) is written; the
                # grey plane of SURVEY's 5 B/px accounting is never materialised (the decode stage re-derives the grey levels
                # it samples).  achieved / frac count the bytes that move; the 5 B/px figure is kept under its own name.
                "bytes_per_pixel": K1_BYTES_PER_PIXEL,
                "bytes_per_launch": k1_bytes,
                "avg_launch_ms": round(k1_avg_ms, 4),
                "launches_timed": k1_n,
                "survey_5Bpp_gbs": round(survey_gbs, 1),
                "survey_5Bpp_frac": round(survey_gbs / HBM_PEAK_GBS, 4),
            },
            # threshold: the timed steps; contour / decode: the warm-up steps (every stage timed there, see above)
            "stage_ms_per_step": {"threshold": round(k1_avg_ms, 3), "contour": stage_ms.get("contour"), "decode": stage_ms.get("decode")},
            "stepping": "one context, synchronous" if args.no_pipeline else "two contexts on one stream: step i+1 submitted before step i is collected",
            "stats": stats,
            "frames_with_all_ids_correct": f"{id_ok}/{n}",
            "frame_synthesis_s": round(t_gen, 1),
        }
        if gathered is not None:
            out["gathered"] = gathered
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, d)
        if not args.no_other_workloads and world == 1:
            # free the headline batch first: the 4K batch below needs room only in the sense of tidiness (288 GB of HBM)
            out["other_workloads"] = other_workloads(local_rank, with_cpu=not args.no_cpu_baseline)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def pmc_traffic_bytes():
    """HBM bytes per K1 launch from the committed PMC passes of this same workload (tools/pmc_k1.sh ->
    profiles/<PROFILE_TAG>_pmc_bench_c2.json): (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for wide coalesced reads on gfx950.  Counters cannot be read inside this process, so this is the profiled value for the
    default 256-frame batch, or None when the summary is missing."""
    path = ROOT / "profiles" / f"{PROFILE_TAG}_pmc_bench_c2.json"
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


def other_workloads(device, with_cpu=True, budget_s=60.0):
    """The rest of BASELINE.json's configurations as driver-visible numbers, each on frames resident in HBM and with the
    single-thread oracle ("port") timed on a few of the same frames:
      C0  the reference's own bench recipe (benches/detect_markers.rs:29-51): uniform-noise RGB at 1920x1080, ARUCO
      C4  APRILTAG_36H11, 1280x720, +-15 degrees, Gaussian noise sigma 8
      C5  3840x2160, 16 markers, detect + IPPE pose in one call (a3_detect_batch_pose)
    One context, synchronous calls (a3_detect_batch), median of `reps` calls after two warm-up calls."""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    t_start = time.perf_counter()
    dev = torch.device("cuda", device)
    res = {}

    def run(name, frames_dev, dname, pose_mm=None, reps=7, cpu_frames=2, truths=None, note=""):
        if time.perf_counter() - t_start > budget_s:
            res[name] = {"skipped": "time budget"}
            return
        d = ARDictionary.new_from_named_dict(dname)
        ctx = Detector(DetectorConfig.default(), d, device=device)._context()
        n, h, w, c = frames_dev.shape
        a = (frames_dev.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)

        def call():
            if pose_mm:
                return ctx.detect_batch_pose(*a, pose_mm, None, n * 64)
            return ctx.detect_batch(*a, out_cap=n * 64)

        call(); call()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            r = call()
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        dt = sorted(ts)[len(ts) // 2]
        st = ctx.stats()
        o = {"value": round(n / dt, 1), "unit": "frames/s", "ms_per_batch": round(dt * 1e3, 3), "frames_per_batch": n,
             "resolution": [w, h], "dictionary": dname, "markers_found": int(len(r[0])), "darts_per_frame": int(st["darts"] // n),
             "borders_per_frame": int(st["contours_traced"] // n), "chunks": st["chunks"]}
        if note:
            o["workload"] = note
        if truths is not None:
            pos, ok = 0, 0
            for f in range(n):
                got = sorted(int(m["id"]) for m in r[0][pos: pos + int(r[1][f])]); pos += int(r[1][f])
                ok += got == sorted(t.id for t in truths[f])
            # recall of the reference ALGORITHM on this workload (the oracle finds the same: parity is what tests/ check)
            o["frames_with_all_drawn_ids_found"] = f"{ok}/{n}"
        if with_cpu:
            from oracle import a3oracle
            a3oracle.build()
            host = frames_dev[:cpu_frames].cpu().numpy()
            codes = np.ascontiguousarray(d.code_list)
            t0 = time.perf_counter()
            for f in range(cpu_frames):
                a3oracle.detect_markers_only(host[f], codes, d.num_bits, d._tau)
            o["cpu_baseline"] = {"value": round(cpu_frames / (time.perf_counter() - t0), 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                 "sample": f"{cpu_frames} of the same frames, single thread, detection only"}
        res[name] = o
        ctx.close()

    g = torch.Generator(device=dev); g.manual_seed(20261004)
    noise = torch.randint(0, 256, (32, 1080, 1920, 3), dtype=torch.uint8, device=dev, generator=g)
    run("C0_reference_bench_noise_1080p", noise, "ARUCO", cpu_frames=2,
        note="benches/detect_markers.rs:29-51 recipe: every channel of every pixel uniform random u8; no markers, ~1.6 M darts per frame")
    del noise
    spec4, name4 = synth.config_spec(4)
    d4 = ARDictionary.new_from_named_dict(name4)
    f4, t4 = synth.render_frames_device(spec4, d4.code_list, d4.num_bits, [synth.frame_seed(4, i) for i in range(32)], device=device)
    run("C4_apriltag36h11_720p_noise", f4, name4, cpu_frames=4, truths=t4, note="BASELINE config 4")
    del f4
    spec5, name5 = synth.config_spec(5)
    d5 = ARDictionary.new_from_named_dict(name5)
    f5, t5 = synth.render_frames_device(spec5, d5.code_list, d5.num_bits, [synth.frame_seed(5, i) for i in range(16)], device=device)
    run("C5_4k_16_markers_detect_plus_pose", f5, name5, pose_mm=40.0, cpu_frames=2, truths=t5,
        note="BASELINE config 5 on one GPU: detect + solve_with_undistorted_points of every marker in one call")
    return res


if __name__ == "__main__":
    main()
