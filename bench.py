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
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), and two streams that
# share a queue run in order whatever their events say.  This process has the detection stream, the library's decode and copy
# streams, a side stream for the collective and -- for the free-running side measurement -- two more: with 4 queues some of them
# collide (measured: the two free-running contexts then run in lock-step).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

FRAMES_PER_GPU = 256
WIDTH, HEIGHT = 1920, 1080
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
K1_BYTES_PER_PIXEL = 3.125         # what K1 has to move: 3 B RGB read + 1/8 B bit-packed binary written; NO grey plane is written
K1_SURVEY_BYTES_PER_PIXEL = 5      # SURVEY.md section 8d's figure (3 B read + 1 B grey + 1 B byte-wide binary): reported separately
PROFILE_TAGS = ("r03", "r02")      # profiles/<tag>_pmc_bench_c2.json holds the PMC passes of this same command (newest first)


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
    ap.add_argument("--repeats", type=int, default=0, help="the timed region (exactly --steps steps between barrier + synchronize) is run "
                                                           "this many times back to back; the median is reported, every value listed "
                                                           "(20 steps are 16 ms: one region alone measures clocks ramping).  0 = as many as "
                                                           "make the timed regions total --min-timed-s seconds, at least 25")
    ap.add_argument("--min-timed-s", type=float, default=3.0, help="with --repeats 0: seconds the timed regions add up to (the GPU is busy that long)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path (process group, dictionary broadcast, device-packed "
                                                                "records, all-gather) even with one rank: lets a 1-GPU box exercise the RCCL branch")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other_workloads block (reference bench recipe, configs 4 and 5)")
    ap.add_argument("--streams", choices=("own", "shared"), default="shared",
                    help="shared: both contexts enqueue on ONE stream (steps run in order; only the deferred decode stage overlaps); "
                         "own: every context on a stream of its own (consecutive steps overlap wherever the GPU has room)")
    ap.add_argument("--overlap", type=int, default=-1, help="measurement aid (a3_internal.h: a3_debug_set_overlap): where the decode stage of a "
                                                            "submitted batch is released, 0 never deferred / 1 / 2; -1 = the library's default")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 started without a launcher: seconds the parent waits for its ranks")
    ap.add_argument("--frames-cache", default="", help="npz path: reuse rendered frames between runs (profiling runs use it so that "
                                                        "nothing forks under the profiler)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` typed as is: this process becomes the launcher.  It has not imported torch and never touches
        # HIP; it starts N FRESH child processes (one rank per GPU) and relays rank 0's JSON line.
        raise SystemExit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")

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

    if args.overlap >= 0:
        assert _lib.load().a3_debug_set_overlap(args.overlap) == 0
    d = ARDictionary.new_from_named_dict("ARUCO") if rank == 0 or world == 1 else None
    if use_dist:
        d = shard.broadcast_dictionary(d, coll_dev, 0)   # RCCL broadcast, once
    # two contexts on ONE stream: kernels of consecutive steps never overlap (K1 is timed alone), but the host enqueues step
    # i+1 while step i runs, so the GPU does not idle between steps
    dets = [Detector(DetectorConfig.default(), d, device=local_rank) for _ in range(1 if args.no_pipeline else 2)]
    ctxs = [x._context() for x in dets]
    stream = torch.cuda.Stream(device=dev)   # an explicit stream: handle 0 (the default stream) would mean "the context's own"
    own_streams = args.streams == "own" and not args.no_pipeline
    for ctx in ctxs:
        if not own_streams:
            ctx.set_stream(stream.cuda_stream)
        ctx.set_profiling(True)
    # the stream every context enqueues on, as a torch stream (for the waits on the pack of its previous batch)
    ctx_stream = {id(cx): (torch.cuda.ExternalStream(cx.stream_ptr, device=dev) if own_streams else stream) for cx in ctxs}
    ctx_stream_ptr = {id(cx): cx.stream_ptr for cx in ctxs}
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
    side = torch.cuda.Stream(device=dev)   # pack + collective run beside the detection stream, not in it
    pack_done = {}                         # context -> event after its pack: the context's next batch overwrites the marker list

    pinned_rec = {}

    def pack(cx):
        # Per batch: fixed-capacity records written by a kernel from the device-resident marker list (a3_pack_detections); no host
        # copy in between.  collect() has returned, so batch i is complete: its pack needs no ordering against the detection
        # stream and goes to the side stream AT ONCE, beside the kernels of the batches already submitted -- queued on the
        # detection stream it would start a whole step late.  Only the enqueue happens here; the collective follows
        # in all_gather(), after the next batch has been submitted.
        cx.set_stream(side.cuda_stream)
        try:
            with torch.cuda.stream(side):
                rec = shard.pack_detections_device(cx, n, first_frame, dev)
                pack_done[id(cx)] = side.record_event()
        finally:
            cx.set_stream(ctx_stream_ptr[id(cx)])
        return rec

    def all_gather(rec):
        with torch.cuda.stream(side):
            if coll_dev.type == "cpu":     # gloo rehearsal: host tensors, through a pinned buffer (a pageable D2H copy from a side
                key = tuple(rec.shape)     # stream stalls for tens of milliseconds under a busy detection stream on this runtime)
                if key not in pinned_rec:
                    pinned_rec[key] = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
                pinned_rec[key].copy_(rec, non_blocking=True)
                side.synchronize()
                rec = pinned_rec[key]
            last_gather[0] = shard._all_gather(rec, n)

    def submit(cx):
        ev = pack_done.pop(id(cx), None)
        if ev is not None:
            ctx_stream[id(cx)].wait_event(ev)        # the pack of this context's previous batch has read the marker list
        cx.submit(*batch_args, out_cap=n * 64)

    def run_steps(k):
        """k steps; a step = one pass of Detector::detect over the rank's batch, results on the host (and all-gathered)."""
        markers, per = None, None
        if args.no_pipeline:
            for _ in range(k):
                ev = pack_done.pop(id(ctx), None)
                if ev is not None:
                    ctx_stream[id(ctx)].wait_event(ev)
                markers, per = ctx.detect_batch(*batch_args, out_cap=n * 64)
                if use_dist:
                    all_gather(pack(ctx))
            return markers, per
        # Two batches ahead of the host: batch i+2 goes out (on the context batch i has just been collected from) before anything
        # else happens, so the GPU always finds the next threshold kernel queued when a contour stage ends, however long the host
        # takes over the results, the pack and the collective.
        for i in range(min(2, k)):
            submit(ctxs[i % 2])
        for i in range(k):
            cx = ctxs[i % 2]
            markers, per = cx.collect()
            rec = pack(cx) if use_dist else None
            if i + 2 < k:
                submit(cx)
            if use_dist:
                all_gather(rec)
        return markers, per

    # set-up, not steps: every context allocates its device buffers on its first batches (hipMalloc is slow and synchronous)
    for cx in ctxs:
        for _ in range(2):
            cx.detect_batch(*batch_args, out_cap=n * 64)
    for cx in ctxs:
        for st_id in (_lib.STAGE_THRESHOLD, _lib.STAGE_CONTOUR, _lib.STAGE_DECODE):
            cx.profile(st_id, reset=True)
    # Warm-up with every stage timed (the breakdown reported as stage_ms_per_step); the timed steps keep only the two event
    # records around the threshold kernel, on every 4th batch of a context -- the roofline figure must be measured live, but
    # each record between two kernels costs ~6 us of device time.
    markers, per = run_steps(args.warmup)
    stage_ms = {}
    for name, st_id in (("threshold", _lib.STAGE_THRESHOLD), ("contour", _lib.STAGE_CONTOUR), ("decode", _lib.STAGE_DECODE)):
        tot = cnt = 0
        for cx in ctxs:
            a, b = cx.profile(st_id, reset=True); tot += a; cnt += b
        stage_ms[name] = round(tot / cnt, 3) if cnt else None
    if args.warmup > 0:
        for cx in ctxs:
            cx.set_profiling(_lib.PROFILE_THRESHOLD_SAMPLED)   # the kernel of every 4th batch of a context is timed

    # The timed region: EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks.  It is run
    # --repeats times back to back and the median region is the one reported (all are listed in ms_per_step_all).
    regions, k1_region_ms = [], []
    # The one slow region that showed up at the same index in every run of rounds 1 and 2 (one region of ~60 ms among ~16 ms ones,
    # with a normal threshold-kernel time inside it): a full collection of Python's cyclic garbage collector, triggered by
    # allocation count -- host time, nothing of the detector's.  Collect once here and keep the collector out of the timed
    # regions (the steps allocate nothing cyclic).
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    def k1_totals():
        tot = cnt = 0
        for cx in ctxs:
            a, b = cx.profile(_lib.STAGE_THRESHOLD); tot += a; cnt += b
        return tot, cnt

    def one_region():
        k0 = k1_totals()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = run_steps(args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        regions.append(time.perf_counter() - t0)
        k1 = k1_totals()
        k1_region_ms.append((k1[0] - k0[0]) / max(k1[1] - k0[1], 1))
        return res

    n_regions = max(1, args.repeats)
    if args.repeats <= 0:
        # as many regions as make the timed time add up to --min-timed-s (every rank must run the same number: the count comes
        # from the slowest rank's first five regions)
        for _ in range(5):
            markers, per = one_region()
        pilot = torch.tensor([sum(regions) / len(regions)], dtype=torch.float64, device=coll_dev if use_dist else "cpu")
        if use_dist:
            dist.all_reduce(pilot, op=dist.ReduceOp.MAX)
        n_regions = int(min(2000, max(25, np.ceil(args.min_timed_s / max(float(pilot[0]), 1e-6)))))
    while len(regions) < n_regions:
        markers, per = one_region()
    if use_dist:
        t = torch.tensor(regions, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in t.tolist()]
    gc.enable()
    elapsed = sorted(regions)[len(regions) // 2]
    # regions far off the median (a one-off stall somewhere): counted and located, with the threshold kernel's own average in
    # that region beside it -- a normal kernel time there says the stall was not on the GPU's side of K1
    outliers = [{"region": i, "ms_per_step": round(r / args.steps * 1e3, 4), "k1_ms_in_region": round(k1_region_ms[i], 4)}
                for i, r in enumerate(regions) if r > 1.5 * elapsed]

    # sanity: what was rendered is what was read (ids per frame), on this rank's last step
    pos, id_ok = 0, 0
    for f in range(n):
        got = sorted(int(m["id"]) for m in markers[pos: pos + int(per[f])])
        pos += int(per[f])
        id_ok += got == sorted(truth_ids[f])

    k1_ms = k1_n = 0
    for cx in ctxs:
        a, b = cx.profile(_lib.STAGE_THRESHOLD); k1_ms += a; k1_n += b

    # The other way to step, measured beside the headline (one rank, both contexts on streams of their OWN, nothing deferred):
    # consecutive batches overlap wherever the GPU has room -- the threshold kernel of batch i+1 runs beside the contour and decode
    # stages of batch i.  More frames per second, but no kernel runs alone any more: the threshold kernel's launches then last
    # ~0.44-0.48 ms, of which only part is its own, and a roofline fraction computed from that duration would describe the
    # company, not the kernel.  The headline keeps the steps in order (K1 alone, timed alone) and this number is reported as what
    # it is.
    free_running = None
    if not use_dist and not args.no_pipeline and not own_streams and not args.no_other_workloads:   # (a side measurement like those)
        try:
            L = _lib.load()
            L.a3_debug_set_overlap(0)
            fctx = [Detector(DetectorConfig.default(), d, device=local_rank)._context() for _ in range(2)]   # (never given a stream: their own)
            for cx in fctx:
                cx.set_profiling(_lib.PROFILE_THRESHOLD_SAMPLED)
                for _ in range(2):
                    cx.detect_batch(*batch_args, out_cap=n * 64)

            def free_steps(k):
                for i in range(min(2, k)):
                    fctx[i % 2].submit(*batch_args, out_cap=n * 64)
                res = None
                for i in range(k):
                    res = fctx[i % 2].collect()
                    if i + 2 < k:
                        fctx[i % 2].submit(*batch_args, out_cap=n * 64)
                return res

            free_steps(args.steps)
            for cx in fctx:
                cx.profile(_lib.STAGE_THRESHOLD, reset=True)
            fr_regions = []
            for _ in range(20):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                m2, p2 = free_steps(args.steps)
                torch.cuda.synchronize(); fr_regions.append(time.perf_counter() - t0)
            fr = sorted(fr_regions)[len(fr_regions) // 2]
            fk = [cx.profile(_lib.STAGE_THRESHOLD) for cx in fctx]
            free_running = {"value": round(args.frames * args.steps / fr, 2), "unit": "frames/s", "ms_per_step": round(fr / args.steps * 1e3, 4),
                            "regions": len(fr_regions), "same_markers": bool(len(m2) == len(markers) and np.array_equal(p2, per)),
                            "threshold_kernel_ms_in_company": round(sum(a for a, _ in fk) / max(sum(b for _, b in fk), 1), 4),
                            "stepping": "two contexts, each on a stream of its own, two batches ahead, no deferred decode: steps overlap freely"}
            for cx in fctx:
                cx.close()
        except Exception as e:   # a side measurement must not take the line down
            free_running = {"error": repr(e)}
        finally:
            _lib.load().a3_debug_set_overlap(args.overlap if args.overlap >= 0 else 2)
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
            "ms_per_step_all": [round(r / args.steps * 1e3, 3) for r in regions],
            "ms_per_step_min_max": [round(min(regions) / args.steps * 1e3, 4), round(max(regions) / args.steps * 1e3, 4)],
            "outlier_regions": {"count": len(outliers), "threshold": "1.5 x median", "first": outliers[:8]},
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
            "stepping": "one context, synchronous" if args.no_pipeline else "two contexts on one stream, two batches ahead: step i+2 is submitted as soon as step i is collected; "
                        "the decode stage of a submitted batch runs on the device's decode stream, released behind the next batch's k_local_contract",
            "stats": stats,
            "frames_with_all_ids_correct": f"{id_ok}/{n}",
            "frame_synthesis_s": round(t_gen, 1),
        }
        if free_running is not None:
            out["free_running_streams"] = free_running
        if gathered is not None:
            out["gathered"] = gathered
        if use_dist:   # what the ranks themselves saw
            out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                           "launcher": os.environ.get("A3_BENCH_LAUNCHER", "external (torch.distributed.run)" if "RANK" in os.environ else "none (one process, --force-dist)"),
                           "pack_and_collective": "side stream, started when collect() returns (beside the next batch's kernels)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(frames, d)
        if not args.no_other_workloads and world == 1:
            # free the headline batch first: the 4K batch below needs room only in the sense of tidiness (288 GB of HBM)
            try:
                out["other_workloads"] = other_workloads(local_rank, with_cpu=not args.no_cpu_baseline)
            except Exception as e:   # side measurements never take the line down
                out["other_workloads"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def launch_ranks(args):
    """The front door for N > 1 without a launcher: start one child per rank -- `python bench.py <same flags>` with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set, exactly what torch.distributed.run would hand them -- wait for all
    of them, pass on rank 0's JSON line.  Returns the exit code: non-zero if any rank fails or the launch times out (the
    children that are still alive are then killed by PID).  The parent itself never initialises the GPU, and no process that
    has is ever re-executed."""
    import socket
    import subprocess

    n = args.gpus
    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()   # a free rendezvous port
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    base["A3_BENCH_LAUNCHER"] = "bench.py --gpus N (self-launched child ranks)"
    base.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:]
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout carries the JSON line; the other ranks' stdout goes to our stderr so that the line stays alone
        procs.append(subprocess.Popen(cmd, env=env, cwd=str(ROOT), stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # drain rank 0's pipe while we wait
    reader.start()
    deadline = time.time() + args.launch_timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if failed:      # one rank down: the others would wait for it at the next collective until their own timeout
            r, c = failed[0]
            print(f"bench.py: rank {r} exited with {c}", file=sys.stderr)
            rc = c if c > 0 else 1
            break
        if all(c == 0 for c in codes):
            break
        if time.time() > deadline:
            print(f"bench.py: launch of {n} ranks timed out after {args.launch_timeout} s", file=sys.stderr)
            rc = 124
            break
        time.sleep(0.05)
    for p in procs:
        if p.poll() is None:     # still running after a failure / timeout: kill exactly that PID
            p.kill()
        p.wait()
    reader.join(timeout=10)
    out0 = chunks[0] if chunks else ""
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    for ln in (out0 or "").splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if rc == 0 and not lines:
        print("bench.py: rank 0 printed no JSON line", file=sys.stderr)
        rc = 1
    if lines:
        print(lines[-1], flush=True)
    return rc


def pmc_traffic_bytes():
    """HBM bytes per K1 launch from the committed PMC passes of this same workload (tools/pmc_k1.sh ->
    profiles/<tag>_pmc_bench_c2.json): (2 x FETCH_SIZE + WRITE_SIZE) KiB, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for wide coalesced reads on gfx950.  Counters cannot be read inside this process, so this is the profiled value for the
    default 256-frame batch, or None when the summary is missing."""
    for tag in PROFILE_TAGS:
        try:
            pmc = json.loads((ROOT / "profiles" / f"{tag}_pmc_bench_c2.json").read_text())
            k1 = next(v for k, v in pmc.items() if "k_grey_threshold7" in k)
            return int((2.0 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024)
        except Exception:
            continue
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
        # the same workload stepped like the headline: two contexts, submit / collect two batches ahead (the decode stage of a
        # batch then runs beside the next batch's contour stage)
        try:
            ctx2 = Detector(DetectorConfig.default(), d, device=device)._context()
            pair = [ctx, ctx2]

            def sub(cx):
                if pose_mm:
                    cx.submit_pose(*a, pose_mm, None, n * 64)
                else:
                    cx.submit(*a, out_cap=n * 64)

            col = (lambda cx: cx.collect_pose()) if pose_mm else (lambda cx: cx.collect())
            ctx2.detect_batch(*a, out_cap=n * 64); ctx2.detect_batch(*a, out_cap=n * 64)
            k = 12
            best = None
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sub(pair[0]); sub(pair[1])
                for i in range(k):
                    rr = col(pair[i % 2])
                    if i + 2 < k:
                        sub(pair[i % 2])
                torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / k
                best = dtp if best is None else min(best, dtp)
            assert len(rr[0]) == len(r[0])
            o["pipelined"] = {"value": round(n / best, 1), "unit": "frames/s", "ms_per_batch": round(best * 1e3, 3),
                              "stepping": "two contexts, submit / collect, two batches ahead"}
            ctx2.close()
        except Exception as e:   # a side measurement must not take the line down
            o["pipelined"] = {"error": repr(e)}
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
    # the reference's bench matrix, benches/detect_markers.rs:29: 1920x1080, 1280x720, 960x540, 512x512 uniform noise
    for (nw, nh), nb in (((1920, 1080), 32), ((1280, 720), 32), ((960, 540), 32), ((512, 512), 32)):
        noise = torch.randint(0, 256, (nb, nh, nw, 3), dtype=torch.uint8, device=dev, generator=g)
        run(f"C0_reference_bench_noise_{nw}x{nh}" if (nw, nh) != (1920, 1080) else "C0_reference_bench_noise_1080p", noise, "ARUCO",
            cpu_frames=2 if nw >= 1280 else 4, reps=5,
            note="benches/detect_markers.rs:29-51 recipe: every channel of every pixel uniform random u8; no markers")
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
    del f5
    try:
        res["C1_single_frame_from_host"] = caller_latency(device, with_cpu)
        if time.perf_counter() - t_start < budget_s + 30.0:
            res["C2_from_host_frames"] = host_ingest(device)
    except Exception as e:   # the headline must not die with a side measurement
        res["caller_path_error"] = repr(e)
    return res


def caller_latency(device, with_cpu=True, calls=200):
    """BASELINE config 1 the way the reference's own caller meets it (benches/detect_markers.rs:25,48-50: one host image per
    `detect` call): ONE 640x480 frame with 4 ARUCO_DEFAULT markers in host memory through a3_detect_batch(A3_MEM_HOST, n = 1) --
    H2D copy, every kernel, marker read-back -- as a LATENCY, with the debug taps off (Detection.markers only) and on
    (Detection.grey / .candidates / .homographies filled like src/aruco.rs:115-120, each download included), beside the
    single-thread oracle on the same frame."""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames, truth = synth.config_frames(1, 1)
    d = ARDictionary.new_from_named_dict("ARUCO_DEFAULT")
    ctx = Detector(DetectorConfig.default(), d, device=device)._context()
    n, h, w, c = frames.shape
    pinned = _lib.PinnedBuffer(frames.nbytes)
    pinned.array[:] = frames.reshape(-1)
    out = {"workload": "BASELINE config 1: one 640x480 RGB frame, 4 ARUCO_DEFAULT markers, host memory, n = 1 per call", "calls": calls}

    def timeit(fn, k=calls):
        fn(); fn(); fn()
        ts = []
        for _ in range(k):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        ts.sort()
        return r, {"median_ms": round(ts[len(ts) // 2] * 1e3, 4), "p10_ms": round(ts[len(ts) // 10] * 1e3, 4), "p90_ms": round(ts[(9 * len(ts)) // 10] * 1e3, 4)}

    for label, ptr in (("pageable", frames.ctypes.data), ("pinned", pinned.ptr)):
        a = (ptr, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, 1)
        ctx.set_debug_taps(False)
        r, t = timeit(lambda: ctx.detect_batch(*a, out_cap=64))
        t["markers"] = int(len(r[0])); t["ids_correct"] = sorted(int(m["id"]) for m in r[0]) == sorted(x.id for x in truth[0])
        out[f"markers_only_{label}"] = t
        ctx.set_debug_taps(True)

        def populated():
            res = ctx.detect_batch(*a, out_cap=64)
            grey = ctx.download_grey(0, w, h); cand = ctx.candidates(0); hom = ctx.homographies(0)
            return res, grey, cand, hom

        r, t = timeit(populated, max(20, calls // 4))
        t["candidates"] = int(len(r[2])); t["patches"] = int(len(r[3][0]))
        out[f"detection_fully_populated_{label}"] = t
    ctx.set_debug_taps(False)
    if with_cpu:
        from oracle import a3oracle
        a3oracle.build()
        codes = np.ascontiguousarray(d.code_list)
        _, t = timeit(lambda: a3oracle.detect_markers_only(frames[0], codes, d.num_bits, d._tau), 20)
        out["cpu_baseline"] = dict(t, cores=1, kind="port", sample="the same frame, single thread, 20 calls (markers only)")
    pinned.close()
    # the reference's own bench, call for call (benches/detect_markers.rs:29-51): ONE 1920x1080 uniform-noise frame per detect()
    rng = np.random.default_rng(29)
    noise = rng.integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    ctx_n = Detector(DetectorConfig.default(), ARDictionary.new_from_named_dict("ARUCO"), device=device)._context()
    pin_n = _lib.PinnedBuffer(noise.nbytes)
    pin_n.array[:] = noise.reshape(-1)
    row = {"workload": "benches/detect_markers.rs recipe as it is called: one 1920x1080 uniform-noise RGB frame in host memory per call"}
    for label, ptr in (("pageable", noise.ctypes.data), ("pinned", pin_n.ptr)):
        a = (ptr, _lib.MEM_HOST, _lib.FMT_RGB8, 1920, 1080, 1920 * 3, 1080 * 1920 * 3, 1)
        r, t = timeit(lambda: ctx_n.detect_batch(*a, out_cap=64), max(20, calls // 4))
        t["markers"] = int(len(r[0]))
        row[f"markers_only_{label}"] = t
    if with_cpu:
        dn = ARDictionary.new_from_named_dict("ARUCO")
        codes_n = np.ascontiguousarray(dn.code_list)
        _, t = timeit(lambda: a3oracle.detect_markers_only(noise, codes_n, dn.num_bits, dn._tau), 3)
        row["cpu_baseline"] = dict(t, cores=1, kind="port", sample="the same frame, single thread, 3 calls")
    out["reference_bench_one_frame_per_call_1080p_noise"] = row
    pin_n.close(); ctx_n.close(); ctx.close()
    return out


def host_ingest(device, batches=6, frames_per_batch=256):
    """BASELINE config 2 from HOST frames (H2D inclusive; never the headline `value`): 256-frame batches of the 1080p workload in
    pageable and in pinned host memory, two contexts in submit / collect so that the copy of batch i+1 (each context's copy
    stream) runs under the kernels of batch i, beside the bare pinned-H2D rate of the same bytes on this box."""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    dev = torch.device("cuda", device)
    spec, name = synth.config_spec(2)
    d = ARDictionary.new_from_named_dict(name)
    seeds = [synth.frame_seed(2, i) for i in range(frames_per_batch)]
    d_frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds, device=device)
    n, h, w, c = d_frames.shape
    nbytes = d_frames.numel()
    pageable = d_frames.cpu().numpy()                      # ordinary host memory, as a caller's Vec<u8> would be
    pinned = _lib.PinnedBuffer(nbytes)
    pinned.array[:] = pageable.reshape(-1)
    # the link itself: hipMemcpyAsync of the same bytes from pinned memory, nothing else running
    t_pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    t_pin.copy_(torch.from_numpy(pageable.reshape(-1)))
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        dst.copy_(t_pin, non_blocking=True)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    link_s = sorted(ts)[1]
    out = {"workload": f"BASELINE config 2 from host memory: {frames_per_batch} x 1920x1080 RGB per batch, two contexts, submit / collect",
           "pinned_h2d_alone": {"GBps": round(nbytes / link_s / 1e9, 2), "frames_per_s": round(n / link_s, 1)}}
    del dst, t_pin
    ctxs = [Detector(DetectorConfig.default(), d, device=device)._context() for _ in range(2)]
    for label, ptr in (("pageable", pageable.ctypes.data), ("pinned", pinned.ptr)):
        a = (ptr, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
        for cx in ctxs:
            cx.detect_batch(*a, out_cap=n * 64)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctxs[0].submit(*a, out_cap=n * 64)
        for i in range(batches):
            if i + 1 < batches:
                ctxs[(i + 1) % 2].submit(*a, out_cap=n * 64)
            markers, per = ctxs[i % 2].collect()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out[label] = {"value": round(batches * n / dt, 1), "unit": "frames/s", "GBps_over_the_link": round(batches * nbytes / dt / 1e9, 2),
                      "fraction_of_pinned_h2d_alone": round((batches * n / dt) / (n / link_s), 3), "markers_last_batch": int(len(markers))}
    for cx in ctxs:
        cx.close()
    pinned.close()
    return out


if __name__ == "__main__":
    main()
