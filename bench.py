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
# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default), handed out in the order the
# streams are first used, and two streams that share a queue run in order whatever their events say -- a stream that merely waits
# holds up the others of its queue.  This process has four detection streams, the library's decode and copy streams, a side stream for
# the collective, the backend's internal stream and torch's explicit one: with 8 queues the side stream landed on a context's queue
# (tools/queue_probe.py, profiles/r05_queue_collisions.txt: every collective then stalled that context); with 16 none collide, and the
# one-GPU headline is the same with 8 and 16.  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
K1_BYTES_PER_PIXEL = 3.125         # what K1 has to move: 3 B RGB read + 1/8 B bit-packed binary written; NO grey plane is written
K1_SURVEY_BYTES_PER_PIXEL = 5      # SURVEY.md section 8d's figure (3 B read + 1 B grey + 1 B byte-wide binary): reported separately
E2E_BYTES_PER_PIXEL = 3.25         # what one step must move end to end: K1's 3.125 B/px + the 1/8 B/px re-read of the packed image
PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")   # profiles/<tag>_pmc_bench_c2.json holds the PMC passes of this same command (newest first)

# the workloads this file can step (BASELINE.json configs): c2 is the one the metric is quoted on and the default; c5 is the
# 3840x2160 detect + estimate_pose configuration (BASELINE config 5: `--workload c5 --gpus 4`)
WORKLOADS = {
    "c2": {"config": 2, "frames": 256, "pose_mm": None,
           "metric": "frames/sec at 1920x1080 ARUCO dict",
           "label": "BASELINE config 2: batch of {n} x 1920x1080 synthetic RGB frames per GPU, ARUCO dict, 4-8 markers per frame, frames resident "
                    "in HBM; Detector::detect end to end (grey, threshold, contours, quads, warp+decode, lookup) incl. D2H of the marker list"},
    "c5": {"config": 5, "frames": 64, "pose_mm": 40.0,   # (64 x 4K = the pixels of 256 x 1080p: the threshold kernel's strips are then as tall)
           "metric": "frames/sec at 3840x2160 ARUCO dict, detect + estimate_pose",
           "label": "BASELINE config 5: batch of {n} x 3840x2160 synthetic RGB frames per GPU, ARUCO dict, 16 markers per frame, frames resident "
                    "in HBM; Detector::detect + solve_with_undistorted_points of every marker (a3_detect_batch_pose_submit / _collect), "
                    "D2H of markers and pose pairs"},
}


def _render(args):
    from aruco3_amd import synth
    from aruco3_amd.dictionaries import ARDictionary

    config, idx = args
    spec, name = synth.config_spec(config)
    d = ARDictionary.new_from_named_dict(name)
    img, truth = synth.render_frame(spec, d.code_list, d.num_bits, synth.frame_seed(config, idx))
    return img, [t.id for t in truth]


def make_frames(config, first, count, workers):
    """Frames `first .. first+count` of a BASELINE config (seeded per frame index), rendered by a process pool on the host."""
    jobs = [(config, first + i) for i in range(count)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:
            res = pool.map(_render, jobs, chunksize=4)
    else:
        res = [_render(j) for j in jobs]
    frames = np.stack([r[0] for r in res])
    return frames, [r[1] for r in res]


def split_by_frame(markers, per):
    out, pos = [], 0
    for c in per:
        out.append(markers[pos: pos + int(c)]); pos += int(c)
    return out


def hip_marker_tuples(arr):
    return [(int(m["id"]), int(m["code"]), tuple(int(v) for v in m["corners"]), int(m["hamming_distance"]), int(m["rotation"])) for m in arr]


def oracle_marker_tuples(res):
    return [(m["id"], m["code"], tuple(v for c in m["corners"] for v in c), m["hamming_distance"], m["rotation"]) for m in res["markers"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=tuple(WORKLOADS), default="c2", help="c2 (default): BASELINE config 2, the configuration the metric is quoted "
                                                                                "on; c5: BASELINE config 5 (3840x2160, 16 markers, detect + pose)")
    ap.add_argument("--frames", type=int, default=0, help="frames per GPU per step (0 = the workload's: 256 for c2, 64 for c5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device-synth", action="store_true", help="render the frames on the GPU (a3_synth_render) instead of on the host: no "
                                                                  "host rendering, no H2D copy (same layouts and ids; pixels may differ by "
                                                                  "a grey level at cell edges)")
    ap.add_argument("--no-pipeline", action="store_true", help="one context, a3_detect_batch per step: every kernel runs alone (the GPU idles "
                                                                "while the host collects a batch)")
    ap.add_argument("--contexts", type=int, default=0, help="contexts in flight (submit / collect in rotation); 0 = 4 with --streams own, 2 with shared")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                       "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--synth-workers", type=int, default=0, help="host processes rendering frames (0 = auto)")
    ap.add_argument("--repeats", type=int, default=0, help="the timed region (exactly --steps steps between barrier + synchronize) is run "
                                                           "this many times back to back; the median is reported, every value listed "
                                                           "(20 steps are 13 ms: one region alone measures clocks ramping).  0 = as many as "
                                                           "make the timed regions total --min-timed-s seconds, at least 25")
    ap.add_argument("--min-timed-s", type=float, default=3.0, help="with --repeats 0: seconds the timed regions add up to (the GPU is busy that long)")
    ap.add_argument("--force-dist", action="store_true", help="run the multi-rank code path (process group, dictionary broadcast, device-packed "
                                                                "records, all-gather) even with one rank: lets a 1-GPU box exercise the RCCL branch")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip the other_workloads block (reference bench recipe, configs 4 and 5) and the "
                                                                       "shared-stream A/B")
    ap.add_argument("--streams", choices=("own", "shared"), default="own",
                    help="own (default): every context on a stream of its own -- the threshold kernels of consecutive batches serialise (each "
                         "fills the chip's register file), the contour / decode chains of the batches in flight overlap one another; "
                         "shared: all contexts enqueue on ONE stream (steps run in order; only the deferred decode stage overlaps)")
    ap.add_argument("--gates", choices=("burst", "none"), default="none",
                    help="with --streams own: none (default) = free-running rotation: batch i + N is submitted as soon as batch i is collected, the "
                         "hardware interleaves threshold kernels and chains as they come (with a batch of its own per context this is the fastest "
                         "arrangement measured, by 2 %%: profiles/r05_ab_streams.txt); burst = before each submit context k calls a3_order_after for the "
                         "contexts k+1 .. N-1: the threshold kernels of one rotation run back to back after the previous rotation's chains have "
                         "drained, the library holds the chains of all but the last member (the same process times it as `ab_burst_gates`)")
    ap.add_argument("--overlap", type=int, default=-1, help="measurement aid (a3_internal.h: a3_debug_set_overlap): force where the decode stage of "
                                                            "a submitted batch is released, 0 never deferred / 1 / 2, for every batch of the process; "
                                                            "-1 (default) = nothing is switched: the library decides per batch, as it does for any caller")
    ap.add_argument("--verify-gathers", action="store_true", help="N > 1 / --force-dist test aid: the batches change hands every rotation and EVERY "
                                                                     "collective's output is kept and checked at the end (global frame indices, ids)")
    ap.add_argument("--gather-delay-us", type=float, default=0.0, help="test aid: a sleep of this many microseconds on the side stream ahead of every "
                                                                         "collective (a collective that waits for lagging ranks)")
    ap.add_argument("--no-stream-probe", action="store_true", help="N > 1: skip the hardware-queue probe that keeps the contexts' streams off the "
                                                                     "collective's queues (aruco3_amd/streams.py)")
    ap.add_argument("--no-gather-backpressure", action="store_true", help="test aid: packs do NOT wait for the collective that last read their record "
                                                                             "buffer (round 4's behaviour: with --gather-delay-us records are overwritten before they are sent)")
    ap.add_argument("--isolated-launches", type=int, default=24, help="synchronous batches run one at a time, every stage between events, before the "
                                                                      "timed steps: the threshold kernel's launch duration ALONE (roofline) and the stage table")
    ap.add_argument("--max-markers", type=int, default=0, help="N > 1: markers per frame a gather record holds; 0 = calibrated on the first batch "
                                                               "(2 x the largest count on any rank, at least 8); a frame that holds more later is an error, never a clip")
    ap.add_argument("--launch-timeout", type=float, default=1500.0, help="N > 1 started without a launcher: seconds the parent waits for its ranks")
    ap.add_argument("--fail-rank", type=int, default=-1, help="test aid: this rank exits with code 3 once the process group is up (a rank that dies under "
                                                               "its peers: the launcher must end the launch with a non-zero code, not hang)")
    ap.add_argument("--frames-cache", default="", help="npz path: reuse rendered frames between runs (profiling runs use it so that "
                                                        "nothing forks under the profiler)")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    if args.frames <= 0:
        args.frames = wl["frames"]
    pose_mm = wl["pose_mm"]

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` typed as is: this process becomes the launcher.  It has not imported torch and never touches
        # HIP; it starts N FRESH child processes (one rank per GPU) and relays rank 0's JSON line.
        raise SystemExit(launch_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus must agree")

    own_streams = args.streams == "own" and not args.no_pipeline
    n_ctx = 1 if args.no_pipeline else (args.contexts or (4 if own_streams else 2))
    # Every context steps a batch of its OWN: n_ctx distinct batches are resident and in flight (a burst is four different batches in
    # any real use; four contexts re-reading one batch could share it in the 256 MiB Infinity Cache).  Batch j of this rank holds
    # the frames (rank * n_ctx + j) * frames .. + frames of the workload's seeded generator.
    n_bufs = n_ctx
    first_of = lambda j: (rank * n_bufs + j) * args.frames

    # host-side frame synthesis first (forks a pool; nothing has touched the GPU yet): batch 0 of the rank is rendered by the host
    # generator as in every round so far; the other batches in flight are rendered on the device below (same seeded layouts, a3_synth_render)
    workers = args.synth_workers or max(1, min(32, (os.cpu_count() or 8) // max(1, world)))
    t0 = time.time()
    cache = Path(f"{args.frames_cache}.{args.workload}.n{args.frames}.r{rank}.npz") if args.frames_cache else None
    frames0, truth0 = None, None
    if args.device_synth:
        pass                                    # every batch is rendered on the device, once it is set up
    elif cache is not None and cache.exists():
        z = np.load(cache, allow_pickle=True)
        frames0, truth0 = z["frames"], [list(t) for t in z["truth"]]
        assert frames0.shape[0] == args.frames
    else:
        frames0, truth0 = make_frames(wl["config"], first_of(0), args.frames, workers)
        if cache is not None:
            cache.parent.mkdir(parents=True, exist_ok=True)
            np.savez(cache, frames=frames0, truth=np.array(truth0, dtype=object))
    t_gen = time.time() - t0

    import torch
    import torch.distributed as dist

    from aruco3_amd import _lib, shard, synth
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
        if args.fail_rank == rank:
            os._exit(3)

    # The stepping below is what the LIBRARY does behind include/aruco3_hip.h: contexts on streams of their own + a3_order_after
    # gates -> bursts with held chains; contexts on one shared stream -> deferred decode.  No process-wide switch is set for the
    # headline (`library.internal_switches_used` lists what --overlap, a measurement aid, switched -- nothing by default).
    L = _lib.load()
    internal_switches_used = []
    if args.overlap >= 0:
        assert L.a3_debug_set_overlap(args.overlap) == 0
        internal_switches_used.append(f"a3_debug_set_overlap({args.overlap}) [--overlap]")
    spec, dict_name = synth.config_spec(wl["config"])
    d = ARDictionary.new_from_named_dict(dict_name) if rank == 0 or world == 1 else None
    if use_dist:
        d = shard.broadcast_dictionary(d, coll_dev, 0)   # RCCL broadcast, once
    dets = [Detector(DetectorConfig.default(), d, device=local_rank) for _ in range(n_ctx)]
    ctxs = [x._context() for x in dets]
    stream = torch.cuda.Stream(device=dev)   # an explicit stream: handle 0 (the default stream) would mean "the context's own"
    for ctx in ctxs:
        if not own_streams:
            ctx.set_stream(stream.cuda_stream)
    # the stream every context enqueues on, as a torch stream (for the waits on the pack of its previous batch)
    ctx_stream = {id(cx): (torch.cuda.ExternalStream(cx.stream_ptr, device=dev) if own_streams else stream) for cx in ctxs}
    ctx = ctxs[0]
    side = torch.cuda.Stream(device=dev)   # pack events + collective run beside the detection streams, not in them
    # N > 1: the collective's streams (the side stream it is issued from, the backend's internal one) must not share a HARDWARE queue
    # with a context's stream -- streams of one queue run in order, so a collective that waits for its peers would hold that context
    # up for as long as it waits (measured: aruco3_amd/streams.py, profiles/r05_queue_collisions.txt).  Probe, and move the contexts
    # onto streams that are held up neither by the collective nor by one another (public a3_set_stream; contexts on distinct
    # streams are stepped exactly like contexts on streams of their own).
    hw_queues = None
    if use_dist and own_streams and not args.no_stream_probe:
        from aruco3_amd import streams as a3streams

        tiny_in = torch.zeros(64, dtype=torch.uint8, device=coll_dev)
        tiny_out = torch.zeros(64 * world, dtype=torch.uint8, device=coll_dev)

        def collective_behind_a_sleep():
            with torch.cuda.stream(side):
                torch.cuda._sleep(int(1.5 * 2.4e6))
                if coll_dev.type == "cuda":
                    dist.all_gather_into_tensor(tiny_out, tiny_in)

        cands = [ctx_stream[id(cx)] for cx in ctxs] + [torch.cuda.Stream(device=dev) for _ in range(12)]
        chosen, hw_queues = a3streams.pick_streams(n_ctx, cands, [collective_behind_a_sleep], dev)
        hw_queues["own_streams_kept"] = sum(1 for c in chosen if any(c is ctx_stream[id(cx)] for cx in ctxs))
        taken = set()
        for cx in ctxs:            # a context whose own stream was chosen keeps it; the others take the remaining chosen streams
            if any(c is ctx_stream[id(cx)] for c in chosen):
                taken.add(id(ctx_stream[id(cx)]))
        spare = [c for c in chosen if id(c) not in taken]
        for cx in ctxs:
            if id(ctx_stream[id(cx)]) not in taken:
                st_new = spare.pop(0)
                cx.set_stream(st_new.cuda_stream)
                ctx_stream[id(cx)] = st_new
        torch.cuda.synchronize()

    t0 = time.time()
    d_bufs, truth_ids = [], []
    for j in range(n_bufs):
        if j == 0 and frames0 is not None:
            d_bufs.append(torch.from_numpy(frames0).to(dev)); truth_ids.append(truth0)      # inputs resident in HBM before the timed region
            continue
        seeds = [synth.frame_seed(wl["config"], first_of(j) + i) for i in range(args.frames)]
        df, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds, device=local_rank)
        d_bufs.append(df); truth_ids.append([[t.id for t in tr] for tr in truths])
    t_gen += time.time() - t0
    want_host = rank == 0 and not args.no_cpu_baseline and world == 1     # only the CPU baseline reads host copies
    frames_h = [frames0 if (j == 0 and frames0 is not None) else d_bufs[j].cpu().numpy() for j in range(n_bufs)] if want_host else None
    torch.cuda.synchronize()
    n, h, w, c = d_bufs[0].shape
    out_cap = n * 64

    buf_args = [(df.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n) for df in d_bufs]
    ctx_index = {id(cx): k for k, cx in enumerate(ctxs)}
    in_flight_buf = [0] * n_ctx         # the batch (index into d_bufs) context k has in flight / delivered last
    gstep = [0]                         # steps submitted since the process started

    def buf_for(k, rotation):
        # --verify-gathers: the batches change hands every rotation, so that the records of consecutive rotations differ and a record
        # overwritten before its collective has sent it shows (bench default: context k keeps batch k)
        return (k + rotation) % n_bufs if args.verify_gathers else k % n_bufs

    def detect_sync(cx, j=None):
        a = buf_args[ctx_index[id(cx)] % n_bufs if j is None else j]
        if pose_mm:
            return cx.detect_batch_pose(*a, pose_mm, None, out_cap)
        return cx.detect_batch(*a, out_cap=out_cap)

    def submit_raw(cx, j):
        if pose_mm:
            cx.submit_pose(*buf_args[j], pose_mm, None, out_cap)
        else:
            cx.submit(*buf_args[j], out_cap=out_cap)

    def collect_raw(cx):
        return cx.collect_pose() if pose_mm else cx.collect()

    # set-up, not steps: every context allocates its device buffers on its first batches (hipMalloc is slow and synchronous)
    for cx in ctxs:
        for _ in range(2):
            res0 = detect_sync(cx)

    # markers per frame a gather record holds: from the argument, or calibrated on the first batch over all ranks
    maxm = args.max_markers
    if use_dist and maxm <= 0:
        t = torch.tensor([int(res0[1].max()) if len(res0[1]) else 0], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        maxm = max(8, 2 * int(t[0]))

    last_gather = [None]
    pinned_rec = {}
    # Records of one ROTATION (n_ctx batches) are packed into one buffer and gathered by ONE collective: a collective per batch costs
    # the stepping loop ~0.1 ms of host time and a kernel's company per step; per rotation it is a quarter of that.  Two buffers in
    # turn: the collective of one rotation may still be reading while the next rotation's packs write -- and before a pack writes
    # into a buffer, its stream waits for the collective that last READ that buffer (two rotations earlier): `gather_done`.  A rank
    # whose peers lag therefore stalls on the device, behind its own collective, instead of overwriting records not yet sent.
    rec_bytes = shard.record_bytes(maxm, bool(pose_mm)) if use_dist else 0
    rec_bufs = [torch.empty((n_ctx, n, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(2)] if use_dist else None
    gather_out = [torch.empty((world * n_ctx * n, rec_bytes), dtype=torch.uint8, device=coll_dev) for _ in range(2)] if use_dist else None
    gather_done = [None, None]            # per record buffer: event behind the last collective that read it (recorded on `side`)
    rot = [0]
    gather_log = []                       # --verify-gathers: (clone of the collective's output, batches, [first global frame of every slot])
    slot_first = [0] * n_ctx
    slot_res = [None] * n_ctx             # --verify-gathers: what collect() handed the host for the batch packed into each slot

    pack_events = []
    timeline = []        # --verify-gathers: (kind, rotation, slot, event) in enqueue order

    def pack(cx, slot):
        # Per batch: fixed-capacity records written by a kernel from the device-resident marker list (a3_pack_detections); no host
        # copy in between.  The kernel (one wave per frame) goes onto the context's OWN stream, behind the batch just collected and
        # ahead of the context's next batch (which overwrites the marker list): no stream switching, no event to wait for before
        # the next submit.  An event behind it tells the side stream when the rotation's records are complete.
        b = rot[0] & 1
        if gather_done[b] is not None and not args.no_gather_backpressure:
            ctx_stream[id(cx)].wait_event(gather_done[b])     # write-after-read: the collective of rotation rot-2 read this buffer
        slot_first[slot] = first_of(in_flight_buf[ctx_index[id(cx)]])
        if args.verify_gathers:
            slot_res[slot] = last_res[ctx_index[id(cx)]][0]
        shard.pack_detections_device(cx, n, slot_first[slot], dev, maxm=maxm, with_poses=bool(pose_mm), out=rec_bufs[b][slot])
        ev = torch.cuda.Event(enable_timing=args.verify_gathers)
        ev.record(ctx_stream[id(cx)])
        pack_events.append(ev)
        if args.verify_gathers:
            timeline.append(("pack", rot[0], slot, ev))

    def all_gather(n_batches):
        b = rot[0] & 1
        rec = rec_bufs[b][:n_batches].view(n_batches * n, rec_bytes)
        for ev in pack_events:
            side.wait_event(ev)
        pack_events.clear()
        with torch.cuda.stream(side):
            if args.gather_delay_us > 0:   # test aid: a collective that waits (ranks out of step), stood in for by a sleep ahead of it
                torch.cuda._sleep(int(args.gather_delay_us * 2100))
            if coll_dev.type == "cpu":     # gloo rehearsal: host tensors, through a pinned buffer (a pageable D2H copy from a side
                key = tuple(rec.shape)     # stream stalls for tens of milliseconds under a busy detection stream on this runtime)
                if key not in pinned_rec:
                    pinned_rec[key] = torch.empty(rec.shape, dtype=rec.dtype, pin_memory=True)
                pinned_rec[key].copy_(rec, non_blocking=True)
                side.synchronize()
                rec = pinned_rec[key]
            out = gather_out[b][: world * n_batches * n]     # (read only on `side` -- the clone below -- or after a device-wide sync)
            last_gather[0] = (shard._all_gather(rec, n_batches * n, out=out), n_batches, list(slot_first[:n_batches]))
            if args.verify_gathers:
                gather_log.append((last_gather[0][0].clone(), n_batches, list(slot_first[:n_batches]), list(slot_res[:n_batches])))
            done = torch.cuda.Event(enable_timing=args.verify_gathers)
            done.record(side)
            gather_done[b] = done
            if args.verify_gathers:
                timeline.append(("gather", rot[0], n_batches, done))
        rot[0] += 1

    gated = own_streams and args.gates == "burst" and n_ctx > 1

    use_gates = [gated]

    def submit(cx):
        k = ctx_index[id(cx)]
        if use_gates[0]:   # bursts: this batch's threshold kernel starts once the previous rotation's chains (contexts k+1 ..) have drained
            for other in ctxs[k + 1:]:
                cx.order_after(other)
        j = buf_for(k, gstep[0] // n_ctx)
        in_flight_buf[k] = j
        gstep[0] += 1
        submit_raw(cx, j)

    last_res = {}     # context index -> (result, batch index) of its last collected batch

    def run_steps(k):
        """k steps; a step = one pass of Detector::detect over one batch of the rank's, results on the host (and, N > 1, packed on the
        device and all-gathered -- one collective per rotation of n_ctx batches)."""
        res = None
        if args.no_pipeline:
            for _ in range(k):
                res = detect_sync(ctx)
                last_res[0] = (res, 0)
                if use_dist:
                    pack(ctx, 0)
                    all_gather(1)
            return res
        # n_ctx batches ahead of the host: batch i + n_ctx goes out (on the context batch i has just been collected from) before
        # anything else happens, so the GPU always finds work queued however long the host takes over the results, the pack and
        # the collective.
        if gstep[0] % n_ctx:          # (a region always starts a rotation: the burst's last member is then the last context)
            gstep[0] += n_ctx - gstep[0] % n_ctx
        for i in range(min(n_ctx, k)):
            submit(ctxs[i % n_ctx])
        for i in range(k):
            cx = ctxs[i % n_ctx]
            res = collect_raw(cx)
            last_res[i % n_ctx] = (res, in_flight_buf[i % n_ctx])
            if use_dist:
                pack(cx, i % n_ctx)
            if i + n_ctx < k:
                submit(cx)
            if use_dist and (i % n_ctx == n_ctx - 1 or i == k - 1):
                all_gather(i % n_ctx + 1)
        return res

    # ---- isolated launches: one synchronous batch at a time, every stage between events, nothing else on the GPU ----
    # The roofline figure of the threshold kernel is its launch duration ALONE; in the stepping below no kernel runs alone.
    stage_ms, k1_ms, k1_n = {}, 0.0, 0
    for j in range(n_bufs):      # (the context meets every batch once before anything is timed: its pools grow to the largest)
        detect_sync(ctx, j)
    ctx.set_profiling(True)
    for st_id in (_lib.STAGE_THRESHOLD, _lib.STAGE_CONTOUR, _lib.STAGE_DECODE):
        ctx.profile(st_id, reset=True)
    torch.cuda.synchronize()
    for it in range(max(1, args.isolated_launches)):
        detect_sync(ctx, it % n_bufs)
    for name, st_id in (("threshold", _lib.STAGE_THRESHOLD), ("contour", _lib.STAGE_CONTOUR), ("decode", _lib.STAGE_DECODE)):
        a, b = ctx.profile(st_id, reset=True)
        stage_ms[name] = round(a / b, 4) if b else None
        if st_id == _lib.STAGE_THRESHOLD:
            k1_ms, k1_n = a, b
    ctx.set_profiling(0)

    res = run_steps(max(args.warmup, 0))
    # what the library did with the last batch of every context (a3_stats.stepping): one more rotation if the warm-up was shorter
    stepping_seen = None
    if not args.no_pipeline:
        if args.warmup < n_ctx:
            run_steps(n_ctx)
        stepping_seen = [cx.stats()["stepping"] for cx in ctxs]

    # The threshold kernel as the stepping runs it: ONE REAL ROTATION of the burst stepping (public calls only: the gates, the
    # submits, A3_PROFILE_THRESHOLD_ONLY = the library's own events around each threshold kernel on its context's stream), started
    # from an idle GPU behind a common start event so that the host's enqueue time stays outside.  The four threshold kernels --
    # each reading its OWN batch -- run back to back, the held chains behind the last of them: the longest of the four event
    # intervals is the span from the start of the first to the end of the last.  Launches in flight together refill each other's
    # retiring wave slots, which a lone launch -- sized to fill the chip in exactly one round -- cannot: the per-launch time here
    # is what the kernel costs inside a step, the isolated one above what it costs alone.
    k1_burst_ms = None
    if own_streams and n_ctx > 1 and args.workload in WORKLOADS:
        spans = []
        use_gates[0] = True
        for cx in ctxs:
            cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY)
        for rep in range(12):
            torch.cuda.synchronize()
            for cx in ctxs:
                cx.profile(_lib.STAGE_THRESHOLD, reset=True)
            s0 = torch.cuda.Event()
            with torch.cuda.stream(ctx_stream[id(ctxs[0])]):
                torch.cuda._sleep(4_000_000)        # ~2 ms: the rotation's submits are enqueued while it runs
            s0.record(ctx_stream[id(ctxs[0])])
            for cx in ctxs[1:]:
                ctx_stream[id(cx)].wait_event(s0)
            for cx in ctxs:
                submit(cx)
            for cx in ctxs:
                collect_raw(cx)
            per_ctx = [cx.profile(_lib.STAGE_THRESHOLD, reset=True) for cx in ctxs]
            if all(b == 1 for _, b in per_ctx):
                spans.append(max(a for a, _ in per_ctx) / n_ctx)
        for cx in ctxs:
            cx.set_profiling(0)
        use_gates[0] = gated
        spans = sorted(spans[2:])
        k1_burst_ms = spans[len(spans) // 2] if spans else None

    # The timed region: EXACTLY --steps steps between barrier + synchronize on both sides, max over ranks.  It is run
    # --repeats times back to back and the median region is the one reported (all are listed in ms_per_step_all).  No event is
    # recorded inside it.
    regions = []
    # Python's cyclic garbage collector is kept out of the timed regions (a full collection triggered by allocation count was the
    # one slow region of rounds 1 and 2); the steps allocate nothing cyclic.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()

    def one_region():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run_steps(args.steps)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        regions.append(time.perf_counter() - t0)
        return r

    n_regions = max(1, args.repeats)
    if args.repeats <= 0:
        # as many regions as make the timed time add up to --min-timed-s (every rank must run the same number: the count comes
        # from the slowest rank's first five regions)
        for _ in range(5):
            res = one_region()
        pilot = torch.tensor([sum(regions) / len(regions)], dtype=torch.float64, device=coll_dev if use_dist else "cpu")
        if use_dist:
            dist.all_reduce(pilot, op=dist.ReduceOp.MAX)
        n_regions = int(min(2000, max(25, np.ceil(args.min_timed_s / max(float(pilot[0]), 1e-6)))))
    while len(regions) < n_regions:
        res = one_region()
    if use_dist:
        t = torch.tensor(regions, dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        regions = [float(v) for v in t.tolist()]
    gc.enable()
    elapsed = sorted(regions)[len(regions) // 2]
    outliers = [{"region": i, "ms_per_step": round(r / args.steps * 1e3, 4)} for i, r in enumerate(regions) if r > 1.5 * elapsed]

    # sanity: what was rendered is what was read (ids per frame), on the last batch every context delivered
    id_ok, id_total = 0, 0
    by_frame_of, per_of, poses_of = {}, {}, {}
    for k, (r_k, j) in sorted(last_res.items()):
        bf = split_by_frame(r_k[0], r_k[1])
        by_frame_of[j], per_of[j] = bf, r_k[1]
        poses_of[j] = r_k[2] if pose_mm else None
        id_ok += sum(sorted(int(m["id"]) for m in bf[f]) == sorted(truth_ids[j][f]) for f in range(n))
        id_total += n
    markers, per = res[0], res[1]

    # the threshold kernel's launch duration IN COMPANY (a few more steps with its launches between events, outside the timed
    # regions): what the kernel trace of this run shows for it -- it waits for and shares the chip with the other batches
    k1_company = None
    if not args.no_pipeline:
        for cx in ctxs:
            cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY); cx.profile(_lib.STAGE_THRESHOLD, reset=True)
        run_steps(max(8, 2 * n_ctx))
        tot = [cx.profile(_lib.STAGE_THRESHOLD, reset=True) for cx in ctxs]
        k1_company = round(sum(a for a, _ in tot) / max(sum(b for _, b in tot), 1), 4)
        for cx in ctxs:
            cx.set_profiling(0)

    # ==== A/B block: other ways to step, same frames, same box, same minute.  Internal switches (a3_internal.h) may appear from here on;
    # ==== everything above this line calls the public header only (tests/test_bench_public_abi.py greps for it).
    ab_shared, ab_r04_default, ab_gates = None, None, None
    internal_probes = []
    if not use_dist and own_streams and not args.no_other_workloads:
        def timed(fn, k=20):
            fn(args.steps)
            sr = []
            for _ in range(k):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r2 = fn(args.steps)
                torch.cuda.synchronize(); sr.append(time.perf_counter() - t0)
            return sorted(sr)[len(sr) // 2], len(sr), r2

        try:
            # (1) all contexts on ONE stream, two of them: the library defers the decode stage of a submitted batch behind the next
            # batch's k_local_contract by itself (contexts that share a stream) -- round 3's headline; public calls only
            sctx = [Detector(DetectorConfig.default(), d, device=local_rank)._context() for _ in range(2)]
            for q, cx in enumerate(sctx):
                cx.set_stream(stream.cuda_stream)
                for _ in range(2):
                    detect_sync(cx, q % n_bufs)

            def shared_steps(k):
                for i in range(min(2, k)):
                    submit_raw(sctx[i % 2], i % n_bufs)
                r = None
                for i in range(k):
                    r = collect_raw(sctx[i % 2])
                    if i + 2 < k:
                        submit_raw(sctx[i % 2], (i + 2) % n_bufs)
                return r

            med, nr, r2 = timed(shared_steps)
            jlast = (args.steps - 1) % n_bufs
            ab_shared = {"value": round(args.frames * args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(med / args.steps * 1e3, 4),
                         "regions": nr, "same_markers": bool(jlast in per_of and len(r2[0]) == int(per_of[jlast].sum()) and np.array_equal(r2[1], per_of[jlast])),
                         "library_stepping_seen": sctx[0].stats()["stepping"],
                         "stepping": "--streams shared: two contexts on ONE stream, two batches ahead, decode stage of a submitted batch deferred "
                                     "behind the next batch's k_local_contract (round 3's headline stepping; the library does this by itself for "
                                     "contexts that share a stream)"}
            for cx in sctx:
                cx.close()
        except Exception as e:   # a side measurement must not take the line down
            ab_shared = {"error": repr(e)}
        try:
            # (1b) the other way to use a3_order_after (public calls only): the headline free-running -> burst gates + held chains, or vice versa
            use_gates[0] = not gated
            med, nr, r2 = timed(run_steps)
            ab_gates = {"value": round(args.frames * args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(med / args.steps * 1e3, 4), "regions": nr,
                        "library_stepping_seen": [cx.stats()["stepping"] for cx in ctxs],
                        "stepping": ("burst gates (a3_order_after before every submit: context k after the contexts k+1 .. N-1), chains held by the library"
                                     if not gated else "free-running rotation, no gates")}
        except Exception as e:
            ab_gates = {"error": repr(e)}
        finally:
            use_gates[0] = gated
        try:
            # (2) what a caller of the public header got until round 4: own streams + gates, but the library's process-wide default was
            # "decode deferred behind the next k_local_contract" and chains were never held (needs the internal switch now)
            assert L.a3_debug_set_overlap(2) == 0
            internal_probes.append("a3_debug_set_overlap(2) for ab_r04_library_default, reset to -1 afterwards")
            use_gates[0] = True
            med, nr, r2 = timed(run_steps)
            ab_r04_default = {"value": round(args.frames * args.steps / med, 2), "unit": "frames/s", "ms_per_step": round(med / args.steps * 1e3, 4),
                              "regions": nr, "library_stepping_seen": [cx.stats()["stepping"] for cx in ctxs],
                              "stepping": "the headline's calls (four contexts, own streams, a3_order_after gates) on round 4's library default: decode "
                                          "stage deferred onto the device-wide decode stream, no chain held"}
        except Exception as e:
            ab_r04_default = {"error": repr(e)}
        finally:
            use_gates[0] = gated
            L.a3_debug_set_overlap(args.overlap if args.overlap >= 0 else -1)
    stats = ctx.stats()

    # ---- the second kernel north_star names: k_decode (homography warp + Otsu + bit grid + dictionary lookup), timed ALONE like K1 ----
    # Its launch duration comes from the library's kernel probe (a3_debug_kernel_time re-runs the kernel on the work list the last
    # synchronous batch left behind: nothing else on the GPU); beside SURVEY 8(d)'s BYTES fraction it gets the fraction of what
    # bounds it -- the L2's REQUEST rate for its scattered taps: tools/micro/scatterbench issues the same tap pattern with
    # everything else taken away, in this run, on this box.
    roofline_warp = None
    if rank == 0 and not use_dist and not args.no_other_workloads and args.workload == "c2":
        try:
            detect_sync(ctx, 0)
            n_cand = int(ctx.stats()["candidates"])
            # k_decode alone (no k_projection: the product solves those in k_frame_candidates), the frames COLD as the pipeline meets them
            # (kernel 4 overwrites half a gigabyte before every run, outside the timed span -- as scatterbench sweeps 512 MB)
            dec_ms = ctx.debug_kernel_time(4, -5, 10)
            samp_ms = ctx.debug_kernel_time(4, -2, 10)         # the same launch stopped after the sampling loop
            dec_warm_ms = ctx.debug_kernel_time(3, -5, 10)     # (frames still in the caches from the run before: round 4's "71.5-74 us")
            internal_probes.append("a3_debug_kernel_time(decode) for roofline_warp")
            S = 49
            warp_bytes = (S * S * 4 + 32) * n_cand             # 9 604 B read upper bound + 32 B written per candidate
            ceil_us = scatter_ceiling_us(n_cand)
            roofline_warp = {
                "kernel": "k_decode<256,64> (49x49 bilinear warp of every candidate straight from the RGB frames, Otsu, triangle resize, bit grid, "
                          "four rotations, dictionary lookup)",
                "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": round(warp_bytes / (dec_ms * 1e-3) / 1e9, 1), "frac": round(warp_bytes / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "bytes_per_candidate": S * S * 4 + 32, "candidates_per_launch": n_cand, "avg_launch_ms": round(dec_ms, 4),
                "sampling_only_ms": round(samp_ms, 4), "dependent_tail_ms": round(dec_ms - samp_ms, 4), "avg_launch_ms_frames_warm": round(dec_warm_ms, 4),
                "timed_how": "a3_debug_kernel_time: the kernel re-run alone, 10 times, on the work list of a synchronous batch of this run, HIP events around it; "
                             "before every run 512 MB are overwritten so that the kernel finds the frames cold, as inside the pipeline",
                "traffic": pmc_decode_traffic_bytes()[0],
                "traffic_source": pmc_decode_traffic_bytes()[1] + " (committed PMC pass, not this run)",
                # what the kernel is actually bound by: 64-byte sector requests of scattered 12-byte taps
                "request_rate": None if ceil_us is None else {
                    "what": "the tap pattern's sector-request ceiling: tools/micro/scatterbench (same candidates per launch, same 8x8 sample "
                            "blocks, 4-byte reads at the tap addresses, nothing else), run by this process on this box",
                    "ceiling_us_per_launch": round(ceil_us, 1), "frac_of_ceiling_whole_kernel": round(ceil_us / (dec_ms * 1e3), 4),
                    "frac_of_ceiling_sampling_loop": round(ceil_us / (samp_ms * 1e3), 4),
                    "requests_per_candidate_pmc": 1225,
                    "note": "1225 sector requests per candidate = 3.15 M per launch / 2571 candidates (profiles/r04_pmc_decode.txt: L2 -> fabric read "
                            "requests of k_decode); the same pattern in the microbenchmark, so the ratio of times is the ratio of request rates"},
            }
        except Exception as e:   # a side measurement must not take the line down
            roofline_warp = {"error": repr(e)}

    gathered = None
    if use_dist:
        # what the collectives delivered, checked on rank 0: every rank's frames of the rotation, global indices as packed, ids as
        # rendered.  Default: the last collective; --verify-gathers: every collective of the run (cloned on the side stream).
        torch.cuda.synchronize()
        with_p = bool(pose_mm)
        truth_cache = {}

        def truth_of(gf):   # ids rendered into global frame gf (layout only: no pixels are painted for this)
            if gf not in truth_cache:
                truth_cache[gf] = sorted(t.id for t in synth.device_layout(spec, d.code_list, d.num_bits, [synth.frame_seed(wl["config"], gf)])[2][0])
            return truth_cache[gf]

        def check_gather(g_all, g_batches, firsts):
            """-> (records, indices as expected, records whose ids are the rendered ones, rank 0's records)"""
            g_np = g_all.cpu().numpy()      # [world, batches of the rotation * n, record]
            n_rec = idx_ok = ids_ok = 0
            mine = []
            for r in range(world):
                recs = shard.unpack_detections(np.ascontiguousarray(g_np[r]).reshape(-1, g_np.shape[-1]), with_poses=with_p)
                # rank r packed slot s with the batch whose first global frame is firsts[s] shifted by the rank's offset
                want = [f0 - first_of(0) + (r * n_bufs) * args.frames + f for f0 in firsts for f in range(n)]
                n_rec += len(recs)
                idx_ok += int([x[0] for x in recs] == want)
                ids_ok += sum(sorted(int(v) for v in x[1]["id"]) == truth_of(x[0]) for x in recs)
                if r == rank:
                    mine = recs
            return n_rec, idx_ok, ids_ok, mine

        if rank == 0:
            g_all, g_batches, g_firsts = last_gather[0]
            n_rec, idx_ok, ids_ok, mine = check_gather(g_all, g_batches, g_firsts)
            gathered = {"frames": n_rec, "global_frame_indices_in_order": idx_ok == world, "all_ranks_ids_correct": int(ids_ok),
                        "record_bytes": int(g_all.shape[-1]), "max_markers_per_record": int(maxm),
                        "max_markers_from": "--max-markers" if args.max_markers > 0 else "calibrated on the first batch (2 x the largest count on any rank, >= 8)",
                        "packed_on": "device (a3_pack_detections)", "collective": f"all_gather_into_tensor over {args.backend}, one per rotation of {n_ctx} batches",
                        "batches_in_last_collective": int(g_batches), "collectives": int(rot[0]),
                        "write_after_read_guard": "off (--no-gather-backpressure)" if args.no_gather_backpressure else
                                                  "a pack waits, on its context's stream, for the collective that last read its record buffer"}
            if args.verify_gathers:
                bad, trace = [], []
                for gi, (g_c, g_b, g_f, g_res) in enumerate(gather_log):
                    nr, io, ido, mine_c = check_gather(g_c, g_b, g_f)
                    # this rank's records against what collect() returned for the very batches packed into the rotation's slots
                    content_ok = len(mine_c) == g_b * n
                    for sl in range(g_b):
                        bf = split_by_frame(g_res[sl][0], g_res[sl][1])
                        for f in range(n):
                            x = mine_c[sl * n + f] if content_ok else None
                            content_ok = content_ok and [int(v) for v in x[1]["id"]] == [int(m["id"]) for m in bf[f]] \
                                and [tuple(int(v) for v in q) for q in x[1]["corners"]] == [tuple(int(v) for v in m["corners"]) for m in bf[f]]
                    trace.append({"expected_first_frames": [int(v) for v in g_f], "got_first_frames": [int(mine_c[sl * n][0]) for sl in range(g_b)] if len(mine_c) == g_b * n else None})
                    if io != world or not content_ok:
                        bad.append({"collective": gi, "indices_ok_ranks": io, "records_equal_collected_results": bool(content_ok), "records": nr})
                gathered["verified_collectives"] = len(gather_log)
                gathered["collectives_with_wrong_records"] = len(bad)
                gathered["first_wrong"] = bad[:3]
                gathered["trace"] = trace[:10]
                # when things finished on the device (ms after the first pack of the run's last 40 entries): a pack of rotation r+2 that
                # completes BEFORE the collective of rotation r is the write-after-read the guard forbids
                tl = timeline[-40:]
                gathered["timeline_ms"] = [(kind, int(r), int(sl), round(tl[0][3].elapsed_time(ev), 3)) for kind, r, sl, ev in tl]
                gathered["gather_delay_us"] = args.gather_delay_us
            if pose_mm:   # rank 0's own frames of the last batch came back as it produced them, poses included
                j_last = in_flight_buf[(g_batches - 1) % n_ctx]
                lo = first_of(j_last)
                sel = [x for x in mine if lo <= x[0] < lo + n]
                gp = np.concatenate([x[2] for x in sel]) if sel else np.zeros((0, 2, 13), np.float32)
                pz = poses_of.get(j_last)
                gathered["pose_pairs_gathered"] = int(sum(len(x[2]) for x in mine))
                gathered["rank0_poses_bit_equal_after_gather"] = bool(pz is not None and gp.shape == pz.shape and np.array_equal(gp.view(np.uint32), pz.view(np.uint32)))

    if rank == 0:
        total_frames = args.frames * world * args.steps
        value = total_frames / elapsed
        k1_avg_ms = k1_ms / max(k1_n, 1)
        k1_bytes = int(K1_BYTES_PER_PIXEL * w * h * args.frames)     # algorithmic bytes per launch = the bytes that move
        achieved = k1_bytes / (k1_avg_ms * 1e-3) / 1e9 if k1_avg_ms > 0 else 0.0
        survey_gbs = K1_SURVEY_BYTES_PER_PIXEL * w * h * args.frames / (k1_avg_ms * 1e-3) / 1e9 if k1_avg_ms > 0 else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        e2e_bytes = E2E_BYTES_PER_PIXEL * w * h * args.frames
        if args.no_pipeline:
            stepping = "one context, synchronous: every kernel runs alone"
        elif own_streams:
            stepping = (f"{n_ctx} contexts, each on a stream of its own and stepping a batch of its own, used in rotation, {n_ctx} batches ahead of the host (batch i + "
                        f"{n_ctx} is submitted as soon as batch i is collected).  One threshold launch is 2048 waves of 256 VGPRs = every register of the chip: nothing "
                        "co-runs with it; the contour / decode chains of the batches in flight overlap one another.  "
                        + ("BURSTS (a3_order_after: before each submit context k waits for the contexts k+1 .. N-1): the threshold kernels of a rotation run back to "
                           "back once the previous rotation's chains have drained; the library holds the chain of every burst member but the last behind its "
                           "threshold kernel and the last member's submit enqueues them all (the library's behaviour behind include/aruco3_hip.h, no switch)"
                           if gated else
                           "FREE-RUNNING rotation: no gates, no mode, nothing but submit and collect -- the hardware interleaves threshold kernels and chains as "
                           "they come.  With four DIFFERENT batches in flight this is the fastest arrangement measured (burst gates: `ab_burst_gates`, 2 % "
                           "slower; round 4's 437 k frames/s in bursts came from four contexts re-reading ONE batch out of the Infinity Cache)")
                        + " (`library_stepping_seen` is a3_stats.stepping of every context's last batch)")
        else:
            stepping = (f"{n_ctx} contexts on ONE stream, {n_ctx} batches ahead: steps run in order; the decode stage of a submitted batch runs on the "
                        "device's decode stream, released behind the next batch's k_local_contract")
        out = {
            "metric": wl["metric"],
            "value": round(value, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "repeats": len(regions),
            "ms_per_step_all": [round(r / args.steps * 1e3, 3) for r in regions],
            "ms_per_step_min_max": [round(min(regions) / args.steps * 1e3, 4), round(max(regions) / args.steps * 1e3, 4)],
            "outlier_regions": {"count": len(outliers), "threshold": "1.5 x median", "first": outliers[:8]},
            "timed_region_s_total": round(sum(regions), 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic" + (" (rendered on the device)" if args.device_synth else
                                   f" (batch 0 of {n_bufs} rendered by the host generator, the others by the same seeded layouts on the device)" if n_bufs > 1 else ""),
            "config": {
                "workload": wl["label"].format(n=args.frames),
                "frames_per_gpu": args.frames,
                "distinct_batches_in_flight": n_bufs if not args.no_pipeline else 1,
                "resolution": [int(w), int(h)],
                "dictionary": dict_name,
                "sharding": "frames by rank, no data-path collective; dictionary broadcast once, detections all-gathered per batch" if world > 1 else "single GPU",
            },
            "roofline": {
                "kernel": "k_grey_threshold7 (RGB->grey + 15x15 adaptive threshold; output bit-packed, no grey plane)",
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": pmc_traffic_bytes()[0] if args.workload == "c2" and args.frames == 256 else None,
                "traffic_source": (pmc_traffic_bytes()[1] + " (committed PMC pass of this command, not this run)") if args.workload == "c2" and args.frames == 256 else None,
                # 3.125 B/px: the frame is read once (3 B/px) and only the bit-packed binary image (1/8 B/px) is written; the
                # grey plane of SURVEY's 5 B/px accounting is never materialised (the decode stage re-derives the grey levels
                # it samples).  achieved / frac count the bytes that move; the 5 B/px figure is kept under its own name.
                "bytes_per_pixel": K1_BYTES_PER_PIXEL,
                "bytes_per_launch": k1_bytes,
                "avg_launch_ms": round(k1_avg_ms, 4),
                "launches_timed": k1_n,
                "timed_how": "HIP events around the kernel in dedicated synchronous batches of this run, one at a time, nothing else on the GPU "
                             "(--isolated-launches); in the stepping itself the kernel never runs alone",
                "avg_launch_ms_in_company": k1_company,
                # the same kernel as the stepping runs it: a burst's launches back to back on their own streams, nothing else on the GPU
                "in_burst": None if not k1_burst_ms else {
                    "launches_in_flight": n_ctx, "ms_per_launch": round(k1_burst_ms, 4),
                    "achieved": round(k1_bytes / (k1_burst_ms * 1e-3) / 1e9, 1), "frac": round(k1_bytes / (k1_burst_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "distinct_batches_read": n_bufs,
                    "timed_how": "one real rotation of the burst stepping from an idle GPU (public calls only: a3_order_after gates, submits, "
                                 "A3_PROFILE_THRESHOLD_ONLY), all streams released by one start event; the longest of the library's event intervals "
                                 "around the rotation's threshold kernels = start of the first to end of the last, divided by their number; median of 10 rotations"},
                "survey_5Bpp_gbs": round(survey_gbs, 1),
                "survey_5Bpp_frac": round(survey_gbs / HBM_PEAK_GBS, 4),
            },
            # bytes one step must move (3.125 B/px through K1 + the 1/8 B/px re-read of the packed image) over the step's time:
            # how far the PIPELINE is from the HBM bound, beside the kernel's own fraction above
            "e2e_frac": round(e2e_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "e2e_bytes_per_step": int(e2e_bytes),
            # every stage alone (the isolated launches); their sum is longer than a step, which overlaps them
            "stage_ms_per_step": {"threshold": stage_ms.get("threshold"), "contour": stage_ms.get("contour"), "decode": stage_ms.get("decode")},
            "stepping": stepping,
            "stepping_word": "synchronous" if args.no_pipeline else ("free-running" if own_streams and not gated else "burst-gates" if own_streams else "shared-stream"),
            "contexts": n_ctx,
            "streams": "one per context" if own_streams else "shared",
            "gates": "burst (a3_order_after)" if gated else "none",
            "library": dict(_lib.library_info(), internal_switches_used=internal_switches_used,
                            internal_probes_after_the_headline=internal_probes),
            "stats": stats,
            "frames_with_all_ids_correct": f"{id_ok}/{id_total}",
            "library_stepping_seen": stepping_seen,
            "frame_synthesis_s": round(t_gen, 1),
        }
        if roofline_warp is not None:
            out["roofline_warp"] = roofline_warp
        if ab_shared is not None:
            out["ab_shared_stream"] = ab_shared
        if ab_gates is not None:
            out["ab_burst_gates" if not gated else "ab_free_running"] = ab_gates
        if ab_r04_default is not None:
            out["ab_r04_library_default"] = ab_r04_default
        if gathered is not None:
            out["gathered"] = gathered
        if use_dist:   # what the ranks themselves saw
            out["dist"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                           "launcher": os.environ.get("A3_BENCH_LAUNCHER", "external (torch.distributed.run)" if "RANK" in os.environ else "none (one process, --force-dist)"),
                           "hw_queues": hw_queues,
                           "pack_and_collective": "records packed by a kernel on the context's own stream right after collect(); one all-gather per rotation on a side stream, behind the packs' events",
                           "note": "an N > 1 RCCL number exists only where the driver's multi-GPU node produced one; a 1-GPU box can run world_size 1 (nccl) or rehearse ranks over gloo"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"], out["parity_in_run"] = cpu_baseline(frames_h, d, by_frame_of, poses_of, per_of, pose_mm, (w, h))
        if not args.no_other_workloads and world == 1 and args.workload == "c2":
            try:
                out["other_workloads"] = other_workloads(local_rank, with_cpu=not args.no_cpu_baseline)
            except Exception as e:   # side measurements never take the line down
                out["other_workloads"] = {"error": repr(e)}
        emit(out)
    if use_dist:
        dist.destroy_process_group()


LINE_BUDGET = 4096       # bytes the LAST stdout line may take (the driver keeps a bounded tail of stdout: round 5's 22 KB line was lost)


def _short(v, n=96):
    return v if not isinstance(v, str) or len(v) <= n else v[: n - 1].rstrip() + "~"


def _pick(d, keys, n=96):
    return {k: _short(d[k], n) for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def compact_line(out, detail_path="bench_detail.json"):
    """The driver-facing line: the contract's keys, `roofline`, `roofline_warp`, `cpu_baseline`, `parity_in_run`, one-word `stepping` /
    `gates` and a few headline figures of the other BASELINE configurations -- everything else (per-region times, prose, A/B rows,
    the other workloads' tables) stays in `out`, which goes to bench_detail.json.  Never longer than LINE_BUDGET bytes: optional
    keys are dropped, least important first, until it fits."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "repeats", "timed_region_s_total",
                       "higher_is_better", "scaling", "dtype"))
    line["vs_baseline"] = out.get("vs_baseline")
    line["data"] = "synthetic"
    cfg = out.get("config", {})
    line["config"] = dict(_pick(cfg, ("frames_per_gpu", "distinct_batches_in_flight", "resolution", "dictionary")),
                          workload=_short(cfg.get("workload", ""), 110), sharding=_short(cfg.get("sharding", ""), 60))
    rf = out.get("roofline") or {}
    line["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_pixel", "bytes_per_launch",
                                  "avg_launch_ms", "launches_timed", "avg_launch_ms_in_company"), 100)
    line["roofline"]["kernel"] = _short(rf.get("kernel", ""), 60)
    rw = out.get("roofline_warp")
    if isinstance(rw, dict):
        if "error" in rw:
            line["roofline_warp"] = {"error": _short(rw["error"], 120)}
        else:
            line["roofline_warp"] = _pick(rw, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "avg_launch_ms",
                                               "candidates_per_launch", "bytes_per_candidate"), 100)
            line["roofline_warp"]["kernel"] = _short(rw.get("kernel", ""), 40)
            rr = rw.get("request_rate") or {}
            if rr:
                line["roofline_warp"]["frac_of_request_ceiling"] = rr.get("frac_of_ceiling_whole_kernel")
                line["roofline_warp"]["request_ceiling_us"] = rr.get("ceiling_us_per_launch")
    for k in ("e2e_frac", "stage_ms_per_step", "contexts", "gates"):
        if k in out:
            line[k] = out[k]
    line["stepping"] = out.get("stepping_word", "synchronous")
    if "cpu_baseline" in out and out["cpu_baseline"]:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample", "host_cores_available"), 150)
        if isinstance(cb.get("all_cores"), dict):
            line["cpu_baseline"]["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
    if out.get("parity_in_run"):
        line["parity_in_run"] = _pick(out["parity_in_run"], ("frames_compared", "frames_equal", "batches_covered", "pose_max_abs_diff", "against"), 60)
    line["frames_with_all_ids_correct"] = out.get("frames_with_all_ids_correct")
    if isinstance(out.get("library"), dict):
        line["library"] = _pick(out["library"], ("abi", "tuning_build", "internal_switches_used"))
    if isinstance(out.get("gathered"), dict):
        line["gathered"] = _pick(out["gathered"], ("frames", "global_frame_indices_in_order", "all_ranks_ids_correct", "collectives", "verified_collectives",
                                                   "collectives_with_wrong_records", "rank0_poses_bit_equal_after_gather"))
    if isinstance(out.get("dist"), dict):
        line["dist"] = _pick(out["dist"], ("backend", "world_size", "launcher"), 60)
    # the other BASELINE configurations: [synchronous frames/s, pipelined frames/s, frames equal to the oracle / frames compared]
    ow = out.get("other_workloads")
    if isinstance(ow, dict) and "error" not in ow:
        brief = {}
        for name, row in ow.items():
            if isinstance(row, dict) and "value" in row:
                par = row.get("parity_in_run") or {}
                brief[name] = [row["value"], (row.get("pipelined") or {}).get("value"), f"{par.get('frames_equal')}/{par.get('frames_compared')}"]
        c1 = ow.get("C1_single_frame_from_host") or {}
        if isinstance(c1.get("markers_only_pinned"), dict):
            brief["C1_one_frame_per_call_ms"] = c1["markers_only_pinned"].get("median_ms")
        ing = (ow.get("C2_from_host_frames") or {}).get("pinned")
        if isinstance(ing, dict):
            brief["C2_from_pinned_host_fps"] = ing.get("value")
        brief["columns"] = "frames/s synchronous, frames/s pipelined, frames equal to the oracle"
        line["other_workloads"] = brief
    elif isinstance(ow, dict):
        line["other_workloads"] = {"error": _short(ow["error"], 120)}
    for k in ("ab_shared_stream", "ab_burst_gates", "ab_free_running", "ab_r04_library_default"):
        if isinstance(out.get(k), dict) and "value" in out[k]:
            line.setdefault("ab_fps", {})[k[3:]] = out[k]["value"]
    line["detail"] = detail_path
    for drop in ("ab_fps", "library", "dist", "other_workloads", "frames_with_all_ids_correct", "gathered", "contexts", "stage_ms_per_step"):
        if len(json.dumps(line)) <= LINE_BUDGET:
            break
        line.pop(drop, None)
    assert len(json.dumps(line)) <= LINE_BUDGET, "bench line over budget"
    return line


def emit(out):
    """stdout carries exactly ONE line, the compact one.  Everything else goes to bench_detail.json (under gpurun_out/ when that
    exists, so that it travels back from a GPU box; A3_BENCH_DETAIL overrides the path) and, as one line that starts with
    `bench_detail `, to stderr."""
    path = Path(os.environ.get("A3_BENCH_DETAIL", "") or ((ROOT / "gpurun_out" if (ROOT / "gpurun_out").is_dir() else ROOT) / "bench_detail.json"))
    try:
        path.write_text(json.dumps(out, indent=1))
        rel = str(path.relative_to(ROOT)) if path.is_relative_to(ROOT) else str(path)
    except OSError:
        rel = "stderr only (bench_detail.json could not be written)"
    print("bench_detail " + json.dumps(out), file=sys.stderr, flush=True)
    print(json.dumps(compact_line(out, rel)), flush=True)


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
            return int((2.0 * k1["FETCH_SIZE"] + k1["WRITE_SIZE"]) * 1024), f"profiles/{tag}_pmc_bench_c2.json"
        except Exception:
            continue
    return None, "no PMC summary under profiles/"


def pmc_decode_traffic_bytes():
    """HBM-side bytes per k_decode launch from the committed PMC passes (tools/pmc_chain.sh -> profiles/<tag>_pmc_chain.json: the
    byte-accurate TCC_EA0_RDREQ_DRAM_32B / WRREQ_*_DRAM_32B counters), or None"""
    for tag in PROFILE_TAGS:
        try:
            pmc = json.loads((ROOT / "profiles" / f"{tag}_pmc_chain.json").read_text())
            k = next(v for name, v in pmc.items() if "k_decode" in name)
            return int((k["read_MB_exact"] + k["write_MB_exact"]) * 1e6), f"profiles/{tag}_pmc_chain.json"
        except Exception:
            continue
    return None, "no PMC summary under profiles/"


def scatter_ceiling_us(n_cand):
    """tools/micro/scatterbench (built by __graft_entry__.build()) as a child process: microseconds per launch of the decode stage's tap
    pattern with 4-byte reads (the request-rate probe), best of its two depths; None when the binary is missing"""
    import re
    import subprocess

    exe = ROOT / "tools" / "micro" / "scatterbench"
    if not exe.exists():
        return None
    try:
        p = subprocess.run([str(exe), str(int(n_cand))], capture_output=True, text=True, timeout=120)
        m = re.search(r"4-byte reads at the same addresses, KU 2 / 4: ([0-9.]+) / ([0-9.]+) us", p.stdout)
        return min(float(m.group(1)), float(m.group(2))) if m else None
    except Exception:
        return None


def parity_row(agree, total, what="markers (id, code, corners, rotation, hamming distance)"):
    return {"frames_compared": total, "frames_equal": agree, "summary": f"{agree}/{total} frames", "compared": what,
            "against": "oracle/a3_oracle.c (CPU restatement of the reference; the reference itself cannot be built here)"}


def cpu_baseline(frames_by_batch, d, gpu_by_frame=None, gpu_poses=None, gpu_per=None, pose_mm=None, image_size=None):
    """The CPU oracle (a restatement of the reference algorithm, NOT the Rust crate, which cannot be built here) on the
    same frames, one thread -- the reference's own execution model -- for about 10 s of CPU work.  Its output is not thrown
    away: the markers of every distinct frame it processed are compared with the GPU's for the same frame -> parity_in_run.
    `frames_by_batch`: the host copies of the batches the contexts step (one per context); gpu_by_frame / gpu_poses / gpu_per:
    dicts batch index -> the GPU's last result for that batch."""
    from oracle import a3oracle

    a3oracle.build()
    codes = np.ascontiguousarray(d.code_list)
    order = [(j, f) for f in range(len(frames_by_batch[0])) for j in range(len(frames_by_batch))     # (batches interleaved: a short budget samples them all)
             if gpu_by_frame is None or j in gpu_by_frame]
    done, t0 = 0, time.perf_counter()
    budget_s, max_calls = 10.0, 2 * len(order)
    oracle_markers = {}
    while done < max_calls:
        j, f = order[done % len(order)]
        r = a3oracle.detect(frames_by_batch[j][f], codes, d.num_bits, d._tau, keep_debug=False)
        oracle_markers.setdefault((j, f), r)
        done += 1
        if done >= min(32, len(order)) and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    h, w = frames_by_batch[0].shape[1:3]
    out = {"value": round(done / dt, 2), "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": f"{done} detect() calls over {len(oracle_markers)} distinct {w}x{h} frames of the {len(frames_by_batch)} batches in flight, single thread, "
                     "oracle/a3_oracle.c (gcc -O2)",
           "host_cores_available": os.cpu_count()}
    parity = None
    if gpu_by_frame is not None:
        agree, pose_dev = 0, 0.0
        pos = {j: np.concatenate([[0], np.cumsum(gpu_per[j])]).astype(np.int64) for j in gpu_per}
        for (j, f), r in oracle_markers.items():
            same = oracle_marker_tuples(r) == hip_marker_tuples(gpu_by_frame[j][f])
            if same and pose_mm:   # both IPPE solutions of every marker, against the oracle's solve_with_undistorted_points
                for k, m in enumerate(r["markers"]):
                    sols = a3oracle.solve_with_undistorted_points(m["corners"], pose_mm, image_size)
                    for q in range(2):
                        err, rot, tr = sols[q].as_tuple() if hasattr(sols[q], "as_tuple") else sols[q]
                        want = np.concatenate([[err], np.asarray(rot, np.float32).reshape(-1), np.asarray(tr, np.float32).reshape(-1)])
                        got = gpu_poses[j][pos[j][f] + k, q]
                        dev = float(np.nanmax(np.abs(want - got))) if not np.isnan(want).all() else 0.0
                        pose_dev = max(pose_dev, dev)
                same = pose_dev <= 1e-4
            agree += same
        parity = parity_row(agree, len(oracle_markers))
        parity["batches_covered"] = len({j for j, _ in oracle_markers})
        if pose_mm:
            parity["pose_max_abs_diff"] = pose_dev
            parity["compared"] += " + both IPPE poses of every marker within 1e-4"
    # the same oracle, frame-parallel over the host's cores (one frame per worker; the C call releases the
    # GIL).  Informational: the reference itself is single-threaded.
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(os.cpu_count() or 1, 64))
    flat = frames_by_batch[0]
    one = lambda f: a3oracle.detect_markers_only(flat[f % len(flat)], codes, d.num_bits, d._tau)
    done_mt, t0 = 0, time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as pool:
        while time.perf_counter() - t0 < 6.0:
            list(pool.map(one, range(done_mt, done_mt + 4 * workers)))
            done_mt += 4 * workers
    dt = time.perf_counter() - t0
    out["all_cores"] = {"value": round(done_mt / dt, 2), "unit": "frames/s", "cores": workers,
                        "sample": f"{done_mt} frames, one frame per worker thread"}
    return out, parity


def other_workloads(device, with_cpu=True, budget_s=100.0):
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

    def run(name, frames_dev, dname, pose_mm=None, reps=7, cpu_frames=None, truths=None, note="", window=7, more=()):
        # cpu_frames None: the oracle processes EVERY frame the row times (parity_in_run then covers what `value` covers)
        # `more`: further batches of the same workload (other seeds): the pipelined measurement gives every context a batch of its own,
        # like the headline (four contexts re-reading ONE batch share it in the Infinity Cache)
        if time.perf_counter() - t_start > budget_s:
            res[name] = {"skipped": "time budget"}
            return
        d = ARDictionary.new_from_named_dict(dname)
        ctx = Detector(DetectorConfig(threshold_window=window), d, device=device)._context()
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
        # the same workload stepped like the headline: four contexts on streams of their own in a free-running rotation, a batch each
        try:
            ring = [ctx] + [Detector(DetectorConfig(threshold_window=window), d, device=device)._context() for _ in range(3)]
            nr = len(ring)
            batches = [frames_dev] + list(more)
            args_of = [(b.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n) for b in batches]
            arg = lambda k: args_of[k % len(args_of)]

            def sub(i):
                cx = ring[i % nr]
                if pose_mm:
                    cx.submit_pose(*arg(i % nr), pose_mm, None, n * 64)
                else:
                    cx.submit(*arg(i % nr), out_cap=n * 64)

            col = (lambda cx: cx.collect_pose()) if pose_mm else (lambda cx: cx.collect())
            for q, cx in enumerate(ring):
                for bq in range(len(args_of)):      # (every batch once: pools grow to the largest before anything is timed)
                    cx.detect_batch(*args_of[bq], out_cap=n * 64)
                cx.detect_batch(*arg(q), out_cap=n * 64)
            k = 16
            best = None
            for _ in range(3):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for i in range(nr):
                    sub(i)
                for i in range(k):
                    rr = col(ring[i % nr])
                    if i + nr < k:
                        sub(i + nr)
                torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / k
                best = dtp if best is None else min(best, dtp)
            if len(args_of) == 1:
                assert len(rr[0]) == len(r[0])
            o["pipelined"] = {"value": round(n / best, 1), "unit": "frames/s", "ms_per_batch": round(best * 1e3, 3), "distinct_batches_in_flight": len(args_of),
                              "stepping": "four contexts on their own streams in a free-running rotation, four batches ahead, a batch of its own per context"}
            for cx in ring[1:]:
                cx.close()
        except Exception as e:   # a side measurement must not take the line down
            o["pipelined"] = {"error": repr(e)}
        if truths is not None:
            pos, ok = 0, 0
            for f in range(n):
                got = sorted(int(m["id"]) for m in r[0][pos: pos + int(r[1][f])]); pos += int(r[1][f])
                ok += got == sorted(t.id for t in truths[f])
            # recall of the reference ALGORITHM on this workload (the oracle finds the same: parity is what tests/ check)
            o["frames_with_all_drawn_ids_found"] = f"{ok}/{n}"
        # every stage alone on this workload (one more synchronous batch with events between the stages)
        try:
            ctx.set_profiling(True)
            for st_id in (0, 1, 2):
                ctx.profile(st_id, reset=True)
            call(); call()
            o["stage_ms_per_batch"] = {k2: round(ctx.profile(i2)[0] / max(ctx.profile(i2)[1], 1), 4) for i2, k2 in enumerate(("threshold", "contour", "decode"))}
            ctx.set_profiling(0)
        except Exception as e:
            o["stage_ms_per_batch"] = {"error": repr(e)}
        if with_cpu:
            from oracle import a3oracle
            a3oracle.build()
            if cpu_frames is None:
                cpu_frames = n
            host = frames_dev[:cpu_frames].cpu().numpy()
            codes = np.ascontiguousarray(d.code_list)
            ocfg = a3oracle.Config.default()
            ocfg.threshold_window = window
            # the first frames one at a time on one thread (the timed CPU baseline), the rest frame-parallel: the checker's verdict is
            # what matters for them, not its speed
            n_timed = min(cpu_frames, 2 if w * h > 1000000 else 4)
            t0 = time.perf_counter()
            ores = [a3oracle.detect(host[f], codes, d.num_bits, d._tau, config=ocfg, keep_debug=False) for f in range(n_timed)]
            dt_cpu = time.perf_counter() - t0
            if cpu_frames > n_timed:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=max(1, min(os.cpu_count() or 1, 32))) as pool:
                    ores += list(pool.map(lambda f: a3oracle.detect(host[f], codes, d.num_bits, d._tau, config=ocfg, keep_debug=False), range(n_timed, cpu_frames)))
            o["cpu_baseline"] = {"value": round(n_timed / dt_cpu, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                                 "sample": f"{n_timed} of the same frames, single thread, detection only (all {cpu_frames} frames of the batch go through the "
                                           "oracle for parity_in_run, the rest frame-parallel)"}
            gpu_frames = split_by_frame(r[0], r[1])
            o["parity_in_run"] = parity_row(sum(oracle_marker_tuples(ores[f]) == hip_marker_tuples(gpu_frames[f]) for f in range(cpu_frames)), cpu_frames)
        res[name] = o
        ctx.close()

    g = torch.Generator(device=dev); g.manual_seed(20261004)
    # the reference's bench matrix, benches/detect_markers.rs:29: 1920x1080, 1280x720, 960x540, 512x512 uniform noise
    for (nw, nh), nb in (((1920, 1080), 32), ((1280, 720), 32), ((960, 540), 32), ((512, 512), 32)):
        noise = torch.randint(0, 256, (nb, nh, nw, 3), dtype=torch.uint8, device=dev, generator=g)
        noise_more = [torch.randint(0, 256, (nb, nh, nw, 3), dtype=torch.uint8, device=dev, generator=g) for _ in range(3)]
        run(f"C0_reference_bench_noise_{nw}x{nh}" if (nw, nh) != (1920, 1080) else "C0_reference_bench_noise_1080p", noise, "ARUCO", more=noise_more,
            reps=5,
            note="benches/detect_markers.rs:29-51 recipe: every channel of every pixel uniform random u8; no markers")
        del noise, noise_more
    # other threshold windows on config 2's batch (src/aruco.rs:35,61: the reference's cost does not depend on the radius; here radii 1..7
    # run the register-resident kernel templated on the radius, 8..31 the fused ring kernel, larger ones a separable three-kernel path);
    # the headline's 256 frames, so that the rows compare with it
    spec2, name2 = synth.config_spec(2)
    d2 = ARDictionary.new_from_named_dict(name2)
    f2, t2 = synth.render_frames_device(spec2, d2.code_list, d2.num_bits, [synth.frame_seed(2, i) for i in range(256)], device=device)
    f2_more = [synth.render_frames_device(spec2, d2.code_list, d2.num_bits, [synth.frame_seed(2, 256 * j + i) for i in range(256)], device=device)[0] for j in (1, 2, 3)]
    for wnd in (3, 11, 21):
        run(f"C2_threshold_window_{wnd}", f2, name2, truths=t2, window=wnd, more=f2_more,
            note=f"BASELINE config 2's frames with DetectorConfig.threshold_window = {wnd} ({2 * wnd + 1} x {2 * wnd + 1})")
    del f2, f2_more
    spec4, name4 = synth.config_spec(4)
    d4 = ARDictionary.new_from_named_dict(name4)
    f4, t4 = synth.render_frames_device(spec4, d4.code_list, d4.num_bits, [synth.frame_seed(4, i) for i in range(32)], device=device)
    f4_more = [synth.render_frames_device(spec4, d4.code_list, d4.num_bits, [synth.frame_seed(4, 32 * j + i) for i in range(32)], device=device)[0] for j in (1, 2, 3)]
    run("C4_apriltag36h11_720p_noise", f4, name4, truths=t4, note="BASELINE config 4", more=f4_more)
    del f4, f4_more
    spec5, name5 = synth.config_spec(5)
    d5 = ARDictionary.new_from_named_dict(name5)
    f5, t5 = synth.render_frames_device(spec5, d5.code_list, d5.num_bits, [synth.frame_seed(5, i) for i in range(16)], device=device)
    f5_more = [synth.render_frames_device(spec5, d5.code_list, d5.num_bits, [synth.frame_seed(5, 16 * j + i) for i in range(16)], device=device)[0] for j in (1, 2, 3)]
    run("C5_4k_16_markers_detect_plus_pose", f5, name5, pose_mm=40.0, truths=t5, more=f5_more,
        note="BASELINE config 5 on one GPU: detect + solve_with_undistorted_points of every marker in one call")
    del f5, f5_more
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
        r_markers_only = r
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
        ores, t = timeit(lambda: a3oracle.detect(frames[0], codes, d.num_bits, d._tau, keep_debug=False), 20)
        out["cpu_baseline"] = dict(t, cores=1, kind="port", sample="the same frame, single thread, 20 calls (markers only)")
        out["parity_in_run"] = parity_row(int(oracle_marker_tuples(ores) == hip_marker_tuples(r_markers_only[0])), 1)
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
        r_noise = r
    if with_cpu:
        dn = ARDictionary.new_from_named_dict("ARUCO")
        codes_n = np.ascontiguousarray(dn.code_list)
        ores_n, t = timeit(lambda: a3oracle.detect(noise, codes_n, dn.num_bits, dn._tau, keep_debug=False), 3)
        row["cpu_baseline"] = dict(t, cores=1, kind="port", sample="the same frame, single thread, 3 calls")
        row["parity_in_run"] = parity_row(int(oracle_marker_tuples(ores_n) == hip_marker_tuples(r_noise[0])), 1)
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
