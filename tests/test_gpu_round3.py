"""Round-3 GPU tests: `python bench.py --gpus N` started with no launcher, host-frame ingest (pageable / pinned, copy stream),
detect + pose in submit / collect form, the gather helper on a context that keeps its own stream, the candidate table growing
past 1024 quads per frame, a 3840x2160 noise frame (the reference bench's recipe at BASELINE config 5's size), CHILITAGS
rendered on the device.  Everything goes through the C ABI; the oracle is the checker.  GPU only."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.util import assert_frame_parity, bench_output, marker_tuples, markers_of_hip, markers_of_oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _detector(dicts, name="ARUCO", **cfg):
    from aruco3_amd.aruco import Detector, DetectorConfig

    return Detector(DetectorConfig(**cfg), dicts.new_from_named_dict(name))


def _args(frames, mem, ptr=None):
    from aruco3_amd import _lib

    n, h, w, c = frames.shape
    fmt = {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c]
    return (ptr if ptr is not None else frames.ctypes.data, mem, fmt, w, h, w * c, h * w * c, n)


# ------------------------------------------------------------------------------------------------------------------
# the driver's own command shape must start the ranks
# ------------------------------------------------------------------------------------------------------------------
def test_bench_front_door_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` typed as is -- no torch.distributed.run in the command, no RANK / WORLD_SIZE in the
    environment: the parent (which never touches the GPU) starts two fresh child ranks on the one leased GPU (gloo), relays
    rank 0's JSON line and exits 0."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "16", "--steps", "3", "--warmup", "1",
           "--device-synth", "--repeats", "2", "--no-other-workloads", "--no-cpu-baseline", "--launch-timeout", "500"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)                                                # ONE JSON line on stdout, within the driver's budget
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 3 and line["gathered"]["global_frame_indices_in_order"] is True
    assert line["dist"]["world_size"] == 2
    g = out["gathered"]       # the last collective: one rotation of four batches of 16 frames from each of the two ranks
    assert g["frames"] == 2 * 16 * g["batches_in_last_collective"] and g["global_frame_indices_in_order"] is True
    assert g["all_ranks_ids_correct"] >= 0.8 * g["frames"]
    assert out["config"]["distinct_batches_in_flight"] == 4
    assert out["dist"]["world_size"] == 2 and out["dist"]["backend"] == "gloo" and "self-launched" in out["dist"]["launcher"]


def test_bench_front_door_reports_a_failing_rank():
    """a rank that dies takes the launch down with a non-zero exit code (and no JSON line), instead of hanging its peers"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--backend", "no-such-backend", "--frames", "4", "--steps", "1", "--warmup", "0",
           "--device-synth", "--repeats", "1", "--no-other-workloads", "--no-cpu-baseline", "--launch-timeout", "300"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


# ------------------------------------------------------------------------------------------------------------------
# the caller's path -- host frames over the copy stream, pageable and pinned, pipelined
# ------------------------------------------------------------------------------------------------------------------
def test_host_frames_pageable_and_pinned_two_contexts(dicts, oracle):
    """Frames in host memory (pageable numpy, and a3_host_alloc'ed pinned memory) through a3_detect_batch and through
    submit / collect on two contexts -- the H2D of one batch on its context's copy stream while the other context's kernels
    run -- give the markers of the device-resident run and of the oracle."""
    import torch

    from aruco3_amd import _lib, synth

    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    frames_a, _ = synth.config_frames(1, 6)
    frames_b, _ = synth.config_frames(1, 6, first=6)
    dets = [_detector(dicts, "ARUCO_DEFAULT") for _ in range(2)]
    ctxs = [x._context() for x in dets]
    dev_a = torch.from_numpy(frames_a).cuda()
    want_a = ctxs[0].detect_batch(*_args(frames_a, _lib.MEM_DEVICE, dev_a.data_ptr()))
    for f in range(len(frames_a)):
        res = oracle.detect(frames_a[f], d.code_list, d.num_bits, d._tau)
        got = want_a[0][int(want_a[1][:f].sum()): int(want_a[1][: f + 1].sum())]
        assert markers_of_hip(got) == markers_of_oracle(res)
    dev_b = torch.from_numpy(frames_b).cuda()
    want_b = ctxs[0].detect_batch(*_args(frames_b, _lib.MEM_DEVICE, dev_b.data_ptr()))
    pin_a, pin_b = _lib.PinnedBuffer(frames_a.nbytes), _lib.PinnedBuffer(frames_b.nbytes)
    pin_a.array[:] = frames_a.reshape(-1); pin_b.array[:] = frames_b.reshape(-1)
    for label, pa, pb in (("pageable", frames_a.ctypes.data, frames_b.ctypes.data), ("pinned", pin_a.ptr, pin_b.ptr)):
        got = ctxs[0].detect_batch(*_args(frames_a, _lib.MEM_HOST, pa))
        assert marker_tuples(got[0]) == marker_tuples(want_a[0]) and np.array_equal(got[1], want_a[1]), label
        # pipelined: a, b, a, b on two contexts, each submitted before the previous one is collected
        seq = [(pa, frames_a, want_a), (pb, frames_b, want_b)] * 3
        ctxs[0].submit(*_args(seq[0][1], _lib.MEM_HOST, seq[0][0]))
        for i, (_, _, want) in enumerate(seq):
            if i + 1 < len(seq):
                ctxs[(i + 1) % 2].submit(*_args(seq[i + 1][1], _lib.MEM_HOST, seq[i + 1][0]))
            m, per = ctxs[i % 2].collect()
            assert marker_tuples(m) == marker_tuples(want[0]) and np.array_equal(per, want[1]), (label, i)
    # registering the caller's own buffer (a capture ring) pins it in place
    ring = np.ascontiguousarray(frames_a.copy())
    L = _lib.load()
    import ctypes as C
    assert L.a3_host_register(C.c_void_p(ring.ctypes.data), ring.nbytes) == 0
    try:
        got = ctxs[1].detect_batch(*_args(ring, _lib.MEM_HOST))
        assert marker_tuples(got[0]) == marker_tuples(want_a[0])
    finally:
        assert L.a3_host_unregister(C.c_void_p(ring.ctypes.data)) == 0
    pin_a.close(); pin_b.close()


def test_populated_detection_comes_back_in_batched_copies(dicts, oracle):
    """Detection.grey / .candidates / .homographies of every frame of a tapped batch (src/aruco.rs:115-120): the per-frame
    counts travel with the results, a frame's patches leave in one copy (it was one blocking copy per candidate) -- values
    equal the oracle's, through the host-side mirror of the reference API."""
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 4)
    det = _detector(dicts, "ARUCO_DEFAULT")
    d = det.dictionary
    outs = det.detect_batch(frames, populate=True)
    ctx = det._context()
    for f, o in enumerate(outs):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        assert np.array_equal(o.grey, res["grey"])
        assert [list(map(tuple, q)) for q in res["candidates"].tolist()] == o.candidates
        assert len(o.homographies) == len(res["homographies"]) > 0
        for p, q, ok in zip(o.homographies, res["homographies"], res["homography_ok"]):
            assert np.array_equal(p, q if ok else np.zeros((1, 1), np.uint8))
        assert_frame_parity(ctx, f, frames[f], res, frames.shape[2], frames.shape[1])


# ------------------------------------------------------------------------------------------------------------------
# pose in submit / collect form, the candidate table, 4K noise, CHILITAGS on the device
# ------------------------------------------------------------------------------------------------------------------
def test_pose_submit_collect_equals_the_synchronous_call(dicts, oracle):
    import torch

    from aruco3_amd import _lib, synth

    spec, name = synth.config_spec(5)
    d = dicts.new_from_named_dict(name)
    seeds = [synth.frame_seed(5, i) for i in range(4)]
    fa, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds[:2])
    fb, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds[2:])
    ctxs = [_detector(dicts, name)._context() for _ in range(2)]

    def a(t):
        n, h, w, c = t.shape
        return (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)

    want = [ctxs[0].detect_batch_pose(*a(t), 40.0) for t in (fa, fb)]
    assert len(want[0][0]) >= 20
    intr = _lib.Intrinsics(3840, 2160, 3000.0, 3000.0, 1920.0, 1080.0)
    want_i = ctxs[0].detect_batch_pose(*a(fa), 40.0, intr)
    ctxs[0].submit_pose(*a(fa), 40.0)
    ctxs[1].submit_pose(*a(fb), 40.0)
    got0 = ctxs[0].collect_pose()
    ctxs[0].submit_pose(*a(fa), 40.0, intr)           # the same context again while the other batch is still out
    got1 = ctxs[1].collect_pose()
    got2 = ctxs[0].collect_pose()
    for got, w in ((got0, want[0]), (got1, want[1]), (got2, want_i)):
        assert marker_tuples(got[0]) == marker_tuples(w[0]) and np.array_equal(got[1], w[1])
        assert np.array_equal(got[2].view(np.uint32), w[2].view(np.uint32))
    # against the oracle, first frame
    host = fa[0].cpu().numpy()
    res = oracle.detect(host, d.code_list, d.num_bits, d._tau)
    k = 0
    for mk in res["markers"]:
        ref = oracle.solve_with_undistorted_points(mk["corners"], 40.0, (3840, 2160))
        for (e, r, t), q in zip(ref, got0[2][k]):
            assert abs(float(q[0]) - e) <= 1e-4 and np.abs(q[1:10].reshape(3, 3) - np.asarray(r).reshape(3, 3)).max() <= 1e-4   # north_star: 1e-4
        k += 1
    # a plain collect on a pose batch is allowed (poses stay on the device for a3_pack_detections); the reverse is an error
    ctxs[0].submit(*a(fa))
    with pytest.raises(_lib.A3Error):
        ctxs[0].collect_pose()
    ctxs[0]._pending = (64 * 2, 2)


def test_gather_helper_orders_itself_on_the_contexts_own_stream(dicts):
    """shard.gather_detections_device on a context that keeps its OWN (non-blocking) stream -- no
    torch.cuda.synchronize() in between, no stream wrapping by the caller: the helper itself orders allocation, pack kernel,
    copy / collective.  World of one rank, gloo on the host and nccl (RCCL) on the device."""
    import torch
    import torch.distributed as dist

    from aruco3_amd import _lib, shard, synth

    frames, _ = synth.config_frames(1, 5)
    det = _detector(dicts, "ARUCO_DEFAULT")
    ctx = det._context()
    assert ctx.stream_ptr != 0                       # the context's own stream, not torch's
    dev = torch.device("cuda", 0)
    d_frames = torch.from_numpy(frames).to(dev)
    n, h, w, c = frames.shape
    a = (d_frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    for backend, coll in (("gloo", "cpu"), ("nccl", None)):
        if backend == "nccl":
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group(backend, rank=0, world_size=1)
        try:
            for rep in range(20):      # many times: a race would not lose every time
                markers, per = ctx.detect_batch(*a)
                junk = torch.full((4 << 20,), 0xEE, dtype=torch.uint8, device=dev)   # work on torch's current stream right before
                del junk
                out = shard.gather_detections_device(ctx, n, 100, dev, coll_device=coll)
                got = shard.unpack_detections(out.cpu().numpy().reshape(-1, out.shape[-1]))
                want = shard.unpack_detections(shard.pack_detections(markers, per, 100))
                assert [(f, marker_tuples(m)) for f, m in got] == [(f, marker_tuples(m)) for f, m in want], (backend, rep)
        finally:
            dist.destroy_process_group()
        os.environ["MASTER_PORT"] = str(port + 1)


def _grid_of_squares(w, h, side, pitch):
    img = np.full((h, w), 255, np.uint8)
    for y in range(pitch // 2, h - side - 8, pitch):
        for x in range(pitch // 2, w - side - 8, pitch):
            img[y: y + side, x: x + side] = 0
    return img


def test_candidate_table_grows_past_1024_quads_per_frame(dicts, oracle):
    """The reference's candidate list is a Vec (src/aruco.rs:124-166).  A frame tiled with small dark squares yields several
    thousand quads: the per-frame tables double and the batch is re-run (1024 -> 2048 -> 4096) instead of failing, and the
    candidates, their order and what discard_too_near leaves equal the oracle's."""
    img = _grid_of_squares(1600, 1200, 18, 30)
    frames = np.ascontiguousarray(img[None, :, :, None])
    det = _detector(dicts, "ARUCO")
    d = det.dictionary
    res = oracle.detect(img, d.code_list, d.num_bits, d._tau)
    assert 1024 < len(res["candidates_pre"]) <= 6144
    from tests.test_gpu_shard_taps import _detect_host

    ctx, m0, per0 = _detect_host(det, frames, taps=False)
    assert ctx.stats()["candidates_pre"] == len(res["candidates_pre"])
    assert ctx.stats()["candidates"] == len(res["candidates"])
    assert markers_of_hip(m0) == markers_of_oracle(res)
    ctx, m1, per1 = _detect_host(det, frames, taps=True)
    assert marker_tuples(m0) == marker_tuples(m1)
    assert_frame_parity(ctx, 0, img, res, 1600, 1200, check_patches=True)
    # a second, ordinary frame on the same context still works (tables stay large)
    from aruco3_amd import synth
    f1, _ = synth.config_frames(1, 1)
    ctx, m, per = _detect_host(det, f1, taps=False)
    assert markers_of_hip(m) == markers_of_oracle(oracle.detect(f1[0], d.code_list, d.num_bits, d._tau))


@pytest.mark.parametrize("w,h,side,pitch,factor,lo,hi", [(3200, 2400, 24, 32, 0.2, 6144, 12288), (3600, 3600, 12, 18, 0.004, 24576, 49152)])
def test_candidate_table_grows_past_the_lds_form(dicts, oracle, w, h, side, pitch, factor, lo, hi):
    """~7 000 and ~40 000 quads in one frame: past the 6144 slots that k_frame_candidates orders and thins in LDS (round 4's limit),
    into its through-memory form (tables of 12 288 / 49 152): candidates, their order, what discard_too_near leaves and the markers
    equal the oracle's."""
    from tests.test_gpu_shard_taps import _detect_host

    img = _grid_of_squares(w, h, side, pitch)
    det = _detector(dicts, "ARUCO", min_side_length_factor=factor)
    d = det.dictionary
    cfg = oracle.Config.default()
    cfg.min_side_length_factor = factor
    res = oracle.detect(img, d.code_list, d.num_bits, d._tau, cfg)
    assert lo < len(res["candidates_pre"]) <= hi
    ctx, m0, per0 = _detect_host(det, np.ascontiguousarray(img[None, :, :, None]), taps=False)
    st = ctx.stats()
    assert st["candidates_pre"] == len(res["candidates_pre"]) and st["candidates"] == len(res["candidates"])
    assert markers_of_hip(m0) == markers_of_oracle(res)
    assert ctx.candidates(0, before_discard=True).tolist() == res["candidates_pre"].tolist()
    assert ctx.candidates(0).tolist() == res["candidates"].tolist()


def test_more_candidates_than_the_limit_is_an_error_not_a_clip(dicts):
    """Beyond 65 536 quads in one frame (a3_marker.candidate_index is 16 bits) the call fails with A3_ERR_LIMIT -- after growing its
    tables seven times -- instead of returning a clipped list."""
    from aruco3_amd import _lib
    from tests.test_gpu_shard_taps import _detect_host

    img = _grid_of_squares(4800, 4800, 10, 16)          # ~ 89 000 squares
    det = _detector(dicts, "ARUCO", min_side_length_factor=0.004)   # (the default, 0.2 x the shorter side, would dismiss borders this short)
    with pytest.raises(_lib.A3Error) as e:
        _detect_host(det, np.ascontiguousarray(img[None, :, :, None]), taps=False)
    assert e.value.code == _lib.ERR_LIMIT


def test_uniform_noise_frame_3840x2160(dicts, oracle):
    """benches/detect_markers.rs:36-45's recipe at BASELINE config 5's size: ~6 M contour-graph nodes in one frame, 4x the
    largest noise frame tested before -- 32-bit dart indices, chunking under the device plan, the global entry rounds.
    Candidates (values and order), decode results and markers against the oracle; tapped and untapped runs agree."""
    from aruco3_amd import synth
    from tests.test_gpu_shard_taps import _detect_host

    det = _detector(dicts, "ARUCO")
    d = det.dictionary
    frames = np.stack([synth.noise_frame(3840, 2160, 31), synth.noise_frame(3840, 2160, 32)])
    ctx, m0, per0 = _detect_host(det, frames, taps=False)
    st = ctx.stats()
    assert st["darts"] > 8_000_000 and st["contours_traced"] > 800_000
    ctx2, m0b, per0b = _detect_host(det, frames, taps=False)       # second call: the device-side plan
    assert marker_tuples(m0) == marker_tuples(m0b)
    ctx, m1, per1 = _detect_host(det, frames, taps=True)
    assert marker_tuples(m0) == marker_tuples(m1)
    pos = 0
    for f in range(2):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        assert markers_of_hip(m0[pos: pos + int(per0[f])]) == markers_of_oracle(res)
        pos += int(per0[f])
        assert_frame_parity(ctx, f, frames[f], res, 3840, 2160)


def test_chilitags_rendered_on_the_device(dicts, oracle):
    """a3_synth_render with 10 x 10 cells (CHILITAGS, src/dictionaries.rs:154-156: 64 bits -> mark size 10; the cell bitmap is
    128 bits wide since ABI 3): what is drawn is what is read, and the HIP path equals the oracle on the rendered frames."""
    from aruco3_amd import synth
    from tests.test_gpu_shard_taps import _detect_host

    d = dicts.new_from_named_dict("CHILITAGS")
    spec = synth.SynthSpec(1280, 720, n_markers=(3, 3), side=(150.0, 220.0), min_center_sep=280.0, rotation_deg=(-25.0, 25.0))
    seeds = [7000 + i for i in range(4)]
    t, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds)
    frames = t.cpu().numpy()
    det = _detector(dicts, "CHILITAGS")
    ctx, m, per = _detect_host(det, frames, taps=False)
    pos, found = 0, 0
    for f in range(len(frames)):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        got = m[pos: pos + int(per[f])]; pos += int(per[f])
        assert markers_of_hip(got) == markers_of_oracle(res)
        found += len(set(int(x["id"]) for x in got) & set(tm.id for tm in truths[f]))
    assert found >= 8       # of the 12 drawn


# ------------------------------------------------------------------------------------------------------------------
# deferred decode (submit / collect): the decode stage of a submitted batch waits on the context's low-priority stream until
# another context submits behind it (released behind that batch's k_local_contract) or the batch is collected
# ------------------------------------------------------------------------------------------------------------------
def test_deferred_decode_in_every_order_of_calls(dicts):
    import torch

    from aruco3_amd import _lib, synth

    frames_a, _ = synth.config_frames(1, 5)
    frames_b, _ = synth.config_frames(1, 5, first=5)
    da, db = torch.from_numpy(frames_a).cuda(), torch.from_numpy(frames_b).cuda()
    ctxs = [_detector(dicts, "ARUCO_DEFAULT")._context() for _ in range(3)]
    aa, ab = _args(frames_a, _lib.MEM_DEVICE, da.data_ptr()), _args(frames_b, _lib.MEM_DEVICE, db.data_ptr())
    L = _lib.load()
    want = {}
    assert L.a3_debug_set_overlap(0) == 0
    want["a"], want["b"] = ctxs[0].detect_batch(*aa), ctxs[0].detect_batch(*ab)
    same = lambda got, w: marker_tuples(got[0]) == marker_tuples(w[0]) and np.array_equal(got[1], w[1])
    try:
        for mode in (0, 1, 2):
            assert L.a3_debug_set_overlap(mode) == 0
            # (1) collected in the order of submission, and in the opposite order
            ctxs[0].submit(*aa); ctxs[1].submit(*ab)
            assert same(ctxs[0].collect(), want["a"]) and same(ctxs[1].collect(), want["b"])
            ctxs[0].submit(*aa); ctxs[1].submit(*ab)
            assert same(ctxs[1].collect(), want["b"]) and same(ctxs[0].collect(), want["a"])
            # (2) a lone submit: nobody releases it, collect does
            ctxs[2].submit(*ab)
            assert same(ctxs[2].collect(), want["b"])
            # (3) a synchronous call on another context while a batch waits: it releases the waiting decode stage and is itself complete
            ctxs[0].submit(*aa)
            assert same(ctxs[1].detect_batch(*ab), want["b"])
            assert same(ctxs[0].collect(), want["a"])
            # (4) three in flight, with debug taps on one of them
            ctxs[1].set_debug_taps(True)
            ctxs[0].submit(*aa); ctxs[1].submit(*ab); ctxs[2].submit(*aa)
            assert same(ctxs[0].collect(), want["a"]) and same(ctxs[1].collect(), want["b"]) and same(ctxs[2].collect(), want["a"])
            assert ctxs[1].candidates(0).shape[0] > 0
            ctxs[1].set_debug_taps(False)
        # (5) a context destroyed with its submitted batch never collected
        extra = _detector(dicts, "ARUCO_DEFAULT")._context()
        extra.submit(*aa)
        extra.close()
        ctxs[0].submit(*aa)
        assert same(ctxs[0].collect(), want["a"])
    finally:
        L.a3_debug_set_overlap(-1)


def test_sampled_threshold_profiling_counts_every_fourth_batch(dicts):
    from aruco3_amd import _lib, synth

    frames, _ = synth.config_frames(1, 2)
    ctx = _detector(dicts, "ARUCO_DEFAULT")._context()
    a = _args(frames, _lib.MEM_HOST)
    ctx.set_profiling(_lib.PROFILE_THRESHOLD_SAMPLED)
    for _ in range(12):
        ctx.detect_batch(*a)
    ms, n = ctx.profile(_lib.STAGE_THRESHOLD)
    assert n == 3 and ms > 0.0
    assert ctx.profile(_lib.STAGE_DECODE)[1] == 0
    ctx.set_profiling(_lib.PROFILE_STAGES)
    ctx.detect_batch(*a)
    assert ctx.profile(_lib.STAGE_DECODE)[1] == 1 and ctx.profile(_lib.STAGE_THRESHOLD)[1] == 4


@pytest.mark.gpu
def test_contexts_driven_from_concurrent_host_threads(dicts):
    """include/aruco3_hip.h promises that contexts are independent: four host threads, a context each on the same device, frames of
    a different size each, 40 rounds of submit / collect (the decode stage of one thread's batch is released from inside another
    thread's submit, on the device's shared decode stream) mixed with synchronous calls and host-memory batches (shared copy
    stream).  Every result must equal the one a single thread got for the same frames."""
    import threading

    import torch

    from aruco3_amd import _lib, synth

    shapes = [(640, 480), (333, 251), (800, 600), (512, 512)]
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    jobs = []
    ref_ctx = _detector(dicts, "ARUCO_DEFAULT")._context()
    for t, (w, h) in enumerate(shapes):
        spec = synth.SynthSpec(w, h, n_markers=(2, 3), side=(min(w, h) * 0.2, min(w, h) * 0.3), min_center_sep=min(w, h) * 0.4)
        sets = []
        for k in range(3):
            fr = np.stack([synth.render_frame(spec, d.code_list, d.num_bits, 1000 * t + 10 * k + i)[0] for i in range(3)])
            dev = torch.from_numpy(fr).cuda()
            a_dev, a_host = _args(fr, _lib.MEM_DEVICE, dev.data_ptr()), _args(fr, _lib.MEM_HOST)
            m, per = ref_ctx.detect_batch(*a_dev)
            sets.append((fr, dev, a_dev, a_host, (marker_tuples(m), per.copy())))
        assert sum(len(s[4][0]) for s in sets) > 0
        jobs.append(sets)
    torch.cuda.synchronize()
    errors = []
    start = threading.Barrier(len(shapes))

    def worker(t):
        try:
            ctx = _detector(dicts, "ARUCO_DEFAULT")._context()
            start.wait()
            for it in range(40):
                fr, dev, a_dev, a_host, (want_m, want_per) = jobs[t][it % 3]
                how = (it + t) % 4
                if how == 0:
                    m, per = ctx.detect_batch(*a_dev)
                elif how == 1:
                    m, per = ctx.detect_batch(*a_host)
                else:
                    ctx.submit(*(a_dev if how == 2 else a_host))
                    m, per = ctx.collect()
                if marker_tuples(m) != want_m or not np.array_equal(per, want_per):
                    errors.append(f"thread {t}, round {it}, path {how}: {len(m)} markers, expected {len(want_m)}")
                    return
            ctx.close()
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(f"thread {t}: {e!r}")

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(len(shapes))]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a worker thread hangs"
    assert not errors, errors
