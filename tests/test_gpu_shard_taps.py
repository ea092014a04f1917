"""Round-2 GPU tests: the HIP path under the sharding code (BASELINE config 3 at size), the device-side gather records,
find_contours opened to inspection, the reference's helper vectors on the device, the patch tap past 32 frames and the
per-batch candidate counters.  Everything goes through the C ABI; the oracle is the checker.  GPU only."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.util import bench_output, marker_tuples, markers_of_hip, markers_of_oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _detector(dicts, name="ARUCO", **cfg):
    from aruco3_amd.aruco import Detector, DetectorConfig

    return Detector(DetectorConfig(**cfg), dicts.new_from_named_dict(name))


def _detect_host(det, frames, taps):
    from aruco3_amd import _lib

    ctx = det._context()
    ctx.set_debug_taps(taps)
    a = np.ascontiguousarray(frames)
    n, h, w, c = a.shape
    fmt = {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c]
    markers, per = ctx.detect_batch(a.ctypes.data, _lib.MEM_HOST, fmt, w, h, w * c, h * w * c, n)
    return ctx, markers, per


# ------------------------------------------------------------------------------------------------------------------
# warped patches used to be stored at frame * 1024 + k in a buffer of 32768 slots
# ------------------------------------------------------------------------------------------------------------------
def test_patch_tap_beyond_32_frames(dicts, oracle):
    """40 small frames, markers in the late ones, debug taps on: the markers equal the untapped run, and the patches of a
    frame past the 32nd come back and equal the oracle's (they used to be written past the end of the tap buffer)."""
    from aruco3_amd import synth

    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    det = _detector(dicts, "ARUCO_DEFAULT")
    spec = synth.SynthSpec(160, 120, n_markers=(1, 1), side=(50.0, 70.0), min_center_sep=60.0)
    frames = np.stack([synth.render_frame(spec, d.code_list, d.num_bits, 900 + i)[0] for i in range(40)])
    _, m0, per0 = _detect_host(det, frames, taps=False)
    ctx, m1, per1 = _detect_host(det, frames, taps=True)
    assert np.array_equal(per0, per1) and marker_tuples(m0) == marker_tuples(m1)
    assert int(per1[32:].sum()) > 0
    for f in (0, 31, 32, 35, 39):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        patches, ok, codes, dec = ctx.homographies(f)
        assert ok.tolist() == res["homography_ok"].tolist()
        assert np.array_equal(patches, res["homographies"]), f
        assert codes.tolist() == res["codes"].tolist()


def test_stats_carry_candidate_counts(dicts, oracle):
    """a3_stats.candidates_pre / .candidates (the per-stage counters are a cheap parity check)"""
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 3)
    det = _detector(dicts, "ARUCO_DEFAULT")
    d = det.dictionary
    for taps in (False, True):
        ctx, markers, per = _detect_host(det, frames, taps)
        st = ctx.stats()
        res = [oracle.detect(f, d.code_list, d.num_bits, d._tau) for f in frames]
        assert st["candidates_pre"] == sum(len(r["candidates_pre"]) for r in res) > 0
        assert st["candidates"] == sum(len(r["candidates"]) for r in res) > 0
        assert st["markers"] == sum(len(r["markers"]) for r in res) == len(markers)


# ------------------------------------------------------------------------------------------------------------------
# multi-GPU bookkeeping around real detections (BASELINE configs 3 and 5)
# ------------------------------------------------------------------------------------------------------------------
def test_device_packed_records_equal_host_packed(dicts):
    """a3_pack_detections (a kernel on the device-resident marker list) writes the records shard.pack_detections states on
    the host; a frame with more markers than the record holds is an error on both sides, never a clip."""
    import torch

    from aruco3_amd import _lib, shard, synth

    frames, _ = synth.config_frames(1, 5)
    det = _detector(dicts, "ARUCO_DEFAULT")
    ctx, markers, per = _detect_host(det, frames, taps=False)
    assert int(per.max()) >= 2
    dev = torch.device("cuda", 0)
    rec = shard.pack_detections_device(ctx, len(frames), 1000, dev)
    torch.cuda.synchronize()
    got = shard.unpack_detections(rec.cpu().numpy())
    want = shard.unpack_detections(shard.pack_detections(markers, per, 1000))
    assert [(f, marker_tuples(m)) for f, m in got] == [(f, marker_tuples(m)) for f, m in want]
    assert [f for f, _ in got] == list(range(1000, 1005))
    assert all(int(m["frame"][0]) == f for f, m in got if len(m))          # .frame carries the global index
    # unused slots are zero, so two ranks' gathers of equal detections are byte-identical
    body = rec.cpu().numpy()[:, 8:].reshape(len(frames), shard.MAXM, 56)
    for f in range(len(frames)):
        assert not body[f, int(per[f]):].any()
    with pytest.raises(shard.RecordOverflow):
        shard.pack_detections_device(ctx, len(frames), 0, dev, maxm=int(per.max()) - 1)
    with pytest.raises(shard.RecordOverflow):
        shard.pack_detections(markers, per, 0, maxm=int(per.max()) - 1)


def test_config3_2048_frames_in_8_rank_slices(dicts, oracle):
    """BASELINE config 3 at its full size on one GPU: 2048 frames of 1920x1080 (seeds 0..2047, rendered on the device) walked
    through shard.partition's 8 rank slices -> detect -> device-packed records -> concatenation (what the all-gather
    delivers) -> unpack.  Global frame indices must come out 0..2047 in order, slice 3 must be what config 2's generator
    gives for frames 768..1023, and a sample of 64 frames spread over all slices is compared with the oracle in full."""
    import torch

    from aruco3_amd import _lib, shard, synth

    spec, name = synth.config_spec(3)
    d = dicts.new_from_named_dict(name)
    det = _detector(dicts, name)
    ctx = det._context()
    ctx.set_debug_taps(False)
    dev = torch.device("cuda", 0)
    total, world = 2048, 8
    sample = {int(v) for v in np.linspace(0, total - 1, 64).astype(int)}
    gathered, truth_ids, kept_frames = [], {}, {}
    buf = None
    for rank in range(world):
        lo, hi = shard.partition(total, world, rank)
        assert hi - lo == 256
        seeds = [synth.frame_seed(3, i) for i in range(lo, hi)]
        buf, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds, out=buf)
        n, h, w, c = buf.shape
        markers, per = ctx.detect_batch(buf.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
        rec = shard.pack_detections_device(ctx, n, lo, dev)
        torch.cuda.synchronize()
        gathered.append(rec.cpu().numpy())
        # the host-side statement of the same records, from the markers the call returned
        assert np.array_equal(shard.pack_detections(markers, per, lo)[:, :8], gathered[-1][:, :8])
        for i in range(lo, hi):
            truth_ids[i] = sorted(t.id for t in truths[i - lo])
            if i in sample:
                kept_frames[i] = buf[i - lo].cpu().numpy()
    out = shard.unpack_detections(np.concatenate(gathered))
    assert [f for f, _ in out] == list(range(total))
    all_ids = 0
    for f, m in out:
        assert all(int(x) == f for x in m["frame"])
        all_ids += sorted(int(x) for x in m["id"]) == truth_ids[f]
        if f in sample:
            res = oracle.detect(kept_frames[f], d.code_list, d.num_bits, d._tau)
            assert markers_of_hip(m) == markers_of_oracle(res), f
    assert all_ids >= int(0.9 * total)     # the rest lose a marker to the reference's quirks Q2/Q3, in the oracle too


def test_config5_4k_pose_in_4_rank_slices(dicts, oracle):
    """BASELINE config 5 (3840x2160, 16 markers, detect + IPPE pose, 4 GPUs) on one GPU: 8 frames walked through the 4 rank
    slices of shard.partition -> a3_detect_batch_pose -> device-packed records that carry both poses of every marker ->
    concatenation -> unpack.  Markers against the oracle bit for bit, poses within 1e-4 (BASELINE.json's tolerance), and the
    device records equal the host statement of the format."""
    import torch

    from aruco3_amd import _lib, shard, synth

    spec, name = synth.config_spec(5)
    d = dicts.new_from_named_dict(name)
    det = _detector(dicts, name)
    ctx = det._context()
    ctx.set_debug_taps(False)
    dev = torch.device("cuda", 0)
    total, world = 8, 4
    gathered, host_frames = [], {}
    for rank in range(world):
        lo, hi = shard.partition(total, world, rank)
        frames, _ = synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(5, i) for i in range(lo, hi)])
        n, h, w, c = frames.shape
        markers, per, poses = ctx.detect_batch_pose(frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n, 40.0)
        rec = shard.pack_detections_device(ctx, n, lo, dev, with_poses=True)
        torch.cuda.synchronize()
        got = rec.cpu().numpy()
        want = shard.pack_detections(markers, per, lo, poses=poses)
        assert got.shape == want.shape == (n, shard.record_bytes(with_poses=True))
        a, b = shard.unpack_detections(got, with_poses=True), shard.unpack_detections(want, with_poses=True)
        assert [(f, marker_tuples(m)) for f, m, _ in a] == [(f, marker_tuples(m)) for f, m, _ in b]
        assert all(np.array_equal(pa.view(np.uint32), pb.view(np.uint32)) for (_, _, pa), (_, _, pb) in zip(a, b))
        gathered.append(got)
        for i in range(lo, hi):
            host_frames[i] = frames[i - lo].cpu().numpy()
        with pytest.raises(_lib.A3Error):      # poses are only there after a detect_batch_pose call
            ctx.detect_batch(frames.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
            shard.pack_detections_device(ctx, n, lo, dev, with_poses=True)
    out = shard.unpack_detections(np.concatenate(gathered), with_poses=True)
    assert [f for f, _, _ in out] == list(range(total))
    n_markers = 0
    for f, m, poses in out:
        res = oracle.detect(host_frames[f], d.code_list, d.num_bits, d._tau)
        assert markers_of_hip(m) == markers_of_oracle(res), f
        for k, mk in enumerate(res["markers"]):
            ref = oracle.solve_with_undistorted_points(mk["corners"], 40.0, (3840, 2160))      # ((error, rotation, translation), (..))
            for (e, r, t), q in zip(ref, poses[k]):
                assert abs(float(q[0]) - e) <= 1e-4 and np.abs(q[1:10].reshape(3, 3) - np.asarray(r).reshape(3, 3)).max() <= 1e-4
                assert np.abs(q[10:13] - np.asarray(t).reshape(3)).max() <= 1e-4 * max(1.0, float(np.abs(t).max()))
        n_markers += len(m)
    assert n_markers >= 100


def test_two_rank_rehearsal_as_child_processes():
    """bench.py --gpus 2 --backend gloo: two ranks started as FRESH child processes (this process has touched the GPU and is
    never re-executed), both on the one leased GPU: partition, dictionary broadcast, detect, device-packed records,
    all-gather, barrier + max-over-ranks timing.  The JSON line must carry both ranks' frames and correct ids."""
    import socket

    sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()   # a free rendezvous port
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "16",
           "--backend", "gloo", "--device-synth", "--repeats", "1", "--no-other-workloads"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)
    assert line["n_gpus"] == 2 and line["frames_with_all_ids_correct"] == out["frames_with_all_ids_correct"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    g = out["gathered"]       # the last collective: one rotation of up to four batches of 16 frames from each of the two ranks
    assert g["frames"] == 2 * 16 * g["batches_in_last_collective"] and g["global_frame_indices_in_order"] is True
    ok, n = (int(v) for v in out["frames_with_all_ids_correct"].split("/"))
    assert n == 64 and ok >= 52       # four distinct batches per rank
    assert g["all_ranks_ids_correct"] >= 0.8 * g["frames"]


# ------------------------------------------------------------------------------------------------------------------
# row E opened to inspection: find_contours itself, not just the quads that survive it
# ------------------------------------------------------------------------------------------------------------------
def _isolated(img, c):
    h, w = img.shape
    if len(c) != 1:
        return False
    x, y = (int(v) for v in c[0])
    return not any((dx or dy) and 0 <= x + dx < w and 0 <= y + dy < h and img[y + dy, x + dx]
                   for dy in (-1, 0, 1) for dx in (-1, 0, 1))


def _compare_contours(det, oracle, grey_frames):
    """every border of every frame: count, discovery order, start pixel + border type, point sequence"""
    frames = np.ascontiguousarray(grey_frames)[..., None]
    ctx, _, _ = _detect_host(det, frames, taps=True)
    n_total = 0
    for f in range(frames.shape[0]):
        img = frames[f, ..., 0]
        h, w = img.shape
        binary = oracle.adaptive_threshold(oracle.to_luma8(img), 7)
        assert np.array_equal(ctx.download_grey(f, w, h, thresholded=True), binary)
        cs, btype, _ = oracle.find_contours(binary)
        keep = [i for i, c in enumerate(cs) if not _isolated(binary, c)]   # 1-pixel components own no border pixel pair
        keys, pts = ctx.contours(f)
        assert len(pts) == len(keep), (f, len(pts), len(keep))
        for j, i in enumerate(keep):
            c = cs[i].astype(np.int64)
            assert int(keys[j]) == 2 * (int(c[0][1]) * w + int(c[0][0])) + int(btype[i]), (f, j)
            assert pts[j].shape == c.shape and np.array_equal(pts[j], c), (f, j)
        n_total += len(keep)
    return ctx, n_total


def test_find_contours_against_oracle_structured(dicts, oracle):
    """Hand-made layouts (shapes on every image edge, 1-pixel strokes whose points are visited twice, nested rings), the
    column-0 start anomaly, and a marker frame."""
    from aruco3_amd import synth

    h, w = 96, 128
    base = np.full((h, w), 210, np.uint8)
    imgs = []
    a = base.copy(); a[:, :20] = 20; a[10:40, 0:60] = 20; a[50:90, 100:128] = 20; imgs.append(a)
    b = base.copy(); b[0:8, :] = 20; b[h - 8:, :] = 20; b[20:70, 30:100] = 20; b[35:55, 45:85] = 210; imgs.append(b)
    c = base.copy()
    for i in range(60):
        c[10 + i, 10 + i] = 20; c[10 + i, 100 - i] = 20
    c[80, 5:120] = 20; imgs.append(c)
    d = np.full((h, w), 30, np.uint8); d[5:90, 5:120] = 220; d[20:70, 20:100] = 30; d[30:60, 30:90] = 220; imgs.append(d)
    e = base.copy(); e[3, 1] = 20; e[4, 0] = 20; e[30:60, 40:90] = 20; imgs.append(e)       # forces the start-resolution fixpoint
    det = _detector(dicts, "ARUCO_DEFAULT")
    ctx, n = _compare_contours(det, oracle, np.stack(imgs))
    assert n > 20 and ctx.stats()["resolve_iterations"] >= 1
    frames, _ = synth.config_frames(1, 1)
    _compare_contours(det, oracle, oracle.to_luma8(frames[0])[None])


def test_find_contours_against_oracle_noise(dicts, oracle):
    """192x160 uniform noise (the reference bench's recipe in small): thousands of tiny borders, one giant component, starts
    in column 0; then the same through the global entry rounds (two frames in one batch, one of them clean)."""
    rng = np.random.default_rng(2026)
    det = _detector(dicts, "ARUCO_DEFAULT")
    noise = rng.integers(0, 256, size=(2, 160, 192), dtype=np.uint8)
    ctx, n = _compare_contours(det, oracle, noise)
    assert n > 2000
    # taps off again: the product path prunes, and must still give the same candidates as the tapped run did
    frames = noise[..., None]
    _, m0, per0 = _detect_host(det, frames, taps=False)
    _, m1, per1 = _detect_host(det, frames, taps=True)
    assert np.array_equal(per0, per1) and marker_tuples(m0) == marker_tuples(m1)


# ------------------------------------------------------------------------------------------------------------------
# the reference's vectors for its small helpers, through the device code that implements them (src/aruco.rs:400-459)
# ------------------------------------------------------------------------------------------------------------------
def test_reference_helper_vectors_on_device(dicts, oracle):
    ctx = _detector(dicts)._context()
    # test_enforce_clockwise, src/aruco.rs:400-412
    cw = [(0, 0), (0, 1), (1, 1), (1, 0)]
    ccw = [(0, 0), (1, 0), (1, 1), (0, 1)]
    out = ctx.debug_clockwise(np.array([cw, ccw]))
    assert out[0].tolist() == out[1].tolist()
    assert np.array_equal(out.astype(np.uint32), oracle.enforce_clockwise_corners(np.array([cw, ccw], dtype=np.uint32)))
    rng = np.random.default_rng(9)
    q = rng.integers(0, 2000, size=(500, 4, 2))
    assert np.array_equal(ctx.debug_clockwise(q).astype(np.uint32), oracle.enforce_clockwise_corners(q.astype(np.uint32)))
    # test_bit_rotate, src/aruco.rs:414-444
    pre = np.array([[1, 1, 1], [1, 0, 0], [0, 1, 0]], dtype=np.uint8)
    post = np.array([[1, 0, 0], [1, 0, 1], [1, 1, 0]], dtype=np.uint8)
    assert np.array_equal(ctx.debug_rotate_bits(pre, 1), post)
    pre = np.array([[1, 1, 1, 1], [1, 1, 1, 0], [1, 1, 0, 0], [1, 0, 0, 0]], dtype=np.uint8)
    post = np.array([[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1]], dtype=np.uint8)
    assert np.array_equal(ctx.debug_rotate_bits(pre, 1), post)
    m = rng.integers(0, 2, size=(7, 7)).astype(np.uint8)
    r = m
    for times in range(1, 5):
        r = oracle.rotate_bit_matrix(r)
        assert np.array_equal(ctx.debug_rotate_bits(m, times), r)
    # test_drop_too_near, src/aruco.rs:446-459
    pts = np.array([
        [(0, 0), (10, 0), (10, 10), (0, 10)],
        [(1, 0), (10, 0), (10, 10), (0, 10)],
        [(0, 0), (10, 2), (10, 10), (0, 10)],
        [(0, 0), (10, 0), (10, 10), (3, 10)],
    ], dtype=np.uint32)
    kept = ctx.debug_discard_too_near(pts, 10.0)
    assert len(kept) == 1
    want, _ = oracle.discard_too_near(pts, 10.0)
    assert np.array_equal(kept, np.asarray(want).reshape(-1, 4, 2))
    for trial in range(20):     # random clusters: order-dependent dead-set semantics
        centres = rng.integers(50, 900, size=(int(rng.integers(1, 6)), 2))
        quads = []
        for _ in range(int(rng.integers(2, 40))):
            cx, cy = centres[int(rng.integers(0, len(centres)))]
            s = int(rng.integers(20, 60))
            j = rng.integers(-6, 7, size=(4, 2))
            quads.append(np.array([(cx - s, cy - s), (cx + s, cy - s), (cx + s, cy + s), (cx - s, cy + s)]) + j)
        quads = np.clip(np.array(quads), 0, 2000).astype(np.uint32)
        want, _ = oracle.discard_too_near(quads, 25.0)
        assert np.array_equal(ctx.debug_discard_too_near(quads, 25.0), np.asarray(want).reshape(-1, 4, 2)), trial
