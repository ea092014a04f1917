"""Round-6 GPU tests: quirk Q4 on the device (a failed projection -> 1 x 1 black patch -> code 0 looked up), eight processes
through the bench's front door, a C11 caller of the public header.  Everything goes through the C ABI.  GPU only."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.util import bench_output, markers_of_hip, markers_of_oracle

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


# ------------------------------------------------------------------------------------------------------------------
# quirk Q4 (src/aruco.rs:255-257, 264-292)
# ------------------------------------------------------------------------------------------------------------------
# quads no convex hull can deliver (from_control_points has no solution for them: |det| of the 8 x 8 system is 0)
_DEGENERATE = [
    [[10, 10], [50, 50], [90, 90], [130, 130]],          # four collinear points
    [[300, 300], [300, 300], [340, 300], [340, 340]],    # a repeated corner
    [[200, 20], [260, 20], [260, 20], [200, 20]],        # two pairs of equal corners (a segment)
    [[77, 401], [77, 401], [77, 401], [77, 401]],        # a single point
    [[500, 100], [560, 100], [620, 100], [560, 160]],    # three collinear corners: a triangle with a point on one side
]


def _q4_case(dicts, oracle, d, filt, extra=()):
    """one 640x480 frame (BASELINE config 1); its real candidates, a few of them, and the degenerate quads interleaved, are injected
    in the same order on both sides -> every stage from discard_too_near on must agree, including the Q4 branch"""
    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig

    frames, _ = synth.config_frames(1, 1)
    img = frames[0]
    h, w = img.shape[:2]
    cfg = DetectorConfig(filter_high_bit_errors=filt)
    ocfg = oracle.Config.default()
    ocfg.filter_high_bit_errors = int(filt)
    plain = oracle.detect(img, d.code_list, d.num_bits, d.tau, config=ocfg)
    real = plain["candidates_pre"]
    assert len(real) >= 4
    quads = []
    for i, q in enumerate(_DEGENERATE + list(extra)):        # degenerate and real quads alternate: the sort / discard sees both kinds
        quads.append(np.asarray(q, np.uint32))
        quads.append(real[i % len(real)])
    quads = np.stack(quads)
    ref = oracle.detect(img, d.code_list, d.num_bits, d.tau, config=ocfg, quads=quads)
    assert 0 in ref["homography_ok"].tolist() and 1 in ref["homography_ok"].tolist()      # the branch IS taken, beside ordinary candidates

    ctx = Detector(cfg, d)._context()
    a = np.ascontiguousarray(frames)
    args = (a.ctypes.data, _lib.MEM_HOST, _lib.FMT_RGB8, w, h, w * 3, h * w * 3, 1)
    got = {}
    for taps in (True, False):                # the tapped path (every stage compared) and the product path (marker list)
        ctx.set_debug_taps(taps)
        ctx.debug_inject_candidates(quads)
        m, per = ctx.detect_batch(*args)
        got[taps] = markers_of_hip(m)
        assert got[taps] == markers_of_oracle(ref), (taps, got[taps], markers_of_oracle(ref))
        if taps:
            assert ctx.candidates(0, before_discard=True).tolist() == quads.tolist()
            assert ctx.candidates(0).tolist() == ref["candidates"].tolist()
            patches, ok, codes, dec = ctx.homographies(0, with_patches=True)
            assert ok.tolist() == ref["homography_ok"].tolist()
            assert dec.tolist() == ref["decode_ok"].tolist()
            assert codes.tolist() == ref["codes"].tolist()
            for k, o in enumerate(ok.tolist()):
                if o:
                    assert np.array_equal(patches[k], ref["homographies"][k]), k
                else:                          # GrayImage::new(1, 1): one zero pixel stands in; code 0 in all four rotations
                    assert dec[k] == 1 and codes[k].tolist() == [0, 0, 0, 0]
    # one shot: the batch after the injected one is the frame's own again
    m, per = ctx.detect_batch(*args)
    assert markers_of_hip(m) == markers_of_oracle(plain)
    return ref


def test_quirk_q4_failed_projection_is_decoded_as_code_zero(dicts, oracle):
    """src/aruco.rs:255-257: `Projection::from_control_points` fails -> `GrayImage::new(1, 1)` -> otsu / threshold / resize of one
    black pixel -> an all-zero bit grid passes the border test -> code 0 in four rotations -> find_nearest(0) -> accepted iff the
    distance is below tau or the filter is off.  The oracle's branch (oracle/a3_oracle.c, detect_impl) against the device's
    (k_frame_candidates' solve -> ok = 0, k_decode's 1 x 1 stand-in), on quads no hull can deliver, led in by a test hook."""
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    # filter on: popcount of every ARUCO code is >= 5 > tau = 3: the stand-ins are looked up and rejected
    ref = _q4_case(dicts, oracle, d, True)
    bad = [k for k, o in enumerate(ref["homography_ok"].tolist()) if not o]
    assert bad and not [m for m in ref["markers"] if m["candidate_index"] in bad]
    # filter off: found_any alone accepts -- id = the first code of minimal popcount, rotation 0, hamming distance = that popcount
    ref = _q4_case(dicts, oracle, d, False)
    q4 = [m for m in ref["markers"] if m["candidate_index"] in bad]
    pc = [bin(int(c)).count("1") for c in d.code_list]
    assert q4 and all(m["code"] == 0 and m["id"] == int(np.argmin(pc)) and m["hamming_distance"] == min(pc) for m in q4)


def test_quirk_q4_survives_the_filter_when_a_code_is_light(dicts, oracle):
    """"survives only if some dictionary code has popcount < tau" -- a dictionary that holds one (code 0x3 among ARUCO's, tau 3):
    the failed projection then yields a MARKER with the filter on (id of that code, hamming distance 2, corners of the degenerate quad)"""
    base = dicts.new_from_named_dict("ARUCO_DEFAULT")
    codes = base.code_list.copy()
    codes[17] = 0x3
    d = dicts(base.num_bits, 3, codes, "ARUCO with a light code")
    ref = _q4_case(dicts, oracle, d, True)
    bad = [k for k, o in enumerate(ref["homography_ok"].tolist()) if not o]
    q4 = [m for m in ref["markers"] if m["candidate_index"] in bad]
    assert q4 and all((m["id"], m["code"], m["hamming_distance"], m["rotation"]) == (17, 0, 2, 0) for m in q4)


# ------------------------------------------------------------------------------------------------------------------
# processes, not slices, through the bench's front door
# ------------------------------------------------------------------------------------------------------------------
# A GPU box of this pool admits six processes on its card at once; this test process is one of them, and one slot is left free for
# whatever the harness around the tests may hold: FOUR fresh child ranks through the front door here (five ran as a rehearsal outside
# pytest: profiles/r06_rehearsal_n5_gloo.*; eight processes as processes run on CPU: tests/test_shard.py,
# test_broadcast_and_gather_world8_gloo and the eight-rank launcher test).
_RANKS = 4


def _front_door(extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE", "A3_HIP_LIB"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(_RANKS), "--backend", "gloo", "--frames", "8", "--steps", "2", "--warmup", "1",
           "--device-synth", "--repeats", "2", "--isolated-launches", "2", "--no-other-workloads", "--no-cpu-baseline", "--launch-timeout", "600"] + extra
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_front_door_with_four_child_ranks():
    """`python bench.py --gpus 4 --backend gloo ...` typed as is: four fresh processes on the one leased GPU, each with four contexts and
    four batches of 8 frames; dictionary broadcast, device-packed records, one all-gather per rotation, barrier + max-over-ranks timing"""
    p = _front_door([])
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)
    assert line["n_gpus"] == _RANKS and line["scaling"] == "weak" and line["dist"]["world_size"] == _RANKS and line["dist"]["backend"] == "gloo"
    g = out["gathered"]
    assert g["frames"] == _RANKS * 8 * g["batches_in_last_collective"] and g["global_frame_indices_in_order"] is True
    assert g["all_ranks_ids_correct"] >= 0.8 * g["frames"]
    assert out["dist"]["hw_queues"] is not None and "self-launched" in out["dist"]["launcher"]
    assert out["value"] > 0 and out["config"]["frames_per_gpu"] == 8
    ok, n = (int(v) for v in out["frames_with_all_ids_correct"].split("/"))
    assert n == 32 and ok >= 25


def test_bench_front_door_ends_the_launch_when_a_rank_dies():
    """rank 3 of 4 exits once the process group is up: its peers would wait for it in the first collective; the parent reports the
    rank, kills the others by PID, exits non-zero well inside the launch timeout and prints no JSON line"""
    import time

    t0 = time.time()
    p = _front_door(["--fail-rank", "3"], timeout=600)
    assert p.returncode != 0 and time.time() - t0 < 400
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank 3 exited with 3" in p.stderr


# ------------------------------------------------------------------------------------------------------------------
# what a3_get_stats says between the submit and the collect of a held batch
# ------------------------------------------------------------------------------------------------------------------
def test_a_held_chain_reports_held_until_it_goes_out(dicts):
    """a gated submit holds its chain: a3_stats.stepping reads A3_STEP_HELD (5) until somebody releases it -- here collect -- and the
    collected batch then says how it went out; a context alone on a CALLER's stream is held the same way (header, a3_order_after)"""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig

    fa, _ = synth.config_frames(1, 3)
    da = torch.from_numpy(fa).cuda()
    n, h, w, c = fa.shape
    aa = (da.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    c0, c1 = (Detector(DetectorConfig(), dicts.new_from_named_dict("ARUCO_DEFAULT"))._context() for _ in range(2))
    want = c0.detect_batch(*aa); c0.detect_batch(*aa); c1.detect_batch(*aa); c1.detect_batch(*aa)
    for own_stream in (True, False):
        s0 = torch.cuda.Stream()
        if not own_stream:
            c0.set_stream(s0.cuda_stream)
        c0.order_after(c1); c0.submit(*aa)
        assert c0.stats()["stepping"] == "held"
        got = c0.collect()
        assert c0.stats()["stepping"] == "held_released_early"
        assert markers_of_hip(got[0]) == markers_of_hip(want[0])
        c0.order_after(c1); c0.submit(*aa)
        assert c0.stats()["stepping"] == "held"
        c1.submit(*aa)                                  # the burst's last member releases it
        assert c0.stats()["stepping"] == "held_released_by_last"
        assert markers_of_hip(c0.collect()[0]) == markers_of_hip(want[0]) and markers_of_hip(c1.collect()[0]) == markers_of_hip(want[0])


# ------------------------------------------------------------------------------------------------------------------
# round 6: a frozen window carries its entry's slot -- the path of a contraction tile that spans more than 64 tiny frames
# ------------------------------------------------------------------------------------------------------------------
def test_thousands_of_tiny_frames_share_contraction_tiles(dicts, oracle):
    """3000 frames of 5 x 4 pixels of noise in one batch: a frame owns a dozen darts, a 1024-dart tile of k_local_contract spans
    some eighty frames -- more than the 64 whose entries it counts in LDS, so the entries of the frame that straddles the tile's end
    take their slots one by one (k_local_contract's `direct` path).  find_contours itself (count, order, start, type, every point)
    against the oracle, frame by frame."""
    from tests.test_gpu_shard_taps import _compare_contours, _detector

    rng = np.random.default_rng(606)
    frames = rng.integers(0, 256, size=(3000, 4, 5), dtype=np.uint8)
    det = _detector(dicts, "ARUCO_DEFAULT")
    ctx, n = _compare_contours(det, oracle, frames)
    st = ctx.stats()
    assert n > 3000 and st["darts"] > 20000 and st["darts"] / 3000 < 15.5, st    # fewer than 16 darts a frame: > 64 frames per 1024-dart tile
