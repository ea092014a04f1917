"""Round-4 GPU tests: burst stepping (a3_order_after) gives the results of synchronous calls in every arrangement, BASELINE
config 5 through the bench's front door with two ranks (poses in the gather records), the library reports which build it is.
Everything goes through the C ABI; the oracle is the checker.  GPU only."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.util import bench_output, marker_tuples

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


def test_the_test_run_uses_the_product_library():
    """A leftover A3_HIP_LIB pointing at a tuning build must not pass for the product."""
    from aruco3_amd import _lib

    info = _lib.library_info()
    assert info["abi"] == 5
    if not info["from_A3_HIP_LIB"]:
        assert info["tuning_build"] is False and info["non_default_kernel_build"] is False
        assert Path(info["path"]) == ROOT / "aruco3_amd" / "libaruco3_hip.so"


def test_burst_stepping_equals_synchronous_calls(dicts):
    """Four contexts on streams of their own, used in rotation with the burst gates of include/aruco3_hip.h (context k orders itself
    after the contexts k+1 .. N-1 before each submit): every batch -- they alternate between two different sets of frames -- must
    equal the synchronous call's result; also with every gate (a stricter order), with gates between contexts that share one
    stream (no-ops), and with a gate on a context whose decode stage is being held back (it is released first)."""
    import torch

    from aruco3_amd import _lib, synth

    fa, _ = synth.config_frames(1, 6)
    fb, _ = synth.config_frames(1, 6, first=6)
    da, db = torch.from_numpy(fa).cuda(), torch.from_numpy(fb).cuda()
    aa, ab = _args(fa, _lib.MEM_DEVICE, da.data_ptr()), _args(fb, _lib.MEM_DEVICE, db.data_ptr())
    L = _lib.load()
    ctxs = [_detector(dicts, "ARUCO_DEFAULT")._context() for _ in range(4)]
    want = [ctxs[0].detect_batch(*aa), ctxs[0].detect_batch(*ab)]
    same = lambda got, w: marker_tuples(got[0]) == marker_tuples(w[0]) and np.array_equal(got[1], w[1])
    assert len(want[0][0]) > 0 and marker_tuples(want[0][0]) != marker_tuples(want[1][0])

    def rotate(k, nc, gate):
        cs = ctxs[:nc]
        which = {}

        def sub(i):
            for m in gate(i % nc, nc):
                cs[i % nc].order_after(cs[m])
            which[i] = i % 2 if i % 3 else 1 - i % 2
            cs[i % nc].submit(*(aa if which[i] == 0 else ab))

        for i in range(min(nc, k)):
            sub(i)
        for i in range(k):
            assert same(cs[i % nc].collect(), want[which[i]]), (i, nc)
            if i + nc < k:
                sub(i + nc)

    try:
        for mode, hold in ((-1, 1), (0, 1), (0, 0), (2, 1)):     # the library's own rule (the default: own streams + gates = chains held back) / the same forced / chains enqueued at submit / round 4's default: decode stage deferred, nothing held
            assert L.a3_debug_set_overlap(mode) == 0 and L.a3_debug_set_hold(hold) == 0
            for nc in (4, 3, 2):
                rotate(13, nc, lambda k, n: range(k + 1, n))                       # the documented rule
                rotate(9, nc, lambda k, n: [m for m in range(n) if m != k])        # every other context (nobody is "last": collect releases)
                rotate(9, nc, lambda k, n: [(k + 1) % n])                          # only the next one
        assert L.a3_debug_set_overlap(-1) == 0 and L.a3_debug_set_hold(1) == 0
        # held chains in awkward orders: collected before the burst's last member ever submits; collected in reverse; a gate on a
        # context whose chain is held (it is released first); a synchronous call in between; a member destroyed while held
        c0, c1, c2, c3 = ctxs
        c0.order_after(c1); c0.submit(*aa)                       # held: nobody releases it ...
        assert same(c0.collect(), want[0])                       # ... collect does
        c0.order_after(c1); c0.submit(*aa); c1.order_after(c2); c1.submit(*ab); c2.submit(*aa)    # c2 is the last member: releases c0, c1
        assert same(c2.collect(), want[0]) and same(c1.collect(), want[1]) and same(c0.collect(), want[0])
        c0.order_after(c3); c0.submit(*ab)                       # held
        c1.order_after(c0)                                       # a gate on the holder: its chain goes out first
        c1.submit(*aa)                                           # (c1 declared a gate: held itself)
        assert same(c3.detect_batch(*ab), want[1])               # a synchronous call on a third context does not disturb them
        assert same(c1.collect(), want[0]) and same(c0.collect(), want[1])
        # the caller moves a holding context to another stream: the held chain goes out on the old one first
        st2 = torch.cuda.Stream()
        c0.order_after(c1); c0.submit(*aa)
        old_stream = c0.stream_ptr
        c0.set_stream(st2.cuda_stream)
        assert same(c0.collect(), want[0])
        c0.set_stream(old_stream)
        extra = _detector(dicts, "ARUCO_DEFAULT")._context()
        extra.order_after(c0); c0.submit(*aa); extra.order_after(c0); extra.submit(*ab)      # extra holds a chain ...
        extra.close()                                            # ... and is destroyed with it
        c1.submit(*ab)                                           # the next last member must not trip over it
        assert same(c0.collect(), want[0]) and same(c1.collect(), want[1])
        # contexts that share the caller's stream: already in order, the call is a no-op and must stay harmless
        st = torch.cuda.Stream()
        for cx in ctxs:
            cx.set_stream(st.cuda_stream)
        rotate(9, 4, lambda k, n: range(k + 1, n))
        # a context orders itself after itself / after a context that has never run anything
        fresh = _detector(dicts, "ARUCO_DEFAULT")._context()
        ctxs[0].order_after(ctxs[0]); ctxs[0].order_after(fresh); fresh.order_after(ctxs[0])
        assert same(fresh.detect_batch(*aa), want[0])
    finally:
        L.a3_debug_set_overlap(-1)
        L.a3_debug_set_hold(1)


def test_held_chains_with_poses_and_host_threads(dicts):
    """bursts with a3_detect_batch_pose_submit (the pose request must survive the hold) and with the members driven from different
    host threads (the last member's thread enqueues the others' chains): results equal the synchronous calls'"""
    import threading

    import torch

    from aruco3_amd import _lib, synth

    fa, _ = synth.config_frames(1, 4)
    da = torch.from_numpy(fa).cuda()
    aa = _args(fa, _lib.MEM_DEVICE, da.data_ptr())
    L = _lib.load()
    ctxs = [_detector(dicts, "ARUCO_DEFAULT")._context() for _ in range(4)]
    try:   # (library defaults: nothing is switched)
        wm, wp, wposes = ctxs[0].detect_batch_pose(*aa, 40.0, None, 256)
        plain = ctxs[0].detect_batch(*aa)
        assert len(wm) > 0
        for rounds in range(3):
            for k, cx in enumerate(ctxs):       # members 0, 2 ask for poses, 1, 3 do not
                for m in range(k + 1, 4):
                    cx.order_after(ctxs[m])
                if k % 2 == 0:
                    cx.submit_pose(*aa, 40.0, None, 256)
                else:
                    cx.submit(*aa, out_cap=256)
            for k, cx in enumerate(ctxs):
                if k % 2 == 0:
                    m, p, poses = cx.collect_pose()
                    assert marker_tuples(m) == marker_tuples(wm) and np.array_equal(poses.view(np.uint32), wposes.view(np.uint32))
                else:
                    m, p = cx.collect()
                    assert marker_tuples(m) == marker_tuples(plain[0])
        # four threads, one context each, a barrier per rotation so that the submits of a burst come from different threads
        errs = []
        bar = threading.Barrier(4)

        def worker(k):
            try:
                cx = ctxs[k]
                for r in range(25):
                    bar.wait()
                    for _ in range(k):
                        pass
                    for m in range(k + 1, 4):
                        cx.order_after(ctxs[m])
                    cx.submit(*aa, out_cap=256)
                    m_, p_ = cx.collect()
                    assert marker_tuples(m_) == marker_tuples(plain[0]), (k, r)
            except Exception as e:   # noqa: BLE001
                errs.append((k, repr(e)))
                bar.abort()

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        assert not errs, errs
    finally:
        pass


def test_bench_config5_two_ranks_gloo_with_poses_in_the_gather():
    """BASELINE config 5 through the bench's front door: `python bench.py --workload c5 --gpus 2 --backend gloo` (two fresh child
    ranks on the one leased GPU): detect + pose in submit / collect form on four contexts per rank, pose pairs inside the gather
    records, rank 0's poses bit-equal after the gather."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", "c5", "--gpus", "2", "--backend", "gloo", "--frames", "4", "--steps", "4", "--warmup", "1",
           "--device-synth", "--repeats", "2", "--no-other-workloads", "--no-cpu-baseline", "--isolated-launches", "2", "--launch-timeout", "500"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)
    assert line["n_gpus"] == 2 and line["config"]["resolution"] == [3840, 2160] and line["gathered"]["rank0_poses_bit_equal_after_gather"] is True
    assert out["n_gpus"] == 2 and out["config"]["resolution"] == [3840, 2160] and "detect + estimate_pose" in out["metric"]
    g = out["gathered"]
    assert g["frames"] == 2 * 4 * g["batches_in_last_collective"] and g["global_frame_indices_in_order"] is True
    assert g["rank0_poses_bit_equal_after_gather"] is True
    assert g["pose_pairs_gathered"] >= 4 * g["batches_in_last_collective"] * 10 and g["max_markers_per_record"] >= 16
    assert g["record_bytes"] == 8 + g["max_markers_per_record"] * (56 + 104)
    assert out["gates"] == "none" and out["contexts"] == 4


def test_bench_line_carries_parity_and_isolated_roofline():
    """one small run of the default workload's code path: parity_in_run against the oracle, the roofline from isolated launches,
    e2e_frac, the A/B against the shared-stream stepping, the library's identity"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--frames", "8", "--steps", "4", "--warmup", "1", "--repeats", "2", "--isolated-launches", "3",
           "--synth-workers", "4"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)
    # what the driver parses: the contract's keys, both rooflines, the CPU baseline and the parity count, in at most 4 KB
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "roofline_warp", "cpu_baseline", "parity_in_run", "e2e_frac", "stage_ms_per_step", "stepping", "gates"):
        assert k in line, k
    assert line["value"] == out["value"] and line["ms_per_step"] == out["ms_per_step"] and line["stepping"] == "free-running"
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "bytes_per_launch", "avg_launch_ms", "launches_timed", "kernel"):
        assert k in line["roofline"] or (k in ("traffic", "traffic_source") and out["roofline"][k] is None), k
    assert line["roofline"]["frac"] == out["roofline"]["frac"] and line["roofline_warp"]["frac"] == out["roofline_warp"]["frac"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"]) and line["cpu_baseline"]["kind"] == "port"
    assert (line["parity_in_run"]["frames_equal"], line["parity_in_run"]["frames_compared"]) == (32, 32)
    assert out["parity_in_run"]["summary"] == "32/32 frames" and out["parity_in_run"]["batches_covered"] == 4     # four distinct batches of 8 in flight
    assert out["config"]["distinct_batches_in_flight"] == 4 and out["library"]["internal_switches_used"] == []
    assert out["library_stepping_seen"] == ["whole"] * 4 and out["gates"] == "none"                     # the headline: free-running rotation
    assert out["ab_burst_gates"]["library_stepping_seen"] == ["held_released_by_last"] * 3 + ["burst_last"]   # the same calls + a3_order_after
    assert out["ab_r04_library_default"]["library_stepping_seen"] == ["decode_deferred"] * 4
    assert out["roofline_warp"]["avg_launch_ms"] > out["roofline_warp"]["sampling_only_ms"] > 0
    assert out["roofline"]["launches_timed"] == 3 and out["roofline"]["avg_launch_ms"] > 0 and 0 < out["e2e_frac"] < 1
    assert out["ab_shared_stream"]["same_markers"] is True
    assert out["library"]["tuning_build"] is False
    ow = out["other_workloads"]
    for k in ("C0_reference_bench_noise_1080p", "C4_apriltag36h11_720p_noise", "C5_4k_16_markers_detect_plus_pose"):
        if "skipped" not in ow[k]:
            pr = ow[k]["parity_in_run"]
            assert pr["frames_equal"] == pr["frames_compared"] > 0, (k, pr)
