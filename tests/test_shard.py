"""The multi-GPU bookkeeping (frame partition, dictionary broadcast, detection gather) on CPU: world_size 2, gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aruco3_amd import _lib, shard
from aruco3_amd.dictionaries import ARDictionary
from tests.util import marker_tuples


def test_partition_covers_every_frame_once():
    for n in (0, 1, 7, 256, 2048, 2049):
        for g in (1, 2, 3, 4, 8):
            blocks = [shard.partition(n, g, r) for r in range(g)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(g - 1))
            sizes = [b - a for a, b in blocks]
            assert max(sizes) - min(sizes) <= 1
    assert shard.partition(2048, 8, 3) == (768, 1024)  # BASELINE config 3: 256 frames per GPU


def _fake_markers(rank, frames):
    rng = np.random.default_rng(100 + rank)
    per = rng.integers(0, 5, frames).astype(np.uint32)
    m = np.zeros(int(per.sum()), dtype=_lib.MARKER_DTYPE)
    m["id"] = rng.integers(0, 1023, m.size)
    m["code"] = rng.integers(0, 1 << 25, m.size)
    m["corners"] = rng.integers(0, 1920, (m.size, 8))
    m["hamming_distance"] = rng.integers(0, 3, m.size)
    m["frame"] = np.repeat(np.arange(frames), per)
    return m, per


def _expect(m, per, first_frame):
    """the markers of one rank as the gathered records must return them: .frame holds the GLOBAL frame index"""
    g = m.copy()
    g["frame"] = g["frame"] + first_frame
    out, pos = [], 0
    for f in range(per.size):
        out.append((first_frame + f, marker_tuples(g[pos: pos + int(per[f])])))
        pos += int(per[f])
    return out


def test_pack_unpack_roundtrip():
    m, per = _fake_markers(0, 9)
    rec = shard.pack_detections(m, per, first_frame=40)
    assert rec.shape == (9, shard.record_bytes())
    got = [(frame, marker_tuples(mm)) for frame, mm in shard.unpack_detections(rec)]
    assert got == _expect(m, per, 40)
    assert m["frame"].max() < 9   # the caller's list keeps its local indices


def test_overflowing_frame_raises_instead_of_clipping():
    per = np.array([1, shard.MAXM + 1, 0], dtype=np.uint32)
    m = np.zeros(int(per.sum()), dtype=_lib.MARKER_DTYPE)
    m["frame"] = np.repeat(np.arange(3), per)
    with pytest.raises(shard.RecordOverflow):
        shard.pack_detections(m, per, 0)
    assert shard.pack_detections(m, per, 0, maxm=shard.MAXM + 1).shape[1] == 8 + (shard.MAXM + 1) * 56


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d0 = ARDictionary.new_from_named_dict("APRILTAG_36H11") if rank == 0 else None
        d = shard.broadcast_dictionary(d0, "cpu", 0)
        ref = ARDictionary.new_from_named_dict("APRILTAG_36H11")
        ok = d.num_bits == 36 and d._tau == 11 and np.array_equal(d.code_list, ref.code_list)
        total = 13 if world == 2 else 8 * 5 + 3   # not a multiple of the world size: the shorter blocks are padded for the all-gather
        lo, hi = shard.partition(total, world, rank)
        rows = max(b - a for a, b in (shard.partition(total, world, r) for r in range(world)))
        m, per = _fake_markers(rank, hi - lo)
        g = shard.gather_detections(m, per, lo, "cpu", rows=rows).numpy()
        seen = []
        for r in range(world):
            a, b = shard.partition(total, world, r)
            mr, perr = _fake_markers(r, b - a)
            got = [(frame, marker_tuples(mm)) for frame, mm in shard.unpack_detections(g[r])]
            ok = ok and got == _expect(mr, perr, a)
            seen += [frame for frame, _ in got]
        ok = ok and seen == list(range(total))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_broadcast_and_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(0, True), (1, True)]


def test_broadcast_and_gather_world8_gloo():
    """BASELINE config 3's world size as eight PROCESSES (not eight slices of one): dictionary broadcast, the contiguous split with
    unequal blocks (43 frames: three ranks hold six, five hold five), padded records, one all-gather, global frame order"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(60)
    assert res == [(r, True) for r in range(8)]


def test_bench_launcher_with_eight_ranks_and_no_gpu_ends_the_launch():
    """`python bench.py --gpus 8` as the driver types it, on a box without a GPU: eight children, all of them fail at the first device
    call; the parent names a rank, exits non-zero within its timeout, prints no JSON line and leaves no child running"""
    import subprocess
    import sys
    import time
    from pathlib import Path

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "8", "--backend", "gloo", "--frames", "2", "--steps", "1", "--warmup", "0",
                        "--device-synth", "--repeats", "1", "--no-other-workloads", "--no-cpu-baseline", "--launch-timeout", "240"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=400)
    assert p.returncode != 0 and time.time() - t0 < 300
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank" in p.stderr and "exited with" in p.stderr


def test_bench_launcher_without_a_gpu_fails_loudly_and_cleanly():
    """`python bench.py --gpus 2` typed as is (no launcher, no RANK in the environment) on a box WITHOUT a GPU: the parent starts
    its two child ranks, they die at the first device call, and the parent reports it -- non-zero exit code, no JSON line, no
    rank left running.  (With a GPU the same front door is exercised for real by tests/test_gpu_round3.py.)"""
    import subprocess
    import sys
    from pathlib import Path

    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", "2", "--backend", "gloo", "--frames", "2", "--steps", "1", "--warmup", "0",
                        "--device-synth", "--repeats", "1", "--no-other-workloads", "--no-cpu-baseline", "--launch-timeout", "120"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=200)
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert "rank" in p.stderr and "exited with" in p.stderr
