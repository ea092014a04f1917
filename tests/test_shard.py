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


def test_pack_unpack_roundtrip():
    m, per = _fake_markers(0, 9)
    rec = shard.pack_detections(m, per, first_frame=40)
    out = shard.unpack_detections(rec)
    pos = 0
    for f, (frame, mm) in enumerate(out):
        assert frame == 40 + f
        assert mm.tobytes() == m[pos: pos + int(per[f])].tobytes()
        pos += int(per[f])


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        d0 = ARDictionary.new_from_named_dict("APRILTAG_36H11") if rank == 0 else None
        d = shard.broadcast_dictionary(d0, "cpu", 0)
        ref = ARDictionary.new_from_named_dict("APRILTAG_36H11")
        ok = d.num_bits == 36 and d._tau == 11 and np.array_equal(d.code_list, ref.code_list)
        frames = 6
        lo, hi = shard.partition(frames * world, world, rank)
        m, per = _fake_markers(rank, hi - lo)
        g = shard.gather_detections(m, per, lo, "cpu").numpy()
        for r in range(world):
            mr, perr = _fake_markers(r, frames)
            got = shard.unpack_detections(g[r])
            pos = 0
            for f, (frame, mm) in enumerate(got):
                ok = ok and frame == r * frames + f and mm.tobytes() == mr[pos: pos + int(perr[f])].tobytes()
                pos += int(perr[f])
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
