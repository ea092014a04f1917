"""Frame-level sharding across the GPUs of one node.

`Detector::detect` is a pure function of (config, dictionary, image) (src/aruco.rs:52-121), so a batch
splits into contiguous blocks of frames, one block per rank, with NO collective on the data path.  Two
small collectives frame the work, both over `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" on CPU for the tests):

  * once:      broadcast of the dictionary {num_bits, tau, n, codes[n]} from rank 0,
  * per batch: all-gather of fixed-capacity detection records (count + MAXM markers per frame).

Payloads are kilobytes to a few megabytes, i.e. latency-bound; nothing here depends on link bandwidth.
"""
from typing import List, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .dictionaries import ARDictionary

MAXM = 32                       # markers kept per frame in the gathered record
_REC_BYTES = _lib.MARKER_DTYPE.itemsize  # 56


def partition(n_frames: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block split: rank r owns frames [r*N/G, (r+1)*N/G)."""
    lo = (n_frames * rank) // world_size
    hi = (n_frames * (rank + 1)) // world_size
    return lo, hi


def broadcast_dictionary(d: ARDictionary = None, device="cpu", src: int = 0) -> ARDictionary:
    """Rank `src` passes its dictionary; every rank returns the same one.  Codes travel as int64 bit patterns."""
    rank = dist.get_rank()
    head = torch.zeros(3, dtype=torch.int64, device=device)
    if rank == src:
        head = torch.tensor([d.num_bits, d._tau, d.code_list.size], dtype=torch.int64, device=device)
    dist.broadcast(head, src)
    num_bits, tau, n = (int(v) for v in head.tolist())
    codes = torch.zeros(n, dtype=torch.int64, device=device)
    if rank == src:
        codes = torch.from_numpy(d.code_list.view(np.int64).copy()).to(device)
    dist.broadcast(codes, src)
    return ARDictionary(num_bits, tau, codes.cpu().numpy().view(np.uint64), d.name if rank == src and d is not None else "")


class RecordOverflow(ValueError):
    """A frame holds more markers than the gather record: raised, never clipped."""


_POSE_PAIR_BYTES = 2 * 52        # two a3_pose records (error, 9 rotation floats row-major, 3 translation floats) per marker


def record_bytes(maxm: int = MAXM, with_poses: bool = False) -> int:
    return 8 + maxm * (_REC_BYTES + (_POSE_PAIR_BYTES if with_poses else 0))


_PAD_FRAME = 0xFFFFFFFF   # global frame index of a padding record (ranks with fewer frames than the largest block)


def pack_detections(markers: np.ndarray, per_frame: np.ndarray, first_frame: int = 0, maxm: int = MAXM, poses: np.ndarray = None) -> np.ndarray:
    """Host statement of the record format a3_pack_detections writes on the device (the tests compare the two):
    -> uint8 [frames, record]: u32 count, u32 global frame index, then maxm a3_marker records whose .frame field holds the
    GLOBAL frame index and, when `poses` (float32 [markers, 2, 13]) is given, maxm pose pairs; unused slots are zero.
    A frame with more than maxm markers raises RecordOverflow."""
    n = per_frame.size
    if n and int(per_frame.max()) > maxm:
        raise RecordOverflow(f"a frame holds {int(per_frame.max())} markers, the gather record only {maxm}")
    rec = np.zeros((n, record_bytes(maxm, poses is not None)), dtype=np.uint8)
    head = rec[:, :8].view(np.uint32)
    head[:, 0] = per_frame
    head[:, 1] = np.arange(first_frame, first_frame + n, dtype=np.uint32)
    body = rec[:, 8: 8 + maxm * _REC_BYTES].reshape(n, maxm, _REC_BYTES)
    total = int(per_frame.sum())
    if total:
        counts = per_frame.astype(np.int64)
        frame_of = np.repeat(np.arange(n), counts)                      # frame of every marker (markers are frame-major)
        rank_in_frame = np.arange(total) - np.repeat(np.cumsum(counts) - counts, counts)
        raw = np.ascontiguousarray(markers[:total]).view(np.uint8).reshape(total, _REC_BYTES).copy()   # the caller's list keeps its local indices
        raw[:, :4] = (first_frame + frame_of).astype("<u4").view(np.uint8).reshape(total, 4)           # bytes 0..3 = a3_marker.frame
        body[frame_of, rank_in_frame] = raw
        if poses is not None:
            pbody = rec[:, 8 + maxm * _REC_BYTES:].reshape(n, maxm, _POSE_PAIR_BYTES)
            pbody[frame_of, rank_in_frame] = np.ascontiguousarray(poses[:total], dtype=np.float32).view(np.uint8).reshape(total, _POSE_PAIR_BYTES)
    return rec


def unpack_detections(rec: np.ndarray, with_poses: bool = False):
    """inverse of pack_detections -> list of (global frame index, marker structured array[, poses float32 [count, 2, 13]]);
    padding records are skipped"""
    out = []
    rec = np.asarray(rec)
    maxm = (rec.shape[-1] - 8) // (_REC_BYTES + (_POSE_PAIR_BYTES if with_poses else 0))
    for row in rec.reshape(-1, rec.shape[-1]):
        cnt, frame = (int(v) for v in row[:8].view(np.uint32))
        if frame == _PAD_FRAME:
            continue
        m = row[8: 8 + cnt * _REC_BYTES].copy().view(_lib.MARKER_DTYPE)
        if with_poses:
            p0 = 8 + maxm * _REC_BYTES
            out.append((frame, m, row[p0: p0 + cnt * _POSE_PAIR_BYTES].copy().view(np.float32).reshape(cnt, 2, 13)))
        else:
            out.append((frame, m))
    return out


def _pad_rows(rec: torch.Tensor, rows: int) -> torch.Tensor:
    """all_gather_into_tensor needs equal contributions: ranks whose block is shorter append padding records"""
    if rec.shape[0] == rows:
        return rec
    pad = torch.zeros((rows - rec.shape[0], rec.shape[1]), dtype=torch.uint8, device=rec.device)
    pad[:, 4:8] = 0xFF   # frame = _PAD_FRAME
    return torch.cat([rec, pad])


def _all_gather(rec: torch.Tensor, rows: int, out: torch.Tensor = None) -> torch.Tensor:
    """`out`: a caller-owned uint8 tensor of world * rows * record bytes to gather into (a stepping loop re-uses one instead of
    allocating per collective)"""
    world = dist.get_world_size()
    rec = _pad_rows(rec, rows)
    if out is None:
        out = torch.empty((world * rows, rec.shape[1]), dtype=torch.uint8, device=rec.device)
    out = out.view(world * rows, rec.shape[1])
    dist.all_gather_into_tensor(out, rec)  # ranks concatenated along dim 0
    return out.view(world, rows, rec.shape[1])


def gather_detections(markers: np.ndarray, per_frame: np.ndarray, first_frame: int, device="cpu", rows: int = None,
                      maxm: int = MAXM) -> torch.Tensor:
    """Host-side variant (records packed with numpy): the CPU rehearsal of the gather.  `rows` = frames of the largest block
    (defaults to this rank's count: equal blocks).  Returns uint8 [world, rows, record] on every rank."""
    rec = torch.from_numpy(pack_detections(markers, per_frame, first_frame, maxm)).to(device)
    return _all_gather(rec, rows if rows is not None else rec.shape[0])


def pack_detections_device(ctx: "_lib.Context", n_frames: int, first_frame: int, device, maxm: int = MAXM, with_poses: bool = False,
                           out: torch.Tensor = None) -> torch.Tensor:
    """The last batch of `ctx` as gather records, written by a kernel straight from the device-resident marker list
    (a3_pack_detections): no D2H, no numpy, no H2D.  with_poses: the batch was a detect_batch_pose call and every marker's
    pose pair travels with it (BASELINE config 5).  Enqueued on the context's stream -- call under
    `torch.cuda.stream(<that stream>)` so that torch orders the collective after it."""
    rec = out if out is not None else torch.empty((n_frames, record_bytes(maxm, with_poses)), dtype=torch.uint8, device=device)
    assert rec.shape == (n_frames, record_bytes(maxm, with_poses)) and rec.dtype == torch.uint8 and rec.is_contiguous()
    try:
        ctx.pack_detections(first_frame, maxm, rec.data_ptr(), rec.numel(), with_poses)
    except _lib.A3Error as e:
        if e.code == _lib.ERR_CAPACITY:
            raise RecordOverflow(str(e)) from e
        raise
    return rec


def gather_detections_device(ctx: "_lib.Context", n_frames: int, first_frame: int, device, coll_device=None, rows: int = None,
                             maxm: int = MAXM, with_poses: bool = False) -> torch.Tensor:
    """Per batch: device-packed records of this rank's frames, all-gathered (RCCL all-gather over xGMI with backend "nccl").
    coll_device = "cpu" moves the packed tensor to the host first (gloo rehearsal on a box with fewer GPUs than ranks).

    Ordering: a3_pack_detections enqueues its kernel on the CONTEXT's stream (a3_get_stream: its own non-blocking stream unless
    a3_set_stream changed it), which torch knows nothing about.  The record tensor is therefore allocated, packed and handed to
    the copy / collective under that stream wrapped as a torch stream, and torch's current stream waits for it before the
    result is returned -- the pack can neither race the allocation nor be read before it has run."""
    cs = torch.cuda.ExternalStream(ctx.stream_ptr, device=device) if ctx.stream_ptr else None
    cur = torch.cuda.current_stream(device)
    if cs is None:           # the context runs on the legacy default stream: torch's default stream is that very stream
        cs = torch.cuda.default_stream(device)
    cs.wait_stream(cur)      # (memory the caching allocator hands out may still be in use on the current stream)
    with torch.cuda.stream(cs):
        rec = pack_detections_device(ctx, n_frames, first_frame, device, maxm, with_poses)
        if coll_device is not None and torch.device(coll_device).type == "cpu":
            rec = rec.cpu()              # ordered behind the pack on cs, and synchronous for the host
            out = _all_gather(rec, rows if rows is not None else n_frames)
        else:
            out = _all_gather(rec, rows if rows is not None else n_frames)   # RCCL orders itself after cs's pending work
            rec.record_stream(cs)
            out.record_stream(cur)
    cur.wait_stream(cs)
    return out
