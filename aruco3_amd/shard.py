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


def pack_detections(markers: np.ndarray, per_frame: np.ndarray, first_frame: int = 0) -> np.ndarray:
    """-> uint8 [frames, 8 + MAXM*56]: u32 count, u32 global frame index, then up to MAXM a3_marker records."""
    n = per_frame.size
    rec = np.zeros((n, 8 + MAXM * _REC_BYTES), dtype=np.uint8)
    head = rec[:, :8].view(np.uint32)
    head[:, 0] = np.minimum(per_frame, MAXM)
    head[:, 1] = np.arange(first_frame, first_frame + n, dtype=np.uint32)
    body = rec[:, 8:].reshape(n, MAXM, _REC_BYTES)
    total = int(per_frame.sum())
    if total:
        raw = np.ascontiguousarray(markers[:total]).view(np.uint8).reshape(total, _REC_BYTES)
        counts = per_frame.astype(np.int64)
        frame_of = np.repeat(np.arange(n), counts)                      # frame of every marker (markers are frame-major)
        rank_in_frame = np.arange(total) - np.repeat(np.cumsum(counts) - counts, counts)
        keep = rank_in_frame < MAXM
        body[frame_of[keep], rank_in_frame[keep]] = raw[keep]
    return rec


def unpack_detections(rec: np.ndarray):
    """inverse of pack_detections -> list of (global frame index, marker structured array)"""
    out = []
    for row in rec:
        cnt, frame = (int(v) for v in row[:8].view(np.uint32))
        m = row[8: 8 + cnt * _REC_BYTES].copy().view(_lib.MARKER_DTYPE)
        out.append((frame, m))
    return out


def gather_detections(markers: np.ndarray, per_frame: np.ndarray, first_frame: int, device="cpu") -> torch.Tensor:
    """All ranks contribute the same number of frames (weak scaling); returns uint8 [world, frames, record] on every rank."""
    rec = torch.from_numpy(pack_detections(markers, per_frame, first_frame)).to(device)
    world = dist.get_world_size()
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, rec)  # ranks concatenated along dim 0
    return out.view(world, rec.shape[0], rec.shape[1])
