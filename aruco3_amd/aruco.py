"""Host-side mirror of src/aruco.rs: `Detector`, `DetectorConfig`, `Detection`, `Marker`.

Same names, same fields, same argument meaning as the reference; the body of
`Detector.detect` is one call into the HIP library (batch of 1).  `detect_batch` is the
natural extension the GPU wants: many independent frames per call, on device memory when
the caller already has them there (torch tensors are accepted for that).
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

from . import _lib
from .dictionaries import ARDictionary


@dataclass
class DetectorConfig:
    """src/aruco.rs:23-43"""
    threshold_window: int = 7
    contour_simplification_epsilon: float = 0.05
    min_side_length_factor: float = 0.2
    min_corner_separation_factor: float = 0.1
    homography_sample_size: int = 49
    filter_high_bit_errors: bool = True

    @classmethod
    def default(cls) -> "DetectorConfig":
        return cls()

    def _c(self) -> _lib.Config:
        return _lib.Config(self.threshold_window, self.contour_simplification_epsilon, self.min_side_length_factor,
                           self.min_corner_separation_factor, self.homography_sample_size, int(self.filter_high_bit_errors))


@dataclass
class Marker:
    """src/aruco.rs:8-13"""
    id: int
    code: int
    corners: List[Tuple[int, int]]
    hamming_distance: int


@dataclass
class Detection:
    """src/aruco.rs:15-21.  grey / candidates / homographies are filled on request only (populate=True):
    copying 2 MB per 1080p frame back to the host would dominate the call."""
    grey: Optional[np.ndarray] = None
    candidates: List[List[Tuple[int, int]]] = field(default_factory=list)
    homographies: List[np.ndarray] = field(default_factory=list)
    markers: List[Marker] = field(default_factory=list)


def _as_frames(image):
    """-> (pointer, memory kind, fmt, w, h, row_stride, frame_stride, n, keepalive)"""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(image, torch.Tensor):
        t = image
        if t.dtype != torch.uint8:
            raise TypeError("frames must be uint8")
        if t.dim() == 2:
            t = t[None, :, :, None]
        elif t.dim() == 3:
            t = t[None] if t.shape[-1] in (1, 3, 4) else t[..., None]
        if t.dim() != 4 or t.shape[-1] not in (1, 3, 4):
            raise ValueError("expected (N,H,W,C) with C in 1,3,4")
        t = t.contiguous()
        n, h, w, c = t.shape
        mem = _lib.MEM_DEVICE if t.is_cuda else _lib.MEM_HOST
        return t.data_ptr(), mem, {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c], w, h, w * c, h * w * c, n, t
    a = np.asarray(image)
    if a.dtype != np.uint8:
        raise TypeError("frames must be uint8")
    if a.ndim == 2:
        a = a[None, :, :, None]
    elif a.ndim == 3:
        a = a[None] if a.shape[-1] in (1, 3, 4) else a[..., None]
    if a.ndim != 4 or a.shape[-1] not in (1, 3, 4):
        raise ValueError("expected (N,H,W,C) with C in 1,3,4")
    a = np.ascontiguousarray(a)
    n, h, w, c = a.shape
    return a.ctypes.data, _lib.MEM_HOST, {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c], w, h, w * c, h * w * c, n, a


class Detector:
    """`Detector { config, dictionary }` (src/aruco.rs:46-49)."""

    def __init__(self, config: DetectorConfig = None, dictionary: ARDictionary = None, device: int = 0):
        self.config = config or DetectorConfig()
        self.dictionary = dictionary or ARDictionary.new_from_named_dict("ARUCO")
        self.device = device
        self._ctx = None
        self._ctx_key = None

    def _context(self) -> _lib.Context:
        key = (tuple(vars(self.config).items()), id(self.dictionary), self.device)
        if self._ctx is None or self._ctx_key != key:
            d = self.dictionary
            self._ctx = _lib.Context(self.config._c(), d.code_list, d.num_bits, d._tau, self.device)
            self._ctx_key = key
        return self._ctx

    # src/aruco.rs:52-121
    def detect(self, image, populate: bool = False) -> Detection:
        return self.detect_batch(image, populate=populate)[0]

    def detect_batch(self, images, populate: bool = False, stream: int = None, out_cap: int = 0, bgra: bool = False) -> List[Detection]:
        """`bgra=True`: 4-channel frames are in webcam byte order B,G,R,A (examples/webcam_kamera.rs:38-52 re-orders them on
        the CPU before `detect`; here the kernel reads them as they are)."""
        ctx = self._context()
        ptr, mem, fmt, w, h, rs, fs, n, keep = _as_frames(images)
        if bgra:
            if fmt != _lib.FMT_RGBA8:
                raise ValueError("bgra=True needs 4-channel frames")
            fmt = _lib.FMT_BGRA8
        if stream is not None:
            ctx.set_stream(stream)
        ctx.set_debug_taps(populate)
        markers, per = ctx.detect_batch(ptr, mem, fmt, w, h, rs, fs, n, out_cap)
        out = []
        pos = 0
        for f in range(n):
            det = Detection()
            for m in markers[pos: pos + int(per[f])]:
                c = m["corners"]
                det.markers.append(Marker(int(m["id"]), int(m["code"]), [(int(c[2 * i]), int(c[2 * i + 1])) for i in range(4)],
                                          int(m["hamming_distance"])))
            pos += int(per[f])
            if populate:
                det.grey = ctx.download_grey(f, w, h)
                det.candidates = [[(int(x), int(y)) for x, y in q] for q in ctx.candidates(f)]
                patches, ok, _, _ = ctx.homographies(f)
                det.homographies = [p if o else np.zeros((1, 1), np.uint8) for p, o in zip(patches, ok)]  # src/aruco.rs:256
            out.append(det)
        return out

    def detect_batch_with_pose(self, images, marker_size_mm: float, intrinsics=None, stream: int = None, out_cap: int = 0):
        """detect + `pose::solve_with_undistorted_points` / `solve_with_intrinsics` of every marker (src/pose.rs:52-81) in one
        device pass, the way examples/webcam_kamera.rs:56-71 chains them.  -> [(Detection, [(MarkerPose, MarkerPose), ...])]"""
        from .pose import MarkerPose

        ctx = self._context()
        ptr, mem, fmt, w, h, rs, fs, n, keep = _as_frames(images)
        if stream is not None:
            ctx.set_stream(stream)
        ctx.set_debug_taps(False)
        intr = None
        if intrinsics is not None:
            ci = intrinsics
            intr = _lib.Intrinsics(ci.image_width, ci.image_height, ci.focal_x, ci.focal_y, ci.principal_x, ci.principal_y)
        markers, per, poses = ctx.detect_batch_pose(ptr, mem, fmt, w, h, rs, fs, n, marker_size_mm, intr, out_cap)
        out = []
        pos = 0
        for f in range(n):
            det = Detection()
            pp = []
            for i in range(pos, pos + int(per[f])):
                m = markers[i]
                c = m["corners"]
                det.markers.append(Marker(int(m["id"]), int(m["code"]), [(int(c[2 * k]), int(c[2 * k + 1])) for k in range(4)],
                                          int(m["hamming_distance"])))
                pp.append(tuple(MarkerPose(float(q[0]), q[1:10].reshape(3, 3).copy(), q[10:13].copy()) for q in poses[i]))
            pos += int(per[f])
            out.append((det, pp))
        return out

    def detect_batch_raw(self, images, stream: int = None, out_cap: int = 0):
        """Batch entry without Python object construction: (structured marker array, per-frame counts)."""
        ctx = self._context()
        ptr, mem, fmt, w, h, rs, fs, n, keep = _as_frames(images)
        if stream is not None:
            ctx.set_stream(stream)
        return ctx.detect_batch(ptr, mem, fmt, w, h, rs, fs, n, out_cap)


def _detections(markers, per) -> List[Detection]:
    out, pos = [], 0
    for f in range(len(per)):
        det = Detection()
        for m in markers[pos: pos + int(per[f])]:
            c = m["corners"]
            det.markers.append(Marker(int(m["id"]), int(m["code"]), [(int(c[2 * i]), int(c[2 * i + 1])) for i in range(4)], int(m["hamming_distance"])))
        pos += int(per[f])
        out.append(det)
    return out


class BatchQueue:
    """Several batches in flight -- the Python twin of `BatchQueue` in integration/aruco3_hip.rs (additive API; public entry points of
    include/aruco3_hip.h only).  `depth` contexts of its own are used in rotation, each on a stream of its own: `submit` hands a batch
    over and returns at once, `collect` waits for the OLDEST batch in flight and returns its detections (markers only).

    gates=False (default): a free-running rotation -- with a batch of its own per context the fastest arrangement measured (DESIGN.md
    section 4.4).  gates=True: before each submit context k calls a3_order_after for the contexts k+1 .. depth-1 (bursts): the library
    holds the chain of every member but the last behind its threshold kernel; `last_stepping` says what it did with the batch just
    collected.  Results never depend on any of it.  Frames must stay valid and unmodified until their batch is collected (the queue
    keeps a reference to what it was handed)."""

    def __init__(self, detector: Detector, depth: int = 4, gates: bool = False):
        if not 1 <= depth <= 8:
            raise ValueError("depth must be in 1..8")
        d = detector.dictionary
        self._ctxs = [_lib.Context(detector.config._c(), d.code_list, d.num_bits, d._tau, detector.device) for _ in range(depth)]
        self._keep = [None] * depth
        self._gates = gates
        self._head = self._in_flight = self._submitted = 0
        self.last_stepping = None

    def __len__(self) -> int:
        return self._in_flight

    @property
    def full(self) -> bool:
        return self._in_flight == len(self._ctxs)

    def submit(self, images, out_cap: int = 0) -> None:
        if self.full:
            raise RuntimeError("BatchQueue is full: collect() the oldest batch first")
        depth = len(self._ctxs)
        k = self._submitted % depth
        ctx = self._ctxs[k]
        if self._gates:
            for m in range(k + 1, depth):
                ctx.order_after(self._ctxs[m])
        ptr, mem, fmt, w, h, rs, fs, n, keep = _as_frames(images)
        ctx.set_debug_taps(False)
        ctx.submit(ptr, mem, fmt, w, h, rs, fs, n, out_cap=out_cap or n * 64)
        self._keep[k] = keep
        self._submitted += 1
        self._in_flight += 1

    def collect(self) -> List[Detection]:
        if not self._in_flight:
            raise RuntimeError("BatchQueue.collect with nothing in flight")
        k = self._head
        ctx = self._ctxs[k]
        self._head = (self._head + 1) % len(self._ctxs)
        self._in_flight -= 1
        try:
            markers, per = ctx.collect()
        finally:
            self._keep[k] = None
        self.last_stepping = ctx.stats()["stepping"]
        return _detections(markers, per)

    def close(self) -> None:
        while self._in_flight:
            self.collect()
        for c in self._ctxs:
            c.close()
        self._ctxs = []
