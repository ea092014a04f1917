"""Host-side mirror of `pub mod pose` (src/pose.rs): `MarkerPose` and the three solvers.
The arithmetic is the IPPE kernel behind a3_estimate_pose / a3_estimate_pose_normalized."""
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib
from .pinhole import CameraIntrinsics

_ctx = None


def _context() -> _lib.Context:
    global _ctx
    if _ctx is None:
        _ctx = _lib.Context(_lib.default_config(), np.zeros(1, dtype=np.uint64), 16, 1)
    return _ctx


@dataclass
class MarkerPose:
    """src/pose.rs:8-12"""
    error: float
    rotation: np.ndarray      # 3x3 float32
    translation: np.ndarray   # 3 float32

    # src/pose.rs:17-39 -- tiny per-point affine maps on the caller's points
    def apply_transform_to_points(self, points: Sequence[Tuple[float, float, float]]):
        p = np.asarray(points, dtype=np.float32).reshape(-1, 3)
        return [tuple(map(float, (self.rotation @ v + self.translation).astype(np.float32))) for v in p]

    def apply_inverse_transform_to_points(self, points: Sequence[Tuple[float, float, float]]):
        p = np.asarray(points, dtype=np.float32).reshape(-1, 3)
        return [tuple(map(float, (self.rotation.T @ (v - self.translation)).astype(np.float32))) for v in p]


def _pair(recs, i) -> Tuple[MarkerPose, MarkerPose]:
    out = []
    for r in (recs[2 * i], recs[2 * i + 1]):
        out.append(MarkerPose(float(r.error), np.array(r.rotation, dtype=np.float32).reshape(3, 3), np.array(r.translation, dtype=np.float32)))
    return out[0], out[1]


def solve_with_intrinsics(image_points, marker_size_mm: float, camera_intrinsics: CameraIntrinsics):
    """src/pose.rs:52-55"""
    ci = camera_intrinsics
    intr = _lib.Intrinsics(ci.image_width, ci.image_height, ci.focal_x, ci.focal_y, ci.principal_x, ci.principal_y)
    recs = _context().estimate_pose(np.asarray(image_points, dtype=np.uint32).reshape(1, 8), marker_size_mm, None, intr)
    return _pair(recs, 0)


def solve_with_undistorted_points(image_points, marker_size_mm: float, image_size: Tuple[int, int]):
    """src/pose.rs:59-62"""
    recs = _context().estimate_pose(np.asarray(image_points, dtype=np.uint32).reshape(1, 8), marker_size_mm, image_size, None)
    return _pair(recs, 0)


def solve_with_normalized_points(normalized_image_points, marker_size_mm: float):
    """src/pose.rs:64-81"""
    recs = _context().estimate_pose_normalized(np.asarray(normalized_image_points, dtype=np.float32).reshape(1, 8), marker_size_mm)
    return _pair(recs, 0)


def solve_batch(corners: np.ndarray, marker_size_mm: float, image_size=None, intrinsics: CameraIntrinsics = None) -> List[Tuple[MarkerPose, MarkerPose]]:
    """All markers of a batch in one launch (what BASELINE config 5 needs right after detect)."""
    c = np.asarray(corners, dtype=np.uint32).reshape(-1, 8)
    intr = None
    if intrinsics is not None:
        ci = intrinsics
        intr = _lib.Intrinsics(ci.image_width, ci.image_height, ci.focal_x, ci.focal_y, ci.principal_x, ci.principal_y)
    recs = _context().estimate_pose(c, marker_size_mm, image_size, intr)
    return [_pair(recs, i) for i in range(c.shape[0])]


def estimate_pose(image_size: Tuple[int, int], corners, marker_size_mm: float, intrinsics: CameraIntrinsics = None):
    """The README's `estimate_pose((w,h), &corners, size, None)` facade (README.md:34)."""
    if intrinsics is not None:
        return solve_with_intrinsics(corners, marker_size_mm, intrinsics)
    return solve_with_undistorted_points(corners, marker_size_mm, image_size)
