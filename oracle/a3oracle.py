"""ctypes binding of the CPU oracle (oracle/a3_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, from __graft_entry__.smoke() and
from bench.py's cpu_baseline leg -- never from aruco3_amd/.  See a3_oracle.h for the
parity status ("parity unpinned" for the third-party image stages).
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "liba3oracle.so"

FMT_RGB8, FMT_RGBA8, FMT_L8 = 0, 1, 2


def build(force: bool = False) -> Path:
    src = _HERE / "a3_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < max(src.stat().st_mtime, (_HERE / "a3_oracle.h").stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "liba3oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class Config(C.Structure):
    _fields_ = [
        ("threshold_window", C.c_uint32),
        ("contour_simplification_epsilon", C.c_double),
        ("min_side_length_factor", C.c_float),
        ("min_corner_separation_factor", C.c_float),
        ("homography_sample_size", C.c_uint32),
        ("filter_high_bit_errors", C.c_uint8),
    ]

    @classmethod
    def default(cls):
        return cls(7, 0.05, 0.2, 0.1, 49, 1)


class Marker(C.Structure):
    _fields_ = [
        ("id", C.c_uint32),
        ("rotation", C.c_uint32),
        ("code", C.c_uint64),
        ("corners", C.c_uint32 * 8),
        ("hamming_distance", C.c_uint32),
        ("candidate_index", C.c_uint32),
    ]


class Pose(C.Structure):
    _fields_ = [("error", C.c_float), ("rotation", C.c_float * 9), ("translation", C.c_float * 3)]

    def as_tuple(self):
        return float(self.error), np.array(self.rotation, dtype=np.float32).reshape(3, 3), np.array(self.translation, dtype=np.float32)


class Contours(C.Structure):
    _fields_ = [
        ("offsets", C.POINTER(C.c_uint32)),
        ("points", C.POINTER(C.c_uint32)),
        ("border_type", C.POINTER(C.c_uint8)),
        ("parent", C.POINTER(C.c_int32)),
        ("n_contours", C.c_uint32),
    ]


class Detection(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("grey", C.POINTER(C.c_uint8)),
        ("thresholded", C.POINTER(C.c_uint8)),
        ("n_contours", C.c_uint32),
        ("n_contour_points", C.c_uint64),
        ("stat_reject_point_count", C.c_uint32), ("stat_reject_convexity", C.c_uint32), ("stat_reject_edge_length", C.c_uint32),
        ("n_candidates_pre", C.c_uint32),
        ("candidates_pre", C.POINTER(C.c_uint32)),
        ("candidates_pre_start", C.POINTER(C.c_uint32)),
        ("n_candidates", C.c_uint32),
        ("candidates", C.POINTER(C.c_uint32)),
        ("homographies", C.POINTER(C.c_uint8)),
        ("homography_ok", C.POINTER(C.c_uint8)),
        ("sample", C.c_uint32),
        ("decode_ok", C.POINTER(C.c_int32)),
        ("codes", C.POINTER(C.c_uint64)),
        ("n_markers", C.c_uint32),
        ("markers", C.POINTER(Marker)),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(str(_LIB_PATH))
        u8p, u32p, u64p, f32p = C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)
        L.a3o_hamming_distance.restype = C.c_uint32
        L.a3o_hamming_distance.argtypes = [C.c_uint64, C.c_uint64]
        L.a3o_calculate_tau.restype = C.c_uint8
        L.a3o_calculate_tau.argtypes = [u64p, C.c_size_t]
        L.a3o_mark_size.restype = C.c_uint8
        L.a3o_mark_size.argtypes = [C.c_uint8]
        L.a3o_find_nearest.restype = None
        L.a3o_find_nearest.argtypes = [u64p, C.c_size_t, C.c_uint64, C.POINTER(C.c_size_t), u8p]
        L.a3o_make_binary_image.restype = C.c_uint8
        L.a3o_make_binary_image.argtypes = [C.c_uint64, C.c_uint8, u8p]
        L.a3o_to_luma8.restype = None
        L.a3o_to_luma8.argtypes = [u8p, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, u8p]
        L.a3o_adaptive_threshold.restype = None
        L.a3o_adaptive_threshold.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.a3o_find_contours.restype = C.c_int
        L.a3o_find_contours.argtypes = [u8p, C.c_uint32, C.c_uint32, C.POINTER(Contours)]
        L.a3o_free_contours.restype = None
        L.a3o_free_contours.argtypes = [C.POINTER(Contours)]
        L.a3o_approximate_polygon_dp.restype = C.c_size_t
        L.a3o_approximate_polygon_dp.argtypes = [u32p, C.c_size_t, C.c_double, C.c_int, u32p]
        L.a3o_convex_hull.restype = C.c_size_t
        L.a3o_convex_hull.argtypes = [u32p, C.c_size_t, u32p]
        L.a3o_from_control_points.restype = C.c_int
        L.a3o_from_control_points.argtypes = [f32p, f32p, f32p, f32p]
        L.a3o_warp_into.restype = None
        L.a3o_warp_into.argtypes = [u8p, C.c_uint32, C.c_uint32, f32p, u8p, C.c_uint32, C.c_uint32]
        L.a3o_otsu_level.restype = C.c_uint8
        L.a3o_otsu_level.argtypes = [u8p, C.c_uint32, C.c_uint32]
        L.a3o_resize_triangle.restype = None
        L.a3o_resize_triangle.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u8p]
        L.a3o_enforce_clockwise_corners.restype = None
        L.a3o_enforce_clockwise_corners.argtypes = [u32p, C.c_size_t]
        L.a3o_discard_too_near.restype = C.c_size_t
        L.a3o_discard_too_near.argtypes = [u32p, C.c_size_t, C.c_float, u32p]
        L.a3o_rotate_bit_matrix.restype = None
        L.a3o_rotate_bit_matrix.argtypes = [u8p, C.c_uint32, C.c_uint32, u8p]
        L.a3o_homography_to_code_permutations.restype = C.c_int
        L.a3o_homography_to_code_permutations.argtypes = [u8p, C.c_uint32, C.c_uint32, C.c_uint8, u64p]
        L.a3o_detect.restype = C.c_int
        L.a3o_detect.argtypes = [C.POINTER(Config), u64p, C.c_size_t, C.c_uint8, C.c_uint8, u8p, C.c_int, C.c_uint32, C.c_uint32,
                                 C.c_size_t, C.c_int, C.POINTER(Detection)]
        L.a3o_detect_quads.restype = C.c_int
        L.a3o_detect_quads.argtypes = [C.POINTER(Config), u64p, C.c_size_t, C.c_uint8, C.c_uint8, u8p, C.c_int, C.c_uint32, C.c_uint32,
                                       C.c_size_t, C.c_int, C.POINTER(C.c_uint32), C.c_size_t, C.POINTER(Detection)]
        L.a3o_free_detection.restype = None
        L.a3o_free_detection.argtypes = [C.POINTER(Detection)]
        L.a3o_make_marker_square.restype = None
        L.a3o_make_marker_square.argtypes = [C.c_float, f32p]
        L.a3o_compute_homography_from_marker_square.restype = None
        L.a3o_compute_homography_from_marker_square.argtypes = [C.c_float, f32p, f32p]
        L.a3o_solve_canonical_form.restype = None
        L.a3o_solve_canonical_form.argtypes = [f32p, f32p, f32p, C.POINTER(Pose), C.POINTER(Pose)]
        L.a3o_solve_with_normalized_points.restype = None
        L.a3o_solve_with_normalized_points.argtypes = [f32p, C.c_float, C.POINTER(Pose), C.POINTER(Pose)]
        L.a3o_solve_with_undistorted_points.restype = None
        L.a3o_solve_with_undistorted_points.argtypes = [u32p, C.c_float, C.c_uint32, C.c_uint32, C.POINTER(Pose), C.POINTER(Pose)]
        L.a3o_solve_with_intrinsics.restype = None
        L.a3o_solve_with_intrinsics.argtypes = [u32p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(Pose), C.POINTER(Pose)]
        L.a3o_apply_transform.restype = None
        L.a3o_apply_transform.argtypes = [C.POINTER(Pose), f32p, C.c_size_t, f32p]
        L.a3o_apply_inverse_transform.restype = None
        L.a3o_apply_inverse_transform.argtypes = [C.POINTER(Pose), f32p, C.c_size_t, f32p]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


# ---- numpy-level helpers used by the tests -------------------------------------------

def hamming_distance(a: int, b: int) -> int:
    return int(lib().a3o_hamming_distance(a, b))


def find_nearest(codes: np.ndarray, bits: int):
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    idx, dist = C.c_size_t(), C.c_uint8()
    lib().a3o_find_nearest(_p(codes, C.c_uint64), codes.size, bits, C.byref(idx), C.byref(dist))
    return int(idx.value), int(dist.value)


def calculate_tau(codes: np.ndarray) -> int:
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    return int(lib().a3o_calculate_tau(_p(codes, C.c_uint64), codes.size))


def mark_size(num_bits: int) -> int:
    return int(lib().a3o_mark_size(num_bits))


def make_binary_image(code: int, num_bits: int):
    w = mark_size(num_bits)
    cells = np.zeros(w * w + 2 * w, dtype=np.uint8)
    w2 = lib().a3o_make_binary_image(code, num_bits, _p(cells, C.c_uint8))
    assert w2 == w
    return w, cells[: w * w].reshape(w, w).copy()


def _fmt_of(img: np.ndarray) -> int:
    if img.ndim == 2:
        return FMT_L8
    return {3: FMT_RGB8, 4: FMT_RGBA8, 1: FMT_L8}[img.shape[2]]


def to_luma8(img: np.ndarray) -> np.ndarray:
    img = _u8(img)
    h, w = img.shape[:2]
    out = np.empty((h, w), dtype=np.uint8)
    lib().a3o_to_luma8(_p(img, C.c_uint8), _fmt_of(img), w, h, img.strides[0], _p(out, C.c_uint8))
    return out


def adaptive_threshold(grey: np.ndarray, radius: int = 7) -> np.ndarray:
    grey = _u8(grey)
    h, w = grey.shape
    out = np.empty_like(grey)
    lib().a3o_adaptive_threshold(_p(grey, C.c_uint8), w, h, radius, _p(out, C.c_uint8))
    return out


def find_contours(binary: np.ndarray):
    """-> list of (n,2) uint32 arrays (x,y), border types, parents."""
    binary = _u8(binary)
    h, w = binary.shape
    cs = Contours()
    rc = lib().a3o_find_contours(_p(binary, C.c_uint8), w, h, C.byref(cs))
    assert rc == 0
    n = cs.n_contours
    offs = np.ctypeslib.as_array(cs.offsets, shape=(n + 1,)).copy()
    total = int(offs[n])
    pts = np.ctypeslib.as_array(cs.points, shape=(max(total, 1) * 2,)).copy()[: total * 2].reshape(-1, 2)
    bt = np.ctypeslib.as_array(cs.border_type, shape=(max(n, 1),)).copy()[:n]
    par = np.ctypeslib.as_array(cs.parent, shape=(max(n, 1),)).copy()[:n]
    lib().a3o_free_contours(C.byref(cs))
    return [pts[offs[i]: offs[i + 1]] for i in range(n)], bt, par


def approximate_polygon_dp(points: np.ndarray, epsilon: float, closed: bool = True) -> np.ndarray:
    points = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 2)
    out = np.empty((points.shape[0] + 2, 2), dtype=np.uint32)
    m = lib().a3o_approximate_polygon_dp(_p(points, C.c_uint32), points.shape[0], float(epsilon), int(closed), _p(out, C.c_uint32))
    return out[:m].copy()


def convex_hull(points: np.ndarray) -> np.ndarray:
    points = np.ascontiguousarray(points, dtype=np.uint32).reshape(-1, 2)
    out = np.empty((points.shape[0] + 1, 2), dtype=np.uint32)
    m = lib().a3o_convex_hull(_p(points, C.c_uint32), points.shape[0], _p(out, C.c_uint32))
    return out[:m].copy()


def from_control_points(frm, to):
    frm = np.ascontiguousarray(frm, dtype=np.float32).reshape(8)
    to = np.ascontiguousarray(to, dtype=np.float32).reshape(8)
    t = np.zeros(9, dtype=np.float32)
    inv = np.zeros(9, dtype=np.float32)
    ok = lib().a3o_from_control_points(_p(frm, C.c_float), _p(to, C.c_float), _p(t, C.c_float), _p(inv, C.c_float))
    return bool(ok), t, inv


def warp_into(grey: np.ndarray, mapping: np.ndarray, ow: int, oh: int) -> np.ndarray:
    grey = _u8(grey)
    h, w = grey.shape
    mapping = np.ascontiguousarray(mapping, dtype=np.float32).reshape(9)
    out = np.empty((oh, ow), dtype=np.uint8)
    lib().a3o_warp_into(_p(grey, C.c_uint8), w, h, _p(mapping, C.c_float), _p(out, C.c_uint8), ow, oh)
    return out


def otsu_level(img: np.ndarray) -> int:
    img = _u8(img)
    return int(lib().a3o_otsu_level(_p(img, C.c_uint8), img.shape[1], img.shape[0]))


def resize_triangle(img: np.ndarray, nw: int, nh: int) -> np.ndarray:
    img = _u8(img)
    out = np.empty((nh, nw), dtype=np.uint8)
    lib().a3o_resize_triangle(_p(img, C.c_uint8), img.shape[1], img.shape[0], nw, nh, _p(out, C.c_uint8))
    return out


def enforce_clockwise_corners(quads: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(quads, dtype=np.uint32).reshape(-1, 8).copy()
    lib().a3o_enforce_clockwise_corners(_p(q, C.c_uint32), q.shape[0])
    return q.reshape(-1, 4, 2)


def discard_too_near(quads: np.ndarray, min_distance: float):
    q = np.ascontiguousarray(quads, dtype=np.uint32).reshape(-1, 8).copy()
    kept = np.zeros(max(q.shape[0], 1), dtype=np.uint32)
    m = lib().a3o_discard_too_near(_p(q, C.c_uint32), q.shape[0], C.c_float(min_distance), _p(kept, C.c_uint32))
    return q[:m].reshape(-1, 4, 2).copy(), kept[:m].copy()


def rotate_bit_matrix(bits: np.ndarray) -> np.ndarray:
    b = _u8(bits)
    out = np.empty((b.shape[1], b.shape[0]), dtype=np.uint8)
    lib().a3o_rotate_bit_matrix(_p(b, C.c_uint8), b.shape[0], b.shape[1], _p(out, C.c_uint8))
    return out


def homography_to_code_permutations(patch: np.ndarray, mark_size_: int):
    patch = _u8(patch)
    codes = np.zeros(4, dtype=np.uint64)
    ok = lib().a3o_homography_to_code_permutations(_p(patch, C.c_uint8), patch.shape[1], patch.shape[0], mark_size_, _p(codes, C.c_uint64))
    return (codes if ok else None)


def detect(img: np.ndarray, codes: np.ndarray, num_bits: int, tau: int, config: Config = None, keep_debug: bool = True, quads=None) -> dict:
    """Run the whole restated Detector::detect and return every stage as numpy data.  `quads` (test aid, n x 4 x 2): the candidate
    list handed to discard_too_near / extract_homographies INSTEAD of what the contour stage found (quirk Q4's degenerate quads)."""
    cfg = config or Config.default()
    img = _u8(img)
    h, w = img.shape[:2]
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    d = Detection()
    if quads is None:
        rc = lib().a3o_detect(C.byref(cfg), _p(codes, C.c_uint64), codes.size, num_bits, tau, _p(img, C.c_uint8), _fmt_of(img), w, h,
                              img.strides[0], int(keep_debug), C.byref(d))
    else:
        q = np.ascontiguousarray(quads, dtype=np.uint32).reshape(-1, 8)
        rc = lib().a3o_detect_quads(C.byref(cfg), _p(codes, C.c_uint64), codes.size, num_bits, tau, _p(img, C.c_uint8), _fmt_of(img), w, h,
                                    img.strides[0], int(keep_debug), _p(q, C.c_uint32), q.shape[0], C.byref(d))
    assert rc == 0
    S = d.sample
    nc, npre, nm = d.n_candidates, d.n_candidates_pre, d.n_markers
    res = {
        "n_contours": int(d.n_contours),
        "n_contour_points": int(d.n_contour_points),
        "stats": (int(d.stat_reject_point_count), int(d.stat_reject_convexity), int(d.stat_reject_edge_length)),
        "candidates": np.ctypeslib.as_array(d.candidates, shape=(max(nc, 1) * 8,)).copy()[: nc * 8].reshape(-1, 4, 2),
        "homographies": np.ctypeslib.as_array(d.homographies, shape=(max(nc, 1) * S * S,)).copy()[: nc * S * S].reshape(-1, S, S),
        "homography_ok": np.ctypeslib.as_array(d.homography_ok, shape=(max(nc, 1),)).copy()[:nc],
        "decode_ok": np.ctypeslib.as_array(d.decode_ok, shape=(max(nc, 1),)).copy()[:nc],
        "codes": np.ctypeslib.as_array(d.codes, shape=(max(nc, 1) * 4,)).copy()[: nc * 4].reshape(-1, 4),
        "markers": [
            {
                "id": int(d.markers[i].id),
                "code": int(d.markers[i].code),
                "rotation": int(d.markers[i].rotation),
                "corners": [(int(d.markers[i].corners[2 * k]), int(d.markers[i].corners[2 * k + 1])) for k in range(4)],
                "hamming_distance": int(d.markers[i].hamming_distance),
                "candidate_index": int(d.markers[i].candidate_index),
            }
            for i in range(nm)
        ],
    }
    if keep_debug:
        res["grey"] = np.ctypeslib.as_array(d.grey, shape=(h, w)).copy()
        res["thresholded"] = np.ctypeslib.as_array(d.thresholded, shape=(h, w)).copy()
        res["candidates_pre"] = np.ctypeslib.as_array(d.candidates_pre, shape=(max(npre, 1) * 8,)).copy()[: npre * 8].reshape(-1, 4, 2)
        res["candidates_pre_start"] = np.ctypeslib.as_array(d.candidates_pre_start, shape=(max(npre, 1),)).copy()[:npre]
    lib().a3o_free_detection(C.byref(d))
    return res


def detect_markers_only(img: np.ndarray, codes: np.ndarray, num_bits: int, tau: int, config: Config = None) -> int:
    """Timed entry for bench.py's cpu_baseline: full detect, no debug copies; returns the marker count."""
    cfg = config or Config.default()
    h, w = img.shape[:2]
    d = Detection()
    rc = lib().a3o_detect(C.byref(cfg), _p(codes, C.c_uint64), codes.size, num_bits, tau, _p(img, C.c_uint8), _fmt_of(img), w, h,
                          img.strides[0], 0, C.byref(d))
    assert rc == 0
    n = int(d.n_markers)
    lib().a3o_free_detection(C.byref(d))
    return n


# ---- pose ------------------------------------------------------------------------------

def make_marker_square(size: float) -> np.ndarray:
    out = np.zeros(12, dtype=np.float32)
    lib().a3o_make_marker_square(size, _p(out, C.c_float))
    return out.reshape(4, 3)


def compute_homography_from_marker_square(size: float, pts) -> np.ndarray:
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(8)
    h = np.zeros(9, dtype=np.float32)
    lib().a3o_compute_homography_from_marker_square(size, _p(pts, C.c_float), _p(h, C.c_float))
    return h.reshape(3, 3)


def solve_canonical_form(obj, pts, h):
    obj = np.ascontiguousarray(obj, dtype=np.float32).reshape(12)
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(8)
    h = np.ascontiguousarray(h, dtype=np.float32).reshape(9)
    p1, p2 = Pose(), Pose()
    lib().a3o_solve_canonical_form(_p(obj, C.c_float), _p(pts, C.c_float), _p(h, C.c_float), C.byref(p1), C.byref(p2))
    return p1.as_tuple(), p2.as_tuple()


def solve_with_normalized_points(pts, size: float):
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(8)
    p1, p2 = Pose(), Pose()
    lib().a3o_solve_with_normalized_points(_p(pts, C.c_float), size, C.byref(p1), C.byref(p2))
    return p1.as_tuple(), p2.as_tuple()


def solve_with_undistorted_points(corners, size: float, image_size):
    c = np.ascontiguousarray(corners, dtype=np.uint32).reshape(8)
    p1, p2 = Pose(), Pose()
    lib().a3o_solve_with_undistorted_points(_p(c, C.c_uint32), size, image_size[0], image_size[1], C.byref(p1), C.byref(p2))
    return p1.as_tuple(), p2.as_tuple()


def solve_with_intrinsics(corners, size: float, fx, fy, cx, cy):
    c = np.ascontiguousarray(corners, dtype=np.uint32).reshape(8)
    p1, p2 = Pose(), Pose()
    lib().a3o_solve_with_intrinsics(_p(c, C.c_uint32), size, fx, fy, cx, cy, C.byref(p1), C.byref(p2))
    return p1.as_tuple(), p2.as_tuple()


def _pose_struct(rotation, translation) -> Pose:
    p = Pose()
    p.error = 0.0
    p.rotation = (C.c_float * 9)(*np.asarray(rotation, dtype=np.float32).reshape(9))
    p.translation = (C.c_float * 3)(*np.asarray(translation, dtype=np.float32).reshape(3))
    return p


def apply_transform(rotation, translation, pts, inverse=False) -> np.ndarray:
    pts = np.ascontiguousarray(pts, dtype=np.float32).reshape(-1, 3)
    out = np.empty_like(pts)
    p = _pose_struct(rotation, translation)
    fn = lib().a3o_apply_inverse_transform if inverse else lib().a3o_apply_transform
    fn(C.byref(p), _p(pts, C.c_float), pts.shape[0], _p(out, C.c_float))
    return out
