"""ctypes binding of libaruco3_hip.so (the C ABI of include/aruco3_hip.h).

There is no CPU implementation behind this module: if the shared library is missing, or
no MI355X is visible, every entry point raises.  Build with `python __graft_entry__.py`
(or `make -C aruco3_amd/csrc`).
"""
import ctypes as C
from pathlib import Path

import numpy as np

import os

_HERE = Path(__file__).resolve().parent
# A3_HIP_LIB: the sweep scripts under tools/ point this at a `make tuning` build (build/tuning/libaruco3_hip.so, -DA3_TUNING);
# everything else loads the product library next to this file.
LIB_PATH = Path(os.environ["A3_HIP_LIB"]).resolve() if os.environ.get("A3_HIP_LIB") else _HERE / "libaruco3_hip.so"

OK, ERR_INVALID, ERR_HIP, ERR_CAPACITY, ERR_INTERNAL, ERR_NO_DEVICE, ERR_LIMIT = 0, -1, -2, -3, -4, -5, -6
FMT_RGB8, FMT_RGBA8, FMT_L8, FMT_BGRA8 = 0, 1, 2, 3
MEM_HOST, MEM_DEVICE = 0, 1
PROFILE_OFF, PROFILE_STAGES, PROFILE_THRESHOLD_ONLY, PROFILE_THRESHOLD_SAMPLED = 0, 1, 2, 3
STAGE_THRESHOLD, STAGE_CONTOUR, STAGE_DECODE = 0, 1, 2
# a3_stats.stepping & 0xFF (include/aruco3_hip.h A3_STEP_*)
STEP_WHOLE, STEP_DECODE_DEFERRED, STEP_HELD_RELEASED_BY_LAST, STEP_HELD_RELEASED_EARLY, STEP_BURST_LAST, STEP_HELD = 0, 1, 2, 3, 4, 5
STEP_NAMES = {0: "whole", 1: "decode_deferred", 2: "held_released_by_last", 3: "held_released_early", 4: "burst_last", 5: "held"}

# every symbol include/aruco3_hip.h declares
SYMBOLS = [
    "a3_abi_version", "a3_default_config", "a3_create", "a3_destroy", "a3_last_error", "a3_set_stream", "a3_get_stream", "a3_set_pool_limits",
    "a3_get_tau", "a3_order_after", "a3_set_debug_taps", "a3_detect_batch", "a3_detect_batch_pose", "a3_detect_batch_submit", "a3_detect_batch_collect", "a3_detect_batch_pose_submit", "a3_detect_batch_pose_collect",
    "a3_host_alloc", "a3_host_free", "a3_host_register", "a3_host_unregister", "a3_get_stats", "a3_synth_render", "a3_download_grey", "a3_download_thresholded",
    "a3_candidate_count", "a3_download_candidates", "a3_download_homographies", "a3_estimate_pose", "a3_estimate_pose_normalized",
    "a3_find_nearest", "a3_calculate_tau", "a3_set_profiling", "a3_get_profile",
    "a3_contour_count", "a3_download_contours", "a3_detection_record_bytes", "a3_pack_detections",
]
# aruco3_amd/csrc/a3_internal.h: probes and single-stage hooks for this repository's tests and tools, not for bindings
INTERNAL_SYMBOLS = ["a3_debug_set_k1_stream", "a3_debug_set_overlap", "a3_debug_set_k1_waves", "a3_debug_set_partition", "a3_debug_build_flags", "a3_debug_spin", "a3_debug_set_mark_threshold", "a3_debug_set_hold", "a3_debug_launch_threshold", "a3_debug_stream_wait_threshold", "a3_debug_kernel_time", "a3_selftest_ieee", "a3_debug_clockwise", "a3_debug_rotate_bits", "a3_debug_discard_too_near", "a3_debug_inject_candidates"]


class A3Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"aruco3_hip error {code}: {message}")
        self.code = code


class Config(C.Structure):
    _fields_ = [
        ("threshold_window", C.c_uint32),
        ("contour_simplification_epsilon", C.c_double),
        ("min_side_length_factor", C.c_float),
        ("min_corner_separation_factor", C.c_float),
        ("homography_sample_size", C.c_uint32),
        ("filter_high_bit_errors", C.c_uint8),
    ]


class MarkerRec(C.Structure):
    _fields_ = [
        ("frame", C.c_uint32),
        ("id", C.c_uint32),
        ("code", C.c_uint64),
        ("corners", C.c_uint32 * 8),
        ("hamming_distance", C.c_uint8),
        ("rotation", C.c_uint8),
        ("candidate_index", C.c_uint16),
    ]


MARKER_DTYPE = np.dtype([("frame", "<u4"), ("id", "<u4"), ("code", "<u8"), ("corners", "<u4", (8,)), ("hamming_distance", "u1"),
                         ("rotation", "u1"), ("candidate_index", "<u2")], align=True)
assert MARKER_DTYPE.itemsize == C.sizeof(MarkerRec) == 56


class PoseRec(C.Structure):
    _fields_ = [("error", C.c_float), ("rotation", C.c_float * 9), ("translation", C.c_float * 3)]


class Intrinsics(C.Structure):
    _fields_ = [("image_width", C.c_uint32), ("image_height", C.c_uint32), ("focal_x", C.c_float), ("focal_y", C.c_float),
                ("principal_x", C.c_float), ("principal_y", C.c_float)]


class Stats(C.Structure):
    _fields_ = [("darts", C.c_uint64), ("contours_traced", C.c_uint64), ("contours_materialised", C.c_uint64),
                ("candidates_pre", C.c_uint64), ("candidates", C.c_uint64), ("markers", C.c_uint64),
                ("resolve_iterations", C.c_uint32), ("jump_rounds", C.c_uint32), ("chunks", C.c_uint32), ("stepping", C.c_uint32)]

    def as_dict(self):
        d = {k: int(getattr(self, k)) for k, _ in self._fields_}
        d["released_others"] = (d["stepping"] >> 8) & 0xFF     # (a3_stats.stepping: bits 8-15 = chains of other contexts this batch's submit released)
        d["reruns"] = (d["stepping"] >> 16) & 0xFF              # (bits 16-23: synchronous re-runs the device asked for)
        d["stepping"] = STEP_NAMES.get(d["stepping"] & 0xFF, d["stepping"] & 0xFF)
        return d


_lib = None


def load():
    """dlopen the library and declare the prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP/HSA runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64,
    # and a second copy (from /opt/rocm) initialised in the same process finds no device.  Importing torch first
    # makes the dynamic linker bind this library's NEEDED libamdhip64.so.7 to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:  # a host without PyTorch: the system ROCm runtime is used
        pass
    if not LIB_PATH.exists():
        raise ImportError(f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
                          "aruco3_amd has no CPU fallback.")
    L = C.CDLL(str(LIB_PATH))
    vp, u8p, u32p, u64p, f32p, f64p = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float), C.POINTER(C.c_double)
    L.a3_abi_version.restype = C.c_int
    L.a3_default_config.restype = None
    L.a3_default_config.argtypes = [C.POINTER(Config)]
    L.a3_create.restype = C.c_int
    L.a3_create.argtypes = [C.c_int, C.POINTER(Config), u64p, C.c_size_t, C.c_uint8, C.c_uint8, C.POINTER(vp)]
    L.a3_destroy.restype = None
    L.a3_destroy.argtypes = [vp]
    L.a3_last_error.restype = C.c_char_p
    L.a3_last_error.argtypes = [vp]
    L.a3_set_stream.restype = C.c_int
    L.a3_set_stream.argtypes = [vp, vp]
    L.a3_get_stream.restype = C.c_int
    L.a3_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.a3_detect_batch_pose_submit.restype = C.c_int
    L.a3_detect_batch_pose_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_uint32, C.c_float,
                                              C.POINTER(Intrinsics), C.c_size_t]
    L.a3_detect_batch_pose_collect.restype = C.c_int
    L.a3_detect_batch_pose_collect.argtypes = [vp, vp, vp, C.c_size_t, u32p, C.POINTER(C.c_size_t)]
    L.a3_order_after.restype = C.c_int
    L.a3_order_after.argtypes = [vp, vp]
    L.a3_host_alloc.restype = C.c_int
    L.a3_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    L.a3_host_free.restype = C.c_int
    L.a3_host_free.argtypes = [vp]
    L.a3_host_register.restype = C.c_int
    L.a3_host_register.argtypes = [vp, C.c_size_t]
    L.a3_host_unregister.restype = C.c_int
    L.a3_host_unregister.argtypes = [vp]
    L.a3_set_pool_limits.restype = C.c_int
    L.a3_set_pool_limits.argtypes = [vp, C.c_uint64, C.c_uint64]
    L.a3_get_tau.restype = C.c_int
    L.a3_get_tau.argtypes = [vp, u8p]
    L.a3_set_debug_taps.restype = C.c_int
    L.a3_set_debug_taps.argtypes = [vp, C.c_int]
    L.a3_detect_batch.restype = C.c_int
    L.a3_detect_batch.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_uint32, vp, C.c_size_t, u32p,
                                  C.POINTER(C.c_size_t)]
    L.a3_detect_batch_submit.restype = C.c_int
    L.a3_detect_batch_submit.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_uint32, C.c_size_t]
    L.a3_detect_batch_collect.restype = C.c_int
    L.a3_detect_batch_collect.argtypes = [vp, vp, C.c_size_t, u32p, C.POINTER(C.c_size_t)]
    L.a3_detect_batch_pose.restype = C.c_int
    L.a3_detect_batch_pose.argtypes = [vp, vp, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_size_t, C.c_size_t, C.c_uint32, C.c_float,
                                       C.POINTER(Intrinsics), vp, vp, C.c_size_t, u32p, C.POINTER(C.c_size_t)]
    L.a3_synth_render.restype = C.c_int
    L.a3_synth_render.argtypes = [C.c_int, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_float, C.c_int,
                                  vp, C.c_size_t, C.c_size_t]
    if hasattr(L, "a3_debug_set_overlap"):      # (older builds loaded through A3_HIP_LIB for A/B runs lack it)
        L.a3_debug_set_overlap.restype = C.c_int
        L.a3_debug_set_overlap.argtypes = [C.c_int]
    if hasattr(L, "a3_debug_set_k1_stream"):
        L.a3_debug_set_k1_stream.restype = C.c_int
        L.a3_debug_set_k1_stream.argtypes = [C.c_int]
    if hasattr(L, "a3_debug_set_k1_waves"):
        L.a3_debug_set_k1_waves.restype = C.c_int
        L.a3_debug_set_k1_waves.argtypes = [C.c_int]
    if hasattr(L, "a3_debug_launch_threshold"):
        L.a3_debug_launch_threshold.restype = C.c_int
        L.a3_debug_launch_threshold.argtypes = [vp, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
    if hasattr(L, "a3_debug_set_hold"):
        L.a3_debug_set_hold.restype = C.c_int
        L.a3_debug_set_hold.argtypes = [C.c_int]
    if hasattr(L, "a3_debug_set_mark_threshold"):
        L.a3_debug_set_mark_threshold.restype = C.c_int
        L.a3_debug_set_mark_threshold.argtypes = [C.c_int]
        L.a3_debug_stream_wait_threshold.restype = C.c_int
        L.a3_debug_stream_wait_threshold.argtypes = [vp, vp]
    if hasattr(L, "a3_debug_spin"):
        L.a3_debug_spin.restype = C.c_int
        L.a3_debug_spin.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    if hasattr(L, "a3_debug_build_flags"):
        L.a3_debug_build_flags.restype = C.c_int
        L.a3_debug_build_flags.argtypes = []
    if hasattr(L, "a3_debug_set_partition"):
        L.a3_debug_set_partition.restype = C.c_int
        L.a3_debug_set_partition.argtypes = [C.c_int, C.c_int]
    L.a3_debug_kernel_time.restype = C.c_int
    L.a3_debug_kernel_time.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.a3_get_stats.restype = C.c_int
    L.a3_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.a3_download_grey.restype = C.c_int
    L.a3_download_grey.argtypes = [vp, C.c_uint32, u8p]
    L.a3_download_thresholded.restype = C.c_int
    L.a3_download_thresholded.argtypes = [vp, C.c_uint32, u8p]
    L.a3_candidate_count.restype = C.c_int
    L.a3_candidate_count.argtypes = [vp, C.c_uint32, u32p, u32p]
    L.a3_download_candidates.restype = C.c_int
    L.a3_download_candidates.argtypes = [vp, C.c_uint32, C.c_int, u32p, C.c_size_t]
    L.a3_download_homographies.restype = C.c_int
    L.a3_download_homographies.argtypes = [vp, C.c_uint32, u8p, u8p, u64p, C.POINTER(C.c_int32), C.c_size_t]
    L.a3_estimate_pose.restype = C.c_int
    L.a3_estimate_pose.argtypes = [vp, u32p, C.c_size_t, C.c_float, C.POINTER(Intrinsics), C.c_uint32, C.c_uint32, C.POINTER(PoseRec)]
    L.a3_estimate_pose_normalized.restype = C.c_int
    L.a3_estimate_pose_normalized.argtypes = [vp, f32p, C.c_size_t, C.c_float, C.POINTER(PoseRec)]
    L.a3_find_nearest.restype = C.c_int
    L.a3_find_nearest.argtypes = [vp, u64p, C.c_size_t, u32p, u8p]
    L.a3_calculate_tau.restype = C.c_int
    L.a3_calculate_tau.argtypes = [C.c_int, u64p, C.c_size_t, u8p]
    L.a3_set_profiling.restype = C.c_int
    L.a3_set_profiling.argtypes = [vp, C.c_int]
    L.a3_get_profile.restype = C.c_int
    L.a3_get_profile.argtypes = [vp, C.c_int, f64p, u64p, C.c_int]
    L.a3_selftest_ieee.restype = C.c_int
    L.a3_selftest_ieee.argtypes = [vp, f64p, f64p, C.c_size_t, f64p, f64p, f32p, f32p]
    L.a3_contour_count.restype = C.c_int
    L.a3_contour_count.argtypes = [vp, C.c_uint32, u32p, u64p]
    L.a3_download_contours.restype = C.c_int
    L.a3_download_contours.argtypes = [vp, C.c_uint32, u32p, u32p, u32p, C.c_size_t, C.c_size_t]
    L.a3_detection_record_bytes.restype = C.c_size_t
    L.a3_detection_record_bytes.argtypes = [C.c_uint32, C.c_int]
    L.a3_pack_detections.restype = C.c_int
    L.a3_pack_detections.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int, vp, C.c_size_t]
    L.a3_debug_clockwise.restype = C.c_int
    L.a3_debug_clockwise.argtypes = [vp, C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32)]
    L.a3_debug_rotate_bits.restype = C.c_int
    L.a3_debug_rotate_bits.argtypes = [vp, u8p, C.c_uint32, C.c_uint32, u8p]
    L.a3_debug_inject_candidates.restype = C.c_int
    L.a3_debug_inject_candidates.argtypes = [vp, u32p, C.c_size_t]
    L.a3_debug_discard_too_near.restype = C.c_int
    L.a3_debug_discard_too_near.argtypes = [vp, u32p, C.c_size_t, C.c_float, u32p, C.POINTER(C.c_size_t)]
    _lib = L
    return L


def library_info() -> dict:
    """which shared library this process runs on (bench.py and the GPU tests put it into their output: a leftover A3_HIP_LIB
    pointing at a tuning build must not pass for the product)"""
    L = load()
    flags = int(L.a3_debug_build_flags()) if hasattr(L, "a3_debug_build_flags") else -1
    return {"path": str(LIB_PATH), "from_A3_HIP_LIB": bool(os.environ.get("A3_HIP_LIB")), "tuning_build": bool(flags & 1) if flags >= 0 else None,
            "non_default_kernel_build": bool(flags & 2) if flags >= 0 else None, "abi": int(L.a3_abi_version())}


class PinnedBuffer:
    """a3_host_alloc as a numpy uint8 array (`.array`): frames placed here cross the link asynchronously and at its full rate"""

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        rc = load().a3_host_alloc(nbytes, C.byref(p))
        if rc != OK:
            raise A3Error(rc, load().a3_last_error(None).decode("utf-8", "replace"))
        self.ptr, self.nbytes = p.value, nbytes
        self.array = np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(p.value))

    def close(self):
        if getattr(self, "ptr", None):
            self.array = None
            load().a3_host_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def check(rc, ctx=None):
    if rc != OK:
        raise A3Error(rc, load().a3_last_error(ctx).decode("utf-8", "replace"))


def default_config() -> Config:
    cfg = Config()
    load().a3_default_config(C.byref(cfg))
    return cfg


def synth_render(device: int, frames: np.ndarray, markers: np.ndarray, width: int, height: int, paper: bool, black: float, white: float,
                 supersample: int, out_ptr: int, row_stride: int = 0, frame_stride: int = 0, stream: int = 0) -> None:
    """a3_synth_render: frames / markers are the record arrays of aruco3_amd.synth.device_layout, out_ptr a device pointer."""
    frames = np.ascontiguousarray(frames)
    markers = np.ascontiguousarray(markers)
    rc = load().a3_synth_render(device, C.c_void_p(stream), frames.ctypes.data_as(C.c_void_p), len(frames),
                                markers.ctypes.data_as(C.c_void_p) if len(markers) else None, len(markers), width, height, int(paper),
                                black, white, supersample, C.c_void_p(out_ptr), row_stride, frame_stride)
    if rc != 0:
        raise A3Error(rc, "a3_synth_render failed")


class Context:
    """Owns one a3_ctx (one device, one stream)."""

    def __init__(self, config: Config, codes: np.ndarray, num_bits: int, tau: int, device: int = 0):
        L = load()
        self._codes = np.ascontiguousarray(codes, dtype=np.uint64)
        h = C.c_void_p()
        rc = L.a3_create(device, C.byref(config), _p(self._codes, C.c_uint64), self._codes.size, num_bits, tau, C.byref(h))
        check(rc, None)
        self.handle = h
        self.device = device
        self.sample = int(config.homography_sample_size)

    def close(self):
        if getattr(self, "handle", None):
            load().a3_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def tau(self) -> int:
        t = C.c_uint8()
        check(load().a3_get_tau(self.handle, C.byref(t)), self.handle)
        return int(t.value)

    def set_stream(self, stream_ptr: int):
        check(load().a3_set_stream(self.handle, C.c_void_p(stream_ptr)), self.handle)

    @property
    def stream_ptr(self) -> int:
        """hipStream_t the context enqueues on (a3_get_stream): wrap it with torch.cuda.ExternalStream to order torch work after it"""
        p = C.c_void_p()
        check(load().a3_get_stream(self.handle, C.byref(p)), self.handle)
        return int(p.value or 0)

    def order_after(self, other: "Context"):
        """a3_order_after: this context's next batch starts only after `other`'s work in flight has finished (burst stepping)"""
        check(load().a3_order_after(self.handle, other.handle), self.handle)

    def set_debug_taps(self, on: bool):
        check(load().a3_set_debug_taps(self.handle, int(on)), self.handle)

    def set_pool_limits(self, max_darts: int = 0, max_points: int = 0):
        check(load().a3_set_pool_limits(self.handle, max_darts, max_points), self.handle)

    def set_profiling(self, mode):
        """False / 0 off, True / 1 every stage, PROFILE_THRESHOLD_ONLY (2): the threshold stage only (two event records per batch)"""
        check(load().a3_set_profiling(self.handle, int(mode)), self.handle)

    def profile(self, stage: int, reset: bool = False):
        ms, n = C.c_double(), C.c_uint64()
        check(load().a3_get_profile(self.handle, stage, C.byref(ms), C.byref(n), int(reset)), self.handle)
        return float(ms.value), int(n.value)

    def debug_kernel_time(self, kernel: int, dbg: int = 0, reps: int = 5) -> float:
        ms = C.c_float()
        check(load().a3_debug_kernel_time(self.handle, kernel, dbg, reps, C.byref(ms)), self.handle)
        return ms.value

    def stats(self) -> dict:
        s = Stats()
        check(load().a3_get_stats(self.handle, C.byref(s)), self.handle)
        return s.as_dict()

    def detect_batch(self, pixels_ptr: int, memory: int, fmt: int, width: int, height: int, row_stride: int, frame_stride: int, n_frames: int,
                     out_cap: int = 0):
        """-> (markers structured array, per-frame counts)"""
        cap = out_cap or max(64 * n_frames, 64)
        out = np.empty(cap, dtype=MARKER_DTYPE)   # only the first n records are written and returned
        per = np.zeros(max(n_frames, 1), dtype=np.uint32)
        n = C.c_size_t()
        rc = load().a3_detect_batch(self.handle, C.c_void_p(pixels_ptr), memory, fmt, width, height, row_stride, frame_stride, n_frames,
                                    out.ctypes.data_as(C.c_void_p), cap, _p(per, C.c_uint32), C.byref(n))
        check(rc, self.handle)
        return out[: n.value], per[:n_frames]

    def submit(self, pixels_ptr: int, memory: int, fmt: int, width: int, height: int, row_stride: int, frame_stride: int, n_frames: int,
               out_cap: int = 0):
        """Enqueue a batch without waiting (a3_detect_batch_submit); `collect()` returns what detect_batch would."""
        self._pending = (out_cap or max(64 * n_frames, 64), n_frames)
        check(load().a3_detect_batch_submit(self.handle, C.c_void_p(pixels_ptr), memory, fmt, width, height, row_stride, frame_stride, n_frames,
                                            self._pending[0]), self.handle)

    def collect(self):
        cap, n_frames = self._pending
        out = np.empty(cap, dtype=MARKER_DTYPE)
        per = np.zeros(max(n_frames, 1), dtype=np.uint32)
        n = C.c_size_t()
        check(load().a3_detect_batch_collect(self.handle, out.ctypes.data_as(C.c_void_p), cap, _p(per, C.c_uint32), C.byref(n)), self.handle)
        return out[: n.value], per[:n_frames]

    def detect_batch_pose(self, pixels_ptr: int, memory: int, fmt: int, width: int, height: int, row_stride: int, frame_stride: int,
                          n_frames: int, marker_size_mm: float, intrinsics: "Intrinsics" = None, out_cap: int = 0):
        """-> (markers, per-frame counts, poses float32 [n_markers, 2, 13] = error, 9 rotation row-major, 3 translation)"""
        cap = out_cap or max(64 * n_frames, 64)
        out = np.zeros(cap, dtype=MARKER_DTYPE)
        poses = np.zeros((cap, 2, 13), dtype=np.float32)
        per = np.zeros(max(n_frames, 1), dtype=np.uint32)
        n = C.c_size_t()
        rc = load().a3_detect_batch_pose(self.handle, C.c_void_p(pixels_ptr), memory, fmt, width, height, row_stride, frame_stride, n_frames,
                                         marker_size_mm, C.byref(intrinsics) if intrinsics else None, out.ctypes.data_as(C.c_void_p),
                                         poses.ctypes.data_as(C.c_void_p), cap, _p(per, C.c_uint32), C.byref(n))
        check(rc, self.handle)
        return out[: n.value], per[:n_frames], poses[: n.value]

    def submit_pose(self, pixels_ptr: int, memory: int, fmt: int, width: int, height: int, row_stride: int, frame_stride: int, n_frames: int,
                    marker_size_mm: float, intrinsics: "Intrinsics" = None, out_cap: int = 0):
        """a3_detect_batch_pose_submit; `collect_pose()` returns what detect_batch_pose would."""
        self._pending = (out_cap or max(64 * n_frames, 64), n_frames)
        check(load().a3_detect_batch_pose_submit(self.handle, C.c_void_p(pixels_ptr), memory, fmt, width, height, row_stride, frame_stride, n_frames,
                                                 marker_size_mm, C.byref(intrinsics) if intrinsics else None, self._pending[0]), self.handle)

    def collect_pose(self):
        cap, n_frames = self._pending
        out = np.empty(cap, dtype=MARKER_DTYPE)
        poses = np.zeros((cap, 2, 13), dtype=np.float32)
        per = np.zeros(max(n_frames, 1), dtype=np.uint32)
        n = C.c_size_t()
        check(load().a3_detect_batch_pose_collect(self.handle, out.ctypes.data_as(C.c_void_p), poses.ctypes.data_as(C.c_void_p), cap,
                                                  _p(per, C.c_uint32), C.byref(n)), self.handle)
        return out[: n.value], per[:n_frames], poses[: n.value]

    # ---- Detection.grey / thresholded / candidates / homographies of the last batch ----
    def download_grey(self, frame: int, w: int, h: int, thresholded: bool = False) -> np.ndarray:
        out = np.empty((h, w), dtype=np.uint8)
        fn = load().a3_download_thresholded if thresholded else load().a3_download_grey
        check(fn(self.handle, frame, _p(out, C.c_uint8)), self.handle)
        return out

    def candidates(self, frame: int, before_discard: bool = False) -> np.ndarray:
        a, b = C.c_uint32(), C.c_uint32()
        check(load().a3_candidate_count(self.handle, frame, C.byref(a), C.byref(b)), self.handle)
        cnt = a.value if before_discard else b.value
        out = np.zeros((max(cnt, 1), 4, 2), dtype=np.uint32)
        check(load().a3_download_candidates(self.handle, frame, int(before_discard), _p(out, C.c_uint32), max(cnt, 1)), self.handle)
        return out[:cnt]

    def homographies(self, frame: int, with_patches: bool = True):
        a, b = C.c_uint32(), C.c_uint32()
        check(load().a3_candidate_count(self.handle, frame, C.byref(a), C.byref(b)), self.handle)
        cnt, S = b.value, self.sample
        patches = np.zeros((max(cnt, 1), S, S), dtype=np.uint8)
        ok = np.zeros(max(cnt, 1), dtype=np.uint8)
        codes = np.zeros((max(cnt, 1), 4), dtype=np.uint64)
        dec = np.zeros(max(cnt, 1), dtype=np.int32)
        check(load().a3_download_homographies(self.handle, frame, _p(patches, C.c_uint8) if with_patches else None, _p(ok, C.c_uint8),
                                              _p(codes, C.c_uint64), _p(dec, C.c_int32), max(cnt, 1)), self.handle)
        return patches[:cnt], ok[:cnt], codes[:cnt], dec[:cnt]

    def contours(self, frame: int):
        """find_contours of one frame of the last batch (debug taps on): -> (start_keys u32[n], list of int32 [len, 2] point arrays)
        in the reference's discovery order."""
        nc, npts = C.c_uint32(), C.c_uint64()
        check(load().a3_contour_count(self.handle, frame, C.byref(nc), C.byref(npts)), self.handle)
        keys = np.zeros(max(nc.value, 1), dtype=np.uint32)
        lens = np.zeros(max(nc.value, 1), dtype=np.uint32)
        pts = np.zeros((max(npts.value, 1), 2), dtype=np.uint32)
        check(load().a3_download_contours(self.handle, frame, _p(keys, C.c_uint32), _p(lens, C.c_uint32), _p(pts, C.c_uint32),
                                          max(nc.value, 1), max(npts.value, 1)), self.handle)
        keys, lens = keys[: nc.value], lens[: nc.value]
        offs = np.concatenate([[0], np.cumsum(lens, dtype=np.int64)])
        return keys, [pts[offs[i]: offs[i + 1]].astype(np.int64) for i in range(nc.value)]

    def pack_detections(self, first_frame_global: int, max_markers: int, dst_ptr: int, dst_bytes: int, with_poses: bool = False):
        """a3_pack_detections: the last batch's markers (and, after detect_batch_pose, their poses) as fixed-capacity per-frame
        records in device memory at dst_ptr (enqueued on the context's stream)."""
        check(load().a3_pack_detections(self.handle, first_frame_global, max_markers, int(with_poses), C.c_void_p(dst_ptr), dst_bytes), self.handle)

    # ---- a3_internal.h: the reference's small helpers on the device, for its own vectors ----
    def debug_clockwise(self, quads: np.ndarray) -> np.ndarray:
        q = np.ascontiguousarray(quads, dtype=np.int32).reshape(-1, 8)
        out = np.zeros_like(q)
        check(load().a3_debug_clockwise(self.handle, _p(q, C.c_int32), q.shape[0], _p(out, C.c_int32)), self.handle)
        return out.reshape(-1, 4, 2)

    def debug_rotate_bits(self, bits: np.ndarray, times: int = 1) -> np.ndarray:
        b = np.ascontiguousarray(bits, dtype=np.uint8)
        n = b.shape[0]
        assert b.shape == (n, n)
        out = np.zeros_like(b)
        check(load().a3_debug_rotate_bits(self.handle, _p(b, C.c_uint8), n, times, _p(out, C.c_uint8)), self.handle)
        return out

    def debug_inject_candidates(self, quads: np.ndarray):
        """the next one-frame batch decodes THESE quads (n x 4 x 2, in this order) instead of what its contour stage finds"""
        q = np.ascontiguousarray(quads, dtype=np.uint32).reshape(-1, 8)
        check(load().a3_debug_inject_candidates(self.handle, _p(q, C.c_uint32), q.shape[0]), self)

    def debug_discard_too_near(self, quads: np.ndarray, min_distance: float) -> np.ndarray:
        q = np.ascontiguousarray(quads, dtype=np.uint32).reshape(-1, 8)
        out = np.zeros_like(q)
        n = C.c_size_t()
        check(load().a3_debug_discard_too_near(self.handle, _p(q, C.c_uint32), q.shape[0], min_distance, _p(out, C.c_uint32), C.byref(n)),
              self.handle)
        return out[: n.value].reshape(-1, 4, 2)

    # ---- pose ----
    def estimate_pose(self, corners: np.ndarray, marker_size_mm: float, image_size=None, intrinsics: Intrinsics = None) -> np.ndarray:
        c = np.ascontiguousarray(corners, dtype=np.uint32).reshape(-1, 8)
        out = (PoseRec * (2 * max(c.shape[0], 1)))()
        iw, ih = image_size if image_size else (1, 1)
        check(load().a3_estimate_pose(self.handle, _p(c, C.c_uint32), c.shape[0], marker_size_mm, C.byref(intrinsics) if intrinsics else None,
                                      iw, ih, out), self.handle)
        return out

    def estimate_pose_normalized(self, points: np.ndarray, marker_size_mm: float):
        p = np.ascontiguousarray(points, dtype=np.float32).reshape(-1, 8)
        out = (PoseRec * (2 * max(p.shape[0], 1)))()
        check(load().a3_estimate_pose_normalized(self.handle, _p(p, C.c_float), p.shape[0], marker_size_mm, out), self.handle)
        return out

    def find_nearest(self, bits: np.ndarray):
        b = np.ascontiguousarray(bits, dtype=np.uint64)
        idx = np.zeros(max(b.size, 1), dtype=np.uint32)
        dist = np.zeros(max(b.size, 1), dtype=np.uint8)
        check(load().a3_find_nearest(self.handle, _p(b, C.c_uint64), b.size, _p(idx, C.c_uint32), _p(dist, C.c_uint8)), self.handle)
        return idx[: b.size], dist[: b.size]

    def selftest_ieee(self, a: np.ndarray, b: np.ndarray):
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = np.ascontiguousarray(b, dtype=np.float64)
        n = a.size
        sq, dv = np.zeros(n), np.zeros(n)
        sqf, dvf = np.zeros(n, dtype=np.float32), np.zeros(n, dtype=np.float32)
        check(load().a3_selftest_ieee(self.handle, _p(a, C.c_double), _p(b, C.c_double), n, _p(sq, C.c_double), _p(dv, C.c_double),
                                      _p(sqf, C.c_float), _p(dvf, C.c_float)), self.handle)
        return sq, dv, sqf, dvf


# ---- dictionary helpers used by ARDictionary (device kernels; no host arithmetic) ----
_dict_ctx = {}


def _ctx_for(codes: np.ndarray) -> Context:
    # keyed by the table's contents: an address can be re-used by another array once the first one is freed
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    key = codes.tobytes()
    ctx = _dict_ctx.get(key)
    if ctx is None:
        ctx = Context(default_config(), codes, 64, 1)
        _dict_ctx[key] = ctx
    return ctx


def find_nearest(codes: np.ndarray, bits: np.ndarray):
    return _ctx_for(codes).find_nearest(bits)


def calculate_tau(codes: np.ndarray) -> int:
    codes = np.ascontiguousarray(codes, dtype=np.uint64)
    t = C.c_uint8()
    check(load().a3_calculate_tau(0, _p(codes, C.c_uint64), codes.size, C.byref(t)), None)
    return int(t.value)
