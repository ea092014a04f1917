"""Randomised combinations, on the GPU: every DetectorConfig field (src/aruco.rs:23-30) x dictionary x pixel format x odd frame
size x content, every stage through the C ABI against the oracle run with the same configuration.  The other parity tests move
one knob at a time; a product path that only breaks when two are away from their defaults (a 36-bit dictionary sampled at 31x31
from BGRA frames of width 333 with a 9-pixel threshold window ...) would pass them.
Also a soak driver: `python tests/test_gpu_config_fuzz.py [cases] [first_seed]` on the GPU box."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from tests.util import assert_frame_parity, markers_of_hip, markers_of_oracle  # noqa: E402

DICTS = ("ARUCO", "ARUCO_MIP_16H3", "APRILTAG_16H5", "APRILTAG_25H9", "ARUCO_MIP_36H12", "APRILTAG_36H11", "ARTOOLKITPLUS", "CHILITAGS")


def _case(rng, big_windows=False):
    """one random case -> (config dict, dictionary name, pixel format name, frames [n, h, w, c] in R,G,B(,A) order).
    big_windows: threshold windows 8..33, every other case on a width that is a multiple of 16: the fused kernel of windows
    8..31 (k_threshold_big.hip) with its vector loads and with its per-pixel loads; 32 and 33 go through the separable path."""
    from aruco3_amd import synth
    from aruco3_amd.dictionaries import ARDictionary

    cfg = dict(
        threshold_window=int(rng.choice([7, 7, 7, 3, 5, 9, 12])),
        contour_simplification_epsilon=float(rng.choice([0.05, 0.05, 0.02, 0.03, 0.08, 0.12])),
        min_side_length_factor=float(rng.choice([0.2, 0.2, 0.05, 0.1, 0.35])),
        min_corner_separation_factor=float(rng.choice([0.1, 0.1, 0.02, 0.05, 0.25])),
        homography_sample_size=int(rng.choice([49, 49, 8, 10, 21, 31, 64, 100])),
        filter_high_bit_errors=bool(rng.integers(0, 2)),
    )
    name = DICTS[int(rng.integers(0, len(DICTS)))]
    d = ARDictionary.new_from_named_dict(name)
    fmt = ("RGB8", "RGB8", "RGBA8", "BGRA8", "L8")[int(rng.integers(0, 5))]
    w, h = int(rng.integers(200, 900)), int(rng.integers(160, 700))
    if big_windows:
        cfg["threshold_window"] = int(rng.integers(8, 34))
        if rng.integers(0, 2):
            w = (w + 15) // 16 * 16
        if not rng.integers(0, 6):
            w, h = int(rng.choice([1008, 1040, 1280])), int(rng.integers(60, 200))   # one and two column strips of 992
    n = int(rng.integers(1, 4))
    kind = int(rng.integers(0, 4))
    frames = []
    for i in range(n):
        if kind == 3:   # the reference bench's recipe (benches/detect_markers.rs:33-45)
            frames.append(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8))
            continue
        side_hi = max(40.0, min(w, h) * float(rng.uniform(0.25, 0.6)))
        spec = synth.SynthSpec(w, h, n_markers=(1, 3), side=(side_hi * 0.6, side_hi), min_center_sep=side_hi * 1.3,
                               perspective=float(rng.uniform(0.0, 0.2)), noise_sigma=float(rng.choice([0.0, 0.0, 3.0, 8.0])),
                               background=("flat", "gradient")[int(rng.integers(0, 2))], paper=bool(rng.integers(0, 2)),
                               supersample=2)
        frames.append(synth.render_frame(spec, d.code_list, d.num_bits, int(rng.integers(0, 2 ** 31)))[0])
    rgb = np.stack(frames)
    if fmt == "L8":
        rgb = np.repeat(rgb[..., :1], 3, axis=3)          # what the L8 frame decodes to: grey = the one channel
        return cfg, name, fmt, np.ascontiguousarray(rgb[..., 0][..., None])
    if fmt in ("RGBA8", "BGRA8"):
        a = rng.integers(0, 256, size=rgb.shape[:3] + (1,), dtype=np.uint8)   # alpha is ignored (into_luma8)
        return cfg, name, fmt, np.ascontiguousarray(np.concatenate([rgb, a], axis=3))
    return cfg, name, fmt, np.ascontiguousarray(rgb)


def run_case(oracle, seed):
    from aruco3_amd import _lib
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    rng = np.random.default_rng(seed)
    cfg, name, fmt, frames = _case(rng, big_windows=seed >= 1000000)
    d = ARDictionary.new_from_named_dict(name)
    det = Detector(DetectorConfig(**cfg), d)
    ctx = det._context()
    ocfg = oracle.Config.default()
    for k, v in cfg.items():
        setattr(ocfg, k, int(v) if isinstance(v, bool) else v)
    n, h, w, c = frames.shape
    dev = frames[..., [2, 1, 0, 3]].copy() if fmt == "BGRA8" else frames     # the bytes the library is handed
    code = {"RGB8": _lib.FMT_RGB8, "RGBA8": _lib.FMT_RGBA8, "BGRA8": _lib.FMT_BGRA8, "L8": _lib.FMT_L8}[fmt]
    what = f"seed {seed}: {name} {fmt} {w}x{h} x{n} {cfg}"
    try:
        ctx.set_debug_taps(False)
        m0, p0 = ctx.detect_batch(dev.ctypes.data, _lib.MEM_HOST, code, w, h, w * c, h * w * c, n)
        ctx.set_debug_taps(True)
        m1, p1 = ctx.detect_batch(dev.ctypes.data, _lib.MEM_HOST, code, w, h, w * c, h * w * c, n)
        assert np.array_equal(p0, p1) and np.array_equal(m0, m1), "tapped and untapped runs differ"
        pos = 0
        for f in range(n):
            img = frames[f] if c > 1 else frames[f][..., 0]
            res = oracle.detect(img, d.code_list, d.num_bits, ctx.tau, config=ocfg)
            assert_frame_parity(ctx, f, img, res, w, h)
            assert markers_of_hip(m1[pos: pos + int(p1[f])]) == markers_of_oracle(res), "markers differ"
            pos += int(p1[f])
        assert pos == len(m1)
    except AssertionError as e:
        raise AssertionError(f"{what}: {e}") from e
    return len(m1)


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(4))
def test_random_config_dictionary_format_combinations(oracle, block):
    found = sum(run_case(oracle, 9000 + 12 * block + i) for i in range(12))
    assert found >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(2))
def test_random_combinations_with_windows_8_to_16(oracle, block):
    found = sum(run_case(oracle, 1000000 + 10 * block + i) for i in range(10))
    assert found >= 0


def _projected_squares(rng, n, w, h):
    """corners (u32, as a Marker carries them) of squares seen by a pinhole camera at random poses: the inputs k_pose meets"""
    out = np.zeros((n, 8), dtype=np.uint32)
    f = 0.9 * w
    for i in range(n):
        while True:
            size = float(rng.uniform(20.0, 120.0))
            ax = rng.normal(size=3); ax /= np.linalg.norm(ax)
            ang = float(rng.uniform(0.0, 1.2))
            K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
            R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
            R = R @ np.array([[np.cos(a := float(rng.uniform(0, 6.283))), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
            t = np.array([rng.uniform(-0.3, 0.3) * 1000, rng.uniform(-0.2, 0.2) * 1000, rng.uniform(300.0, 2500.0)])
            sq = np.array([[-1, 1, 0], [1, 1, 0], [1, -1, 0], [-1, -1, 0]], dtype=np.float64) * size / 2
            pc = sq @ R.T + t
            uv = np.stack([f * pc[:, 0] / pc[:, 2] + w / 2, f * pc[:, 1] / pc[:, 2] + h / 2], axis=1)
            if (pc[:, 2] > 50).all() and (uv >= 0).all() and (uv[:, 0] < w).all() and (uv[:, 1] < h).all():
                out[i] = np.round(uv).astype(np.uint32).reshape(8)
                break
    return out


def _same_floats(got, want, what):
    got, want = np.asarray(got, dtype=np.float64).ravel(), np.asarray(want, dtype=np.float64).ravel()
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN pattern differs"
    ok = ~np.isnan(want)
    fin = ok & np.isfinite(want)
    assert np.array_equal(got[ok & ~fin], want[ok & ~fin]), f"{what}: infinities differ"
    tol = 1e-4 * np.maximum(1.0, np.abs(want[fin]))     # north_star's pose tolerance, relative above 1
    bad = np.abs(got[fin] - want[fin]) > tol
    assert not bad.any(), f"{what}: {got[fin][bad][:4]} vs {want[fin][bad][:4]}"


@pytest.mark.gpu
def test_pose_solvers_on_random_quads(oracle):
    """src/pose.rs:52-81 through the device: squares projected at random poses (rounded to the integer corners a Marker has), and
    quads no camera would produce (collinear, repeated and wildly skewed corners): both solutions, their errors, with the image
    size and with explicit intrinsics, against the oracle within 1e-4 (relative above 1); NaNs where the oracle has NaNs."""
    from aruco3_amd import _lib
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    rng = np.random.default_rng(4711)
    ctx = Detector(DetectorConfig(), ARDictionary.new_from_named_dict("ARUCO"))._context()
    w, h = 1920, 1080
    quads = _projected_squares(rng, 600, w, h)
    wild = rng.integers(0, 1080, size=(200, 8)).astype(np.uint32)
    wild[:20, 2:4] = wild[:20, 0:2]                                    # a repeated corner
    wild[20:40] = np.stack([np.arange(8, dtype=np.uint32) * 10 + i for i in range(20)])   # collinear
    corners = np.concatenate([quads, wild])
    size = 40.0
    intr = _lib.Intrinsics(w, h, 1500.0, 1480.0, 955.5, 541.25)
    for label, recs, ref in (
        ("image size", ctx.estimate_pose(corners, size, (w, h)), lambda c: oracle.solve_with_undistorted_points(c, size, (w, h))),
        ("intrinsics", ctx.estimate_pose(corners, size, (w, h), intr), lambda c: oracle.solve_with_intrinsics(c, size, 1500.0, 1480.0, 955.5, 541.25)),
    ):
        for i in range(corners.shape[0]):
            a, b = ref(corners[i])
            for k, want in enumerate((a, b)):
                r = recs[2 * i + k]
                what = f"{label}, quad {i} ({corners[i].tolist()}), solution {k}"
                _same_floats([r.error], [want[0]], what + " error")
                _same_floats(np.array(r.rotation), want[1], what + " rotation")
                _same_floats(np.array(r.translation), want[2], what + " translation")


if __name__ == "__main__":
    import time

    from oracle import a3oracle

    a3oracle.build()
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    t0 = time.time(); found = 0
    for s in range(first, first + cases):
        found += run_case(a3oracle, s)
        if (s - first) % 50 == 49:
            print(f"{s - first + 1} cases ok, {found} markers so far, {time.time() - t0:.0f} s", flush=True)
    print(f"config fuzz ok: {cases} cases, {found} markers, {time.time() - t0:.0f} s")
