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


def _case(rng):
    """one random case -> (config dict, dictionary name, pixel format name, frames [n, h, w, c] in R,G,B(,A) order)"""
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
    cfg, name, fmt, frames = _case(rng)
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
