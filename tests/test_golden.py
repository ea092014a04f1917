"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

CPU: the oracle still reproduces them (a change to the restatement is noticed).
GPU: the HIP path reproduces them bit for bit through the C ABI."""
from pathlib import Path

import numpy as np
import pytest

from tests.util import markers_of_hip

GOLDEN = sorted(Path(__file__).resolve().parent.joinpath("golden").glob("*.npz"))


def _load(path):
    z = np.load(path)
    g = {k: z[k] for k in z.files}
    h, w = g["thr_shape"]
    g["thresholded"] = (np.unpackbits(g["thresholded"], axis=1)[:, :w] * 255).astype(np.uint8)
    g["dictionary"] = str(g["dictionary"])
    return g


@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_oracle_reproduces_golden(oracle, dicts, path):
    g = _load(path)
    d = dicts.new_from_named_dict(g["dictionary"])
    r = oracle.detect(g["image"], d.code_list, d.num_bits, d._tau)
    assert np.array_equal(r["grey"], g["grey"])
    assert np.array_equal(r["thresholded"], g["thresholded"])
    assert r["n_contours"] == int(g["n_contours"])
    assert r["candidates_pre"].tolist() == g["candidates_pre"].tolist()
    assert r["candidates"].tolist() == g["candidates"].tolist()
    assert np.array_equal(r["homographies"], g["homographies"])
    assert r["codes"].tolist() == g["codes"].tolist()
    assert [m["id"] for m in r["markers"]] == g["marker_id"].tolist()
    assert [m["corners"] for m in r["markers"]] == [list(map(tuple, q)) for q in g["marker_corners"].tolist()]
    if "truth_ids" in g and path.stem.startswith("c1"):
        assert sorted(g["marker_id"].tolist()) == sorted(g["truth_ids"].tolist())  # the rendered ids are what gets decoded


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=lambda p: p.stem)
def test_hip_reproduces_golden(dicts, path):
    from aruco3_amd import _lib
    from aruco3_amd.aruco import Detector, DetectorConfig

    g = _load(path)
    det = Detector(DetectorConfig.default(), dicts.new_from_named_dict(g["dictionary"]))
    ctx = det._context()
    ctx.set_debug_taps(True)
    img = np.ascontiguousarray(g["image"])
    h, w, c = img.shape
    fmt = {3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c]
    markers, per = ctx.detect_batch(img.ctypes.data, _lib.MEM_HOST, fmt, w, h, w * c, h * w * c, 1)
    assert np.array_equal(ctx.download_grey(0, w, h), g["grey"])
    assert np.array_equal(ctx.download_grey(0, w, h, thresholded=True), g["thresholded"])
    assert ctx.candidates(0, before_discard=True).tolist() == g["candidates_pre"].tolist()
    assert ctx.candidates(0).tolist() == g["candidates"].tolist()
    patches, ok, codes, dec = ctx.homographies(0)
    assert np.array_equal(patches, g["homographies"])
    assert ok.tolist() == g["homography_ok"].tolist() and dec.tolist() == g["decode_ok"].tolist()
    assert codes.tolist() == g["codes"].tolist()
    got = markers_of_hip(markers)
    assert [m[0] for m in got] == g["marker_id"].tolist()
    assert [m[1] for m in got] == g["marker_code"].tolist()
    assert [list(m[2]) for m in got] == [list(map(tuple, q)) for q in g["marker_corners"].tolist()]
    assert [m[3] for m in got] == g["marker_hamming"].tolist()
    assert [m[4] for m in got] == g["marker_rotation"].tolist()
    assert int(per[0]) == len(got)
