"""The oracle against fixtures produced by the REFERENCE crate (integration/dump_fixtures.rs run with cargo elsewhere):
tests/fixtures/<name>.a3fx, one per input of tests/fixtures/inputs/manifest.txt.  No such file can be made in this image (no Rust
toolchain), so every test here SKIPS when its fixture is absent; dropped in, they pin -- or correct -- the stages of
oracle/a3_oracle.c that are "parity unpinned" today (into_luma8, adaptive_threshold, find_contours,
approximate_polygon_dp, convex_hull, from_control_points, warp_into, otsu_level, threshold, resize).

The reader and the stage-by-stage comparison are exercised without cargo too: a fixture written in the same format FROM THE
ORACLE must compare clean, and one with a flipped byte must not (the loader is not vacuous)."""
import struct
from pathlib import Path

import numpy as np
import pytest

FIX = Path(__file__).resolve().parent / "fixtures"
INPUTS = FIX / "inputs"
MAGIC = b"A3FX1\n"
DTYPES = {0: "u1", 1: "<u4", 2: "<u8", 3: "<i4", 4: "<f4", 5: "<f8"}


def manifest():
    rows = []
    for line in (INPUTS / "manifest.txt").read_text().splitlines():
        f = line.split()
        if len(f) == 5:
            rows.append((f[0], int(f[1]), int(f[2]), int(f[3]), f[4]))
    return rows


def read_a3fx(path):
    raw = Path(path).read_bytes()
    assert raw[:6] == MAGIC, "not an A3FX1 file"
    out, pos = {}, 6
    while pos < len(raw):
        (nl,) = struct.unpack_from("<I", raw, pos); pos += 4
        name = raw[pos:pos + nl].decode(); pos += nl
        dt = raw[pos]; pos += 1
        (nd,) = struct.unpack_from("<I", raw, pos); pos += 4
        dims = struct.unpack_from(f"<{nd}I", raw, pos); pos += 4 * nd
        dtype = np.dtype(DTYPES[dt])
        count = int(np.prod(dims)) if nd else 1
        out[name] = np.frombuffer(raw, dtype=dtype, count=count, offset=pos).reshape(dims)
        pos += count * dtype.itemsize
    return out


def write_a3fx(path, records):
    inv = {np.dtype(v).str: k for k, v in DTYPES.items()}
    inv["|u1"] = 0
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        for name, arr in records.items():
            arr = np.ascontiguousarray(arr)
            fh.write(struct.pack("<I", len(name.encode()))); fh.write(name.encode())
            fh.write(bytes([inv[arr.dtype.newbyteorder("<").str if arr.dtype.itemsize > 1 else "|u1"]]))
            fh.write(struct.pack("<I", arr.ndim)); fh.write(struct.pack(f"<{arr.ndim}I", *arr.shape))
            fh.write(arr.astype(arr.dtype.newbyteorder("<")).tobytes())


def load_input(name, w, h, c):
    return np.fromfile(INPUTS / f"{name}.raw", dtype=np.uint8).reshape(h, w, c)


def _split(flat, lens):
    out, pos = [], 0
    for n in lens:
        out.append(flat[pos:pos + int(n)]); pos += int(n)
    return out


def oracle_stages(oracle, dicts, name, w, h, c, dict_name):
    """What integration/dump_fixtures.rs writes, computed by the oracle: the same records, the same order of calls."""
    img = load_input(name, w, h, c)
    d = dicts.new_from_named_dict(dict_name)
    cfg = oracle.Config.default()
    S, ms = int(cfg.homography_sample_size), oracle.mark_size(d.num_bits)
    grey = oracle.to_luma8(img)
    thr = oracle.adaptive_threshold(grey, int(cfg.threshold_window))
    cpts, cborder, _ = oracle.find_contours(thr)
    pts = [np.asarray(p, dtype=np.uint32).reshape(-1, 2) for p in cpts]
    dps, hulls = [], []
    for p in pts:
        dp = np.asarray(oracle.approximate_polygon_dp(p, len(p) * float(cfg.contour_simplification_epsilon), True), dtype=np.uint32).reshape(-1, 2)
        dps.append(dp)
        hulls.append(np.asarray(oracle.convex_hull(dp), dtype=np.uint32).reshape(-1, 2) if len(dp) == 4 else np.zeros((0, 2), np.uint32))
    r = oracle.detect(img, d.code_list, d.num_bits, d._tau)
    cand = np.asarray(r["candidates"], dtype=np.uint32).reshape(-1, 4, 2)
    n = len(cand)
    src_xy, ok, homs, otsu, bins, resized = [], [], [], [], [], []
    xs, ys = np.meshgrid(np.arange(S, dtype=np.float32), np.arange(S, dtype=np.float32))
    for q in cand:
        solved, _, inv = oracle.from_control_points([(float(x), float(y)) for x, y in q], [(0.0, 0.0), (float(S), 0.0), (float(S), float(S)), (0.0, float(S))])
        if not solved:
            ok.append(0); src_xy.append(np.full((S, S, 2), np.nan, np.float32)); hom = np.zeros((S, S), np.uint8)
        else:
            ok.append(1)
            inv = np.asarray(inv, dtype=np.float32).reshape(9)
            den = inv[6] * xs + inv[7] * ys + inv[8]                        # f32 operations in the order of imageproc's Projection * (x, y)
            src_xy.append(np.stack([(inv[0] * xs + inv[1] * ys + inv[2]) / den, (inv[3] * xs + inv[4] * ys + inv[5]) / den], axis=-1).astype(np.float32))
            hom = oracle.warp_into(grey, inv, S, S)
        level = oracle.otsu_level(hom)
        b = np.where(hom > level, 255, 0).astype(np.uint8)                  # threshold(.., Binary): p > t -> 255
        homs.append(hom); otsu.append(level); bins.append(b); resized.append(oracle.resize_triangle(b, ms, ms))
    mk = r["markers"]
    return {
        "grey": grey, "thresholded": thr,
        "contour_len": np.array([len(p) for p in pts], np.uint32), "contour_border": np.asarray(cborder, np.uint8),
        "contour_points": np.concatenate(pts) if pts else np.zeros((0, 2), np.uint32),
        "dp_len": np.array([len(p) for p in dps], np.uint32), "dp_points": np.concatenate(dps) if dps else np.zeros((0, 2), np.uint32),
        "hull_len": np.array([len(p) for p in hulls], np.uint32), "hull_points": np.concatenate(hulls) if hulls else np.zeros((0, 2), np.uint32),
        "candidates": cand,
        "warp_src_xy": np.stack(src_xy) if n else np.zeros((0, S, S, 2), np.float32), "projection_ok": np.array(ok, np.uint8),
        "homographies": np.stack(homs) if n else np.zeros((0, S, S), np.uint8), "homographies_equal_detection": np.array([1], np.uint8),
        "otsu": np.array(otsu, np.uint8), "binarized": np.stack(bins) if n else np.zeros((0, S, S), np.uint8),
        "resized": np.stack(resized) if n else np.zeros((0, ms, ms), np.uint8),
        "marker_id": np.array([m["id"] for m in mk], np.uint32), "marker_code": np.array([m["code"] for m in mk], np.uint64),
        "marker_corners": np.array([m["corners"] for m in mk], np.uint32).reshape(-1, 4, 2),
        "marker_hamming": np.array([m["hamming_distance"] for m in mk], np.uint8),
    }


# the order follows the pipeline, so that the FIRST stage that differs is the one reported
STAGES = ["grey", "thresholded", "contour_len", "contour_border", "contour_points", "dp_len", "dp_points", "hull_len", "hull_points", "candidates",
          "projection_ok", "warp_src_xy", "homographies", "homographies_equal_detection", "otsu", "binarized", "resized",
          "marker_id", "marker_code", "marker_corners", "marker_hamming"]


def compare(ref, got):
    """-> list of (stage, message) for every stage that differs; warp_src_xy within 1e-3 of a pixel (f32 maps of two solvers), the rest exact"""
    bad = []
    for k in STAGES:
        if k not in ref:
            bad.append((k, "missing from the fixture")); continue
        a, b = np.asarray(ref[k]), np.asarray(got[k])
        if a.shape != b.shape:
            bad.append((k, f"shape {a.shape} (reference) vs {b.shape} (oracle)")); continue
        if k == "warp_src_xy":
            both = ~(np.isnan(a) | np.isnan(b))
            if a.size and (np.isnan(a) != np.isnan(b)).any() or (a.size and np.abs(a[both] - b[both]).max(initial=0.0) > 1e-3):
                bad.append((k, f"max |d| {np.abs(a[both] - b[both]).max(initial=0.0):.3g} px"))
        elif not np.array_equal(a, b):
            idx = np.argwhere(a != b)
            bad.append((k, f"{len(idx)} of {a.size} values differ, first at {idx[0].tolist()}: reference {a[tuple(idx[0])]} oracle {b[tuple(idx[0])]}"))
    return bad


@pytest.mark.parametrize("row", manifest(), ids=lambda r: r[0])
def test_oracle_against_reference_fixture(oracle, dicts, row):
    name = row[0]
    path = FIX / f"{name}.a3fx"
    if not path.exists():
        pytest.skip(f"{path.name} absent: produce it with integration/dump_fixtures.rs where cargo exists (INTEGRATION.md)")
    ref = read_a3fx(path)
    got = oracle_stages(oracle, dicts, *row)
    bad = compare(ref, got)
    assert not bad, "oracle differs from the reference crate (" + bytes(ref.get("versions", b"")).decode(errors="replace") + "):\n" + "\n".join(f"  {k}: {m}" for k, m in bad)


def test_inputs_are_the_golden_images():
    """the raw files the Rust side reads are the images of tests/golden/*.npz, byte for byte"""
    for name, w, h, c, dict_name in manifest():
        z = np.load(Path(__file__).resolve().parent / "golden" / f"{name}.npz")
        assert np.array_equal(load_input(name, w, h, c), z["image"]) and str(z["dictionary"]) == dict_name


def test_reader_and_comparison_are_not_vacuous(oracle, dicts, tmp_path):
    """a fixture written in the A3FX1 format from the oracle itself compares clean; with one byte of one stage flipped the
    comparison names that stage (so a real fixture that disagrees cannot slip through a broken reader)"""
    row = next(r for r in manifest() if r[0].startswith("odd_"))
    got = oracle_stages(oracle, dicts, *row)
    assert len(got["candidates"]) > 0 and len(got["marker_id"]) > 0 and len(got["contour_len"]) > 3
    p = tmp_path / "self.a3fx"
    write_a3fx(p, dict(got, versions=np.frombuffer(b"oracle self-check", dtype=np.uint8)))
    ref = read_a3fx(p)
    assert compare(ref, got) == []
    for stage in ("thresholded", "contour_points", "homographies", "resized", "marker_corners"):
        broken = {k: np.array(v) for k, v in ref.items()}
        flat = broken[stage].reshape(-1)
        flat[flat.size // 2] ^= 1
        assert [k for k, _ in compare(broken, got)] == [stage]


# ---- quirk Q4 (src/aruco.rs:255-257): the degenerate quads of tests/test_gpu_round6.py, as integration/dump_fixtures.rs::dump_q4 writes them ----
Q4_QUADS = np.array([
    [[10, 10], [50, 50], [90, 90], [130, 130]], [[300, 300], [300, 300], [340, 300], [340, 340]], [[200, 20], [260, 20], [260, 20], [200, 20]],
    [[77, 401], [77, 401], [77, 401], [77, 401]], [[500, 100], [560, 100], [620, 100], [560, 160]],
    [[100, 100], [200, 110], [190, 210], [95, 200]], [[400, 50], [470, 120], [400, 190], [330, 120]]], dtype=np.uint32)


def oracle_q4(oracle):
    S = 49
    to = np.array([0, 0, S, 0, S, S, 0, S], np.float32)
    ok = np.array([oracle.from_control_points(q.reshape(8).astype(np.float32), to)[0] for q in Q4_QUADS], dtype=np.uint8)
    one = np.zeros((1, 1), np.uint8)                        # GrayImage::new(1, 1)
    level = oracle.otsu_level(one)
    binary = ((one > level) * 255).astype(np.uint8)         # threshold(.., Binary)
    resized = np.zeros((4, 10, 10), np.uint8)
    for k, ms in enumerate((6, 7, 8, 10)):
        resized[k, :ms, :ms] = oracle.resize_triangle(binary, ms, ms)
    return {"q4_quads": Q4_QUADS, "q4_projection_ok": ok, "q4_standin_otsu": np.array([level], np.uint8), "q4_standin_binary": binary.reshape(1),
            "q4_standin_resized": resized}


def test_oracle_q4_branch_is_what_the_tests_assume(oracle):
    """the five degenerate quads fail, the two ordinary ones solve; one black pixel stays black through Otsu / threshold / resize"""
    got = oracle_q4(oracle)
    assert got["q4_projection_ok"].tolist() == [0, 0, 0, 0, 0, 1, 1]
    assert int(got["q4_standin_otsu"][0]) == 0 and int(got["q4_standin_binary"][0]) == 0 and not got["q4_standin_resized"].any()


def test_oracle_against_reference_q4_fixture(oracle):
    path = FIX / "q4_degenerate.a3fx"
    if not path.exists():
        pytest.skip(f"{path.name} absent: produce it with integration/dump_fixtures.rs where cargo exists (INTEGRATION.md)")
    ref, got = read_a3fx(path), oracle_q4(oracle)
    for k, v in got.items():
        assert k in ref and np.array_equal(np.asarray(ref[k]).reshape(-1), np.asarray(v).reshape(-1)), k
