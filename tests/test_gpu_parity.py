"""Parity of the HIP path against the CPU oracle, stage by stage, through the C ABI.  GPU only.

Bit-exact is the bar for everything integer (grey, threshold, candidates, corner order, codes, ids) and also
for the warped patches (the kernels repeat the oracle's float operations one by one with contraction off);
pose floats are compared at 1e-4 as BASELINE.json states."""
import numpy as np
import pytest

from tests.util import assert_frame_parity, markers_of_hip, markers_of_oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from aruco3_amd import _lib

    _lib.load()
    return _lib


def _detector(dicts, name="ARUCO", **cfg):
    from aruco3_amd.aruco import Detector, DetectorConfig

    return Detector(DetectorConfig(**cfg), dicts.new_from_named_dict(name))


def _run(det, frames, populate=True, out_cap=0):
    ctx = det._context()
    ctx.set_debug_taps(populate)
    a = np.ascontiguousarray(frames)
    if a.ndim == 3:
        a = a[None] if a.shape[-1] in (3, 4) else a[..., None]
    if a.ndim == 3:
        a = a[None]
    n, h, w, c = a.shape
    from aruco3_amd import _lib

    fmt = {1: _lib.FMT_L8, 3: _lib.FMT_RGB8, 4: _lib.FMT_RGBA8}[c]
    markers, per = ctx.detect_batch(a.ctypes.data, _lib.MEM_HOST, fmt, w, h, w * c, h * w * c, n, out_cap)
    return ctx, markers, per


def _check(det, oracle, frames, check_patches=True):
    d = det.dictionary
    # product path first (no debug taps: K1 writes no grey plane, the decode stage samples the caller's frames), then the
    # tapped run whose intermediate stages are compared below; both must return the same markers
    _, markers0, per0 = _run(det, frames, populate=False)
    ctx, markers, per = _run(det, frames)
    assert np.array_equal(per0, per) and np.array_equal(markers0, markers)
    frames = np.asarray(frames)
    if frames.ndim == 3 and frames.shape[-1] in (3, 4):
        frames = frames[None]
    elif frames.ndim == 2:
        frames = frames[None]
    pos = 0
    for f in range(frames.shape[0]):
        img = frames[f]
        res = oracle.detect(img, d.code_list, d.num_bits, det._context().tau)
        h, w = img.shape[:2]
        assert_frame_parity(ctx, f, img, res, w, h, check_patches)
        got = markers_of_hip(markers[pos: pos + int(per[f])])
        pos += int(per[f])
        assert got == markers_of_oracle(res)
    assert pos == len(markers)
    return ctx


def test_ieee_ops_are_correctly_rounded(hip, dicts):
    """f64 sqrt/div (Douglas-Peucker distance, 8x8 solve, Otsu) and f32 sqrt/div must round like the host's."""
    ctx = _detector(dicts)._context()
    rng = np.random.default_rng(5)
    a = np.concatenate([rng.integers(0, 1 << 27, 200000).astype(np.float64), rng.random(100000) * 1e6, [0.0, 1.0, 2.0, 3.0, 1e-300, 1e300]])
    b = np.concatenate([np.sqrt(rng.integers(1, 1 << 26, 200000).astype(np.float64)), rng.random(100000) * 1e3 + 1e-3, [1.0, 3.0, 7.0, 10.0, 1e300, 1e-300]])
    sq, dv, sqf, dvf = ctx.selftest_ieee(a, b)
    assert np.array_equal(sq, np.sqrt(a))
    assert np.array_equal(dv, a / b)
    af, bf = a.astype(np.float32), b.astype(np.float32)
    with np.errstate(over="ignore", under="ignore"):
        assert np.array_equal(sqf, np.sqrt(af))
        assert np.array_equal(dvf.view(np.uint32), (af / bf).view(np.uint32))


def test_find_nearest_and_tau(hip, dicts, oracle):
    """src/dictionaries.rs:239-281 through the device kernels"""
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    assert d.find_nearest(0x1084210) == (0, 0)
    assert d.find_nearest(0x1084209) == (2, 0)
    assert d.find_nearest(0b00000001_00001000_01000010_10001001) == (2, 1)
    assert d.find_nearest(0x1084217) == (1, 0)
    assert d.try_find_nearest(0b01100001_00001000_01000010_00001001)[0] == 2
    assert d.try_find_nearest(int("11111111" "0000100" "01000010" "00001001", 2)) is None
    rng = np.random.default_rng(3)
    q = rng.integers(0, 1 << 25, 500, dtype=np.uint64)
    idx, dist = hip.find_nearest(d.code_list, q)
    for i in range(q.size):
        assert (int(idx[i]), int(dist[i])) == oracle.find_nearest(d.code_list, int(q[i]))
    for name in ("ARTAG", "ARTOOLKITPLUS", "ARUCO", "APRILTAG_16H5"):
        dd = dicts.new_from_named_dict(name)
        assert hip.calculate_tau(dd.code_list) == oracle.calculate_tau(dd.code_list), name
    assert dicts.new_from_named_dict("ARTOOLKITPLUS").tau == oracle.calculate_tau(dicts.new_from_named_dict("ARTOOLKITPLUS").code_list)


@pytest.mark.parametrize("shape", [(480, 640), (479, 641), (64, 64), (17, 300), (300, 17), (5, 3), (1, 1), (2, 9), (250, 244)])
@pytest.mark.parametrize("channels", [3, 4, 1])
def test_threshold_stage_bit_exact(dicts, oracle, shape, channels):
    """K1 against into_luma8 + adaptive_threshold on noise and on smooth ramps (clipped windows at every border)."""
    rng = np.random.default_rng(shape[0] * 1000 + shape[1] + channels)
    h, w = shape
    noise = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    ramp = ((xx * 3 + yy * 5) % 256).astype(np.uint8)[..., None].repeat(channels, axis=2)
    flat = np.full((h, w, channels), 200, np.uint8)
    det = _detector(dicts)
    frames = np.stack([noise, ramp, flat])
    batch = frames if channels > 1 else frames[..., 0][..., None]
    ctx, _, _ = _run(det, batch, populate=True)
    want = []
    for f in range(3):
        img = frames[f] if channels > 1 else frames[f][..., 0]
        grey = oracle.to_luma8(img)
        want.append(oracle.adaptive_threshold(grey, 7))
        assert np.array_equal(ctx.download_grey(f, w, h), grey)
        assert np.array_equal(ctx.download_grey(f, w, h, thresholded=True), want[f])
    # without debug taps K1 writes no grey plane; the thresholded image must not change, and asking for grey is an error
    ctx, _, _ = _run(det, batch, populate=False)
    for f in range(3):
        assert np.array_equal(ctx.download_grey(f, w, h, thresholded=True), want[f])
    from aruco3_amd._lib import A3Error
    with pytest.raises(A3Error):
        ctx.download_grey(0, w, h)


def test_other_threshold_window(dicts, oracle):
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 1)
    for win in (3, 11):
        det = _detector(dicts, threshold_window=win)
        ctx, _, _ = _run(det, frames, populate=False)
        grey = oracle.to_luma8(frames[0])
        assert np.array_equal(ctx.download_grey(0, 640, 480, thresholded=True), oracle.adaptive_threshold(grey, win))


@pytest.mark.parametrize("radius", list(range(8, 32)))
def test_threshold_windows_8_to_31(dicts, oracle, radius):
    """The fused kernel of windows 8..31 (grey ring in registers + LDS, 32-bit sums: k_threshold_big.hip; from 16 on with two apron
    lanes per side and 960 output columns per wave) with its vector loads
    (W % 16 == 0: one and two column strips, images lower and narrower than the window) and with its per-pixel loads (any other
    width), every pixel format, noise / ramps / flat: into_luma8 + adaptive_threshold(&grey, radius) bit for bit, with and
    without a grey plane."""
    for (h, w), channels in (((480, 640), 3), ((250, 1040), 3), ((37, 2000), 4), ((300, 16), 1), ((9, 48), 3), ((1, 16), 4),
                             ((70, 1008), 1), ((129, 641), 3), ((40, 30), 4)):
        rng = np.random.default_rng(radius * 7919 + h * 31 + w + channels)
        noise = rng.integers(0, 256, (h, w, channels), dtype=np.uint8)
        yy, xx = np.mgrid[0:h, 0:w]
        ramp = ((xx * 3 + yy * 5) % 256).astype(np.uint8)[..., None].repeat(channels, axis=2)
        flat = np.full((h, w, channels), 200, np.uint8)
        frames = np.stack([noise, ramp, flat, noise[::-1].copy()])
        batch = frames if channels > 1 else frames[..., 0][..., None]
        det = _detector(dicts, threshold_window=radius)
        want = []
        ctx, _, _ = _run(det, batch, populate=True)
        for f in range(len(frames)):
            grey = oracle.to_luma8(frames[f] if channels > 1 else frames[f][..., 0])
            want.append(oracle.adaptive_threshold(grey, radius))
            assert np.array_equal(ctx.download_grey(f, w, h), grey), (radius, h, w, channels, f)
            assert np.array_equal(ctx.download_grey(f, w, h, thresholded=True), want[f]), (radius, h, w, channels, f)
        ctx, _, _ = _run(det, batch, populate=False)
        for f in range(len(frames)):
            assert np.array_equal(ctx.download_grey(f, w, h, thresholded=True), want[f]), (radius, h, w, channels, f, "no taps")


@pytest.mark.parametrize("config", [1, 2, 4])
def test_baseline_configs_full_parity(dicts, oracle, config):
    """BASELINE.json configs 1, 2 (=3 per GPU) and 4 at full resolution, a few frames each: every stage equal."""
    from aruco3_amd import synth

    spec, name = synth.config_spec(config)
    frames, truth = synth.config_frames(config, 3)
    det = _detector(dicts, name)
    _check(det, oracle, frames)
    if config in (1, 2):
        ctx, markers, per = _run(det, frames, populate=False)
        pos = 0
        for f in range(len(frames)):
            ids = sorted(int(m["id"]) for m in markers[pos: pos + int(per[f])])
            pos += int(per[f])
            assert ids == sorted(t.id for t in truth[f])  # what was rendered is what is read


def test_config5_4k_with_pose(dicts, oracle):
    """BASELINE.json config 5: 3840x2160, 16 markers, detect + IPPE pose; pose floats within 1e-4 (relative to scale)."""
    from aruco3_amd import pose, synth

    frames, truth = synth.config_frames(5, 1)
    det = _detector(dicts, "ARUCO")
    ctx = _check(det, oracle, frames, check_patches=True)
    _, markers, per = _run(det, frames, populate=False)
    assert len(markers) >= 12
    corners = markers["corners"]
    got = pose.solve_batch(corners, 40.0, (3840, 2160))
    for i, m in enumerate(markers):
        (e1, r1, t1), (e2, r2, t2) = oracle.solve_with_undistorted_points(corners[i], 40.0, (3840, 2160))
        for (ge, gr, gt), (e, r, t) in zip([(got[i][0].error, got[i][0].rotation, got[i][0].translation),
                                            (got[i][1].error, got[i][1].rotation, got[i][1].translation)], [(e1, r1, t1), (e2, r2, t2)]):
            assert np.allclose(gr, r, atol=1e-4, rtol=0)
            assert np.allclose(gt, t, atol=1e-4 * max(1.0, float(np.abs(t).max())), rtol=0)
            assert abs(ge - e) <= 1e-4


def test_noise_frames_reference_bench_recipe(dicts, oracle):
    """benches/detect_markers.rs:29-45: uniform random RGB; ~half the pixels are foreground, so this is the
    contour stage's stress case (giant component, hundreds of thousands of tiny borders, column-0 anomalies)."""
    from aruco3_amd import synth

    det = _detector(dicts)
    frames = np.stack([synth.noise_frame(512, 512, 77 + i) for i in range(2)])
    ctx = _check(det, oracle, frames)
    st = ctx.stats()
    assert st["contours_traced"] > 10000


def test_structured_binary_layouts(dicts, oracle):
    """Hand-made layouts that exercise the border follower's corner cases: shapes touching every image edge,
    one-pixel-wide strokes (points visited twice), nested rings, diagonal chains."""
    h, w = 96, 128
    base = np.full((h, w), 210, np.uint8)
    imgs = []
    a = base.copy(); a[:, :20] = 20; a[10:40, 0:60] = 20; a[50:90, 100:128] = 20; imgs.append(a)       # left/right edge blocks
    b = base.copy(); b[0:8, :] = 20; b[h - 8:, :] = 20; b[20:70, 30:100] = 20; b[35:55, 45:85] = 210; imgs.append(b)  # top/bottom + ring
    c = base.copy()
    for i in range(60):
        c[10 + i, 10 + i] = 20; c[10 + i, 100 - i] = 20
    c[80, 5:120] = 20; imgs.append(c)                                                                  # thin strokes
    d = np.full((h, w), 30, np.uint8); d[5:90, 5:120] = 220; d[20:70, 20:100] = 30; d[30:60, 30:90] = 220; imgs.append(d)  # dark frame, nested
    det = _detector(dicts)
    _check(det, oracle, np.stack(imgs)[..., None])


def test_chunked_batches_equal_single(dicts, oracle):
    """A pool too small for the batch splits it into chunks; results must not change."""
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 6)
    det = _detector(dicts, "ARUCO_DEFAULT")
    _, m_all, per_all = _run(det, frames, populate=False)
    det2 = _detector(dicts, "ARUCO_DEFAULT")
    det2._context().set_pool_limits(max_darts=12000, max_points=0)
    ctx2, m_chunk, per_chunk = _run(det2, frames, populate=False)
    assert ctx2.stats()["chunks"] > 1
    assert per_all.tolist() == per_chunk.tolist()
    assert markers_of_hip(m_all) == markers_of_hip(m_chunk)
    assert m_all["frame"].tolist() == m_chunk["frame"].tolist()


def test_error_behaviour(hip, dicts):
    from aruco3_amd.aruco import Detector, DetectorConfig

    d = dicts.new_from_named_dict("ARUCO")
    with pytest.raises(hip.A3Error):  # imageproc asserts block_radius > 0
        Detector(DetectorConfig(threshold_window=0), d)._context()
    det = Detector(DetectorConfig(), d)
    img = np.zeros((480, 640, 3), np.uint8)
    assert det.detect(img).markers == []
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 1)
    with pytest.raises(hip.A3Error) as e:
        det._context().detect_batch(frames.ctypes.data, hip.MEM_HOST, hip.FMT_RGB8, 640, 480, 640 * 3, 640 * 480 * 3, 1, out_cap=1)
    assert e.value.code == hip.ERR_CAPACITY


def test_pose_kats_on_device(dicts):
    """src/pose.rs:514-598 through the pose kernel"""
    from aruco3_amd import pose
    from tests.test_oracle_kat import PA_ROT, PB_ROT

    pa, pb = pose.solve_with_undistorted_points([(90, 89), (95, 150), (80, 170), (75, 90)], 17.0, (1000, 1000))
    assert np.abs(pa.rotation - PA_ROT).sum() < 2e-5 and np.abs(pb.rotation - PB_ROT).sum() < 2e-5
    assert np.abs(pa.translation - np.array([20.32196265994096, 29.69316666108512, 238.3658341694123])).sum() < 0.0005
    assert np.abs(pb.translation - np.array([19.85146615649354, 29.20013946746331, 234.3277337340188])).sum() < 0.0005
    pts = [(-0.090, -0.089), (-0.095, -0.150), (-0.080, -0.170), (-0.075, -0.090)]
    qa, qb = pose.solve_with_normalized_points(pts, 19.0)
    sign = np.array([[-1, -1, -1], [-1, -1, -1], [1, 1, 1]])
    assert np.abs(qa.rotation - PA_ROT * sign).max() <= 1e-5 and np.abs(qb.rotation - PB_ROT * sign).max() <= 1e-5
    assert np.abs(qa.translation - np.array([-22.712781796404, -33.18648038591866, 266.408873483460])).max() <= 1e-3


def test_device_resident_torch_input(dicts, oracle):
    """Frames already in HBM (torch tensor) through Detector.detect_batch, as bench.py feeds them."""
    import torch

    from aruco3_amd import synth
    from aruco3_amd.aruco import Detector, DetectorConfig

    frames, truth = synth.config_frames(1, 2)
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    det = Detector(DetectorConfig(), d)
    out = det.detect_batch(torch.from_numpy(frames).cuda())
    for f in range(2):
        ref = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        assert [(m.id, m.code, m.corners, m.hamming_distance) for m in out[f].markers] == [
            (m["id"], m["code"], m["corners"], m["hamming_distance"]) for m in ref["markers"]]


@pytest.mark.parametrize("name", ["ARUCO_MIP_16H3", "APRILTAG_25H9", "ARUCO_MIP_36H12", "APRILTAG_36H9", "CHILITAGS", "ARTAG"])
def test_other_dictionaries(dicts, oracle, name):
    """16-, 25-, 36- and 64-bit codes (6x6 .. 10x10 cells), a 5329-entry table, and a table whose tau is computed (ARTAG)."""
    from aruco3_amd import synth

    d = dicts.new_from_named_dict(name)
    spec = synth.SynthSpec(800, 600, n_markers=(3, 3), side=(150.0, 190.0), min_center_sep=230.0, perspective=0.08)
    frames = np.stack([synth.render_frame(spec, d.code_list, d.num_bits, 4242 + i)[0] for i in range(2)])
    det = _detector(dicts, name)
    ctx = _check(det, oracle, frames)
    assert ctx.tau == oracle.calculate_tau(d.code_list) if d._tau == 0 else ctx.tau == d._tau


@pytest.mark.parametrize("cfg", [
    dict(filter_high_bit_errors=False),
    dict(homography_sample_size=32),
    dict(homography_sample_size=7),       # == mark size: image::imageops::resize copies instead of filtering
    dict(homography_sample_size=80),
    dict(contour_simplification_epsilon=0.02),
    dict(contour_simplification_epsilon=0.11),
    dict(min_side_length_factor=0.02, min_corner_separation_factor=0.01),
    dict(min_side_length_factor=0.9),
])
def test_detector_config_variants(dicts, oracle, cfg):
    """Every DetectorConfig knob (src/aruco.rs:23-30) away from its default, on clean and on noisy frames."""
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 2)
    noisy = synth.render_frame(synth.SynthSpec(640, 480, n_markers=(2, 2), side=(100.0, 140.0), min_center_sep=200.0, noise_sigma=6.0),
                               dicts.new_from_named_dict("ARUCO_DEFAULT").code_list, 25, 99)[0]
    det = _detector(dicts, "ARUCO_DEFAULT", **cfg)
    ocfg = oracle.Config.default()
    for k, v in cfg.items():
        setattr(ocfg, k, int(v) if isinstance(v, bool) else v)
    d = det.dictionary
    all_frames = np.concatenate([frames, noisy[None]])
    ctx, markers, per = _run(det, all_frames)
    pos = 0
    for f in range(len(all_frames)):
        res = oracle.detect(all_frames[f], d.code_list, d.num_bits, d._tau, config=ocfg)
        assert_frame_parity(ctx, f, all_frames[f], res, 640, 480)
        assert markers_of_hip(markers[pos: pos + int(per[f])]) == markers_of_oracle(res)
        pos += int(per[f])


@pytest.mark.parametrize("window,row_pad,off", [(7, 20, 3), (11, 32, 16), (11, 20, 3)])
def test_strided_and_offset_input(hip, dicts, oracle, window, row_pad, off):
    """row_stride / frame_stride larger than the packed size and a base pointer that is not 16-byte aligned
    (the kernel then takes its per-pixel path); results must not change.  Window 11: the fused kernel of windows 8..31 on padded
    but aligned rows (vector loads) and on unaligned ones (per-pixel loads)."""
    import torch

    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 2)
    n, h, w, c = frames.shape
    row_stride, frame_stride = w * c + row_pad, (w * c + row_pad) * h + 64
    buf = np.zeros(off + n * frame_stride, dtype=np.uint8)
    for f in range(n):
        for y in range(h):
            s = off + f * frame_stride + y * row_stride
            buf[s: s + w * c] = frames[f, y].reshape(-1)
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    det = _detector(dicts, "ARUCO_DEFAULT", threshold_window=window)
    ocfg = oracle.Config.default()
    ocfg.threshold_window = window
    ctx = det._context()
    ctx.set_debug_taps(True)
    for mem, ptr, keep in ((hip.MEM_HOST, buf.ctypes.data + off, buf), (hip.MEM_DEVICE, None, None)):
        if mem == hip.MEM_DEVICE:
            keep = torch.from_numpy(buf).cuda()
            ptr = keep.data_ptr() + off
        ctx.set_debug_taps(False)   # product path: the decode stage samples the strided frames themselves
        markers0, per0 = ctx.detect_batch(ptr, mem, hip.FMT_RGB8, w, h, row_stride, frame_stride, n)
        ctx.set_debug_taps(True)
        markers, per = ctx.detect_batch(ptr, mem, hip.FMT_RGB8, w, h, row_stride, frame_stride, n)
        assert np.array_equal(per0, per) and np.array_equal(markers0, markers)
        pos = 0
        for f in range(n):
            res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau, config=ocfg)
            assert_frame_parity(ctx, f, frames[f], res, w, h)
            assert markers_of_hip(markers[pos: pos + int(per[f])]) == markers_of_oracle(res)
            pos += int(per[f])


def test_full_size_batch_sample_and_properties(dicts, oracle):
    """BASELINE config 2 at full size: 32 frames of 1920x1080 in one batch against the oracle, plus two properties that hold
    at any size: the result of a frame does not depend on its position in the batch, and detection is idempotent."""
    from aruco3_amd import synth

    frames, truth = synth.config_frames(2, 32)
    det = _detector(dicts, "ARUCO")
    d = det.dictionary
    ctx, markers, per = _run(det, frames, populate=False)
    pos = 0
    exact_truth = 0
    for f in range(len(frames)):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau, keep_debug=False)
        got = markers_of_hip(markers[pos: pos + int(per[f])])
        assert got == markers_of_oracle(res), f
        exact_truth += sorted(m[0] for m in got) == sorted(t.id for t in truth[f])
        pos += int(per[f])
    assert exact_truth >= 28   # the reference algorithm itself drops a marker now and then
    perm = np.random.default_rng(1).permutation(len(frames))
    _, m2, per2 = _run(det, frames[perm], populate=False)
    by_frame, pos = {}, 0
    for i, f in enumerate(perm):
        by_frame[int(f)] = markers_of_hip(m2[pos: pos + int(per2[i])]); pos += int(per2[i])
    pos = 0
    for f in range(len(frames)):
        assert by_frame[f] == markers_of_hip(markers[pos: pos + int(per[f])]); pos += int(per[f])
    _, m3, per3 = _run(det, frames, populate=False)
    assert per3.tolist() == per.tolist() and markers_of_hip(m3) == markers_of_hip(markers)


def test_noise_1080p_reference_bench_input(dicts, oracle):
    """benches/detect_markers.rs at its largest size: one 1920x1080 uniform-noise frame (about 4 M contour-graph nodes,
    ~5e5 borders, a giant component) must come out identical, stage by stage."""
    from aruco3_amd import synth

    det = _detector(dicts)
    frame = synth.noise_frame(1920, 1080, 2026)
    ctx = _check(det, oracle, frame[None])
    st = ctx.stats()
    assert st["darts"] > 1_000_000 and st["contours_traced"] > 100_000


def test_bgra_webcam_byte_order(hip, dicts, oracle):
    """frames in B,G,R,A order (what examples/webcam_kamera.rs:38-52 swizzles on the CPU) give the
    result of the same frames in R,G,B,A order; also on a width that forces the per-pixel load path."""
    from aruco3_amd import synth

    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    for (w, h) in ((640, 480), (333, 251)):
        spec = synth.SynthSpec(w, h, n_markers=(2, 2), side=(90.0, 110.0), min_center_sep=130.0)
        rgb = synth.render_frame(spec, d.code_list, d.num_bits, 77)[0]
        rgba = np.concatenate([rgb, np.full((h, w, 1), 255, np.uint8)], axis=2)
        bgra = np.ascontiguousarray(rgba[..., [2, 1, 0, 3]])
        det = _detector(dicts, "ARUCO_DEFAULT")
        ctx = det._context()
        ctx.set_debug_taps(False)
        markers0, per0 = ctx.detect_batch(bgra.ctypes.data, hip.MEM_HOST, hip.FMT_BGRA8, w, h, w * 4, w * h * 4, 1)
        ctx.set_debug_taps(True)
        markers, per = ctx.detect_batch(bgra.ctypes.data, hip.MEM_HOST, hip.FMT_BGRA8, w, h, w * 4, w * h * 4, 1)
        assert np.array_equal(per0, per) and np.array_equal(markers0, markers)
        res = oracle.detect(rgba, d.code_list, d.num_bits, d._tau)
        assert_frame_parity(ctx, 0, rgba, res, w, h)
        assert markers_of_hip(markers) == markers_of_oracle(res)
        out = det.detect_batch(bgra, bgra=True)[0]
        assert [m.id for m in out.markers] == [m["id"] for m in res["markers"]]


def test_make_binary_image_cell_order_q6(dicts, oracle):
    """a marker drawn in ARDictionary::make_binary_image's LSB-first cell order (src/dictionaries.rs:212-232,
    what examples/macroquad_detect.rs:27-43 renders) is the canonical marker turned by 180 degrees; it is still read, with
    rotation index 2, and the corner list starts at the opposite corner."""
    from aruco3_amd import synth

    d = dicts.new_from_named_dict("ARUCO")
    frames = {}
    for order in ("msb", "lsb"):
        spec = synth.SynthSpec(640, 480, n_markers=(2, 2), side=(120.0, 120.0), rotation_deg=(0.0, 0.0), perspective=0.0, min_center_sep=200.0,
                               cell_order=order)
        frames[order] = synth.render_frame(spec, d.code_list, d.num_bits, 5150)
    det = _detector(dicts, "ARUCO")
    _check(det, oracle, np.stack([frames["msb"][0], frames["lsb"][0]]))
    _, markers, per = _run(det, np.stack([frames["msb"][0], frames["lsb"][0]]), populate=False)
    a, b = markers[: int(per[0])], markers[int(per[0]):]
    # (an axis-aligned marker yields two surviving candidates, its outer and its inner border, whose corner lists start at
    #  different corners -- discard_too_near compares same-index corners only, src/aruco.rs:189-190)
    assert set(a["id"].tolist()) == set(t.id for t in frames["msb"][1])
    assert set(b["id"].tolist()) == set(t.id for t in frames["lsb"][1])
    # identical geometry, cells mirrored through the centre: every rotation index moves by 2
    for i in set(a["id"].tolist()):
        ra = sorted((int(m["rotation"]) + 2) % 4 for m in a if m["id"] == i)
        rb = sorted(int(m["rotation"]) for m in b if m["id"] == i)
        assert ra == rb


def test_detect_with_pose_in_one_call(hip, dicts, oracle):
    """SURVEY section 8f item 2 / BASELINE config 5: a3_detect_batch_pose returns the two IPPE poses of every marker with the
    detections; they equal a separate solve on the returned corners, with and without intrinsics."""
    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 3)
    det = _detector(dicts, "ARUCO_DEFAULT")
    ctx = det._context()
    n, h, w, c = frames.shape
    for intr in (None, hip.Intrinsics(w, h, 600.0, 610.0, w / 2.0, h / 2.0)):
        markers, per, poses = ctx.detect_batch_pose(frames.ctypes.data, hip.MEM_HOST, hip.FMT_RGB8, w, h, w * c, h * w * c, n, 25.0, intr)
        assert len(markers) == int(per.sum()) == len(poses) and len(markers) >= 8
        for i, m in enumerate(markers):
            if intr is None:
                ref = oracle.solve_with_undistorted_points(m["corners"], 25.0, (w, h))
            else:
                ref = oracle.solve_with_intrinsics(m["corners"], 25.0, 600.0, 610.0, w / 2.0, h / 2.0)
            for j in range(2):
                e, r, t = ref[j]
                assert abs(poses[i, j, 0] - e) <= 1e-4
                assert np.allclose(poses[i, j, 1:10].reshape(3, 3), r, atol=1e-4, rtol=0)
                assert np.allclose(poses[i, j, 10:13], t, atol=1e-4 * max(1.0, float(np.abs(t).max())), rtol=0)


def test_detector_detect_batch_with_pose(dicts, oracle):
    from aruco3_amd import synth
    from aruco3_amd.pinhole import CameraIntrinsics

    frames, _ = synth.config_frames(1, 2)
    det = _detector(dicts, "ARUCO_DEFAULT")
    h, w = frames.shape[1:3]
    plain = det.detect_batch(frames)
    fused = det.detect_batch_with_pose(frames, 30.0, CameraIntrinsics.new(w, h, 700.0, 700.0))
    assert len(fused) == len(plain)
    for (d, poses), p in zip(fused, plain):
        assert [(m.id, m.corners) for m in d.markers] == [(m.id, m.corners) for m in p.markers]
        assert len(poses) == len(d.markers)
        for a, b in poses:
            assert a.error <= b.error and a.rotation.shape == (3, 3)


def test_start_resolution_fast_and_full_paths(dicts, oracle):
    """k_resolve_fast confirms the natural border starts on ordinary frames (no pass over all darts); a component whose
    first pixel lies in column 0 forces the fixpoint passes.  Both give the oracle's candidates."""
    from aruco3_amd import synth

    det = _detector(dicts, "ARUCO_DEFAULT")
    frames, _ = synth.config_frames(1, 1)
    _check(det, oracle, frames)
    assert det._context().stats()["resolve_iterations"] == 0
    # bright background = one foreground component whose first pixel is (0, 0); two dark pixels touching column 0
    # diagonally make its outer border's natural start event not fire (found by searching tests/dart_model.py)
    a = np.full((96, 128), 210, np.uint8)
    a[3, 1] = 20; a[4, 0] = 20
    a[30:60, 40:90] = 20
    _check(det, oracle, a[None, ..., None])
    assert det._context().stats()["resolve_iterations"] >= 1
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, size=(2, 200, 320, 3), dtype=np.uint8)
    _check(det, oracle, noise)


def test_device_side_plan_and_its_fallback(dicts, oracle):
    """A batch shaped like the previous one is planned on the device from the previous dart total (no read-back in the
    middle of the pipeline).  Same-shape batches with a graph that is suddenly ~50x larger must fall back to the host
    plan and still match the oracle; so must the batches after it."""
    from aruco3_amd import synth

    det = _detector(dicts, "ARUCO_DEFAULT")
    clean, _ = synth.config_frames(1, 2)                      # 640x480, few thousand darts
    rng = np.random.default_rng(11)
    noise = rng.integers(0, 256, size=clean.shape, dtype=np.uint8)
    darts = []
    for frames in (clean, clean, noise, noise, clean, clean):
        _check(det, oracle, frames)
        darts.append(det._context().stats()["darts"])
    assert darts[0] == darts[1] == darts[4] == darts[5] and darts[2] == darts[3] and darts[2] > 20 * darts[0]


def test_submit_collect_equals_detect_batch(hip, dicts, oracle):
    """a3_detect_batch_submit / _collect (two contexts on one stream, the next batch submitted before the previous one is
    collected) return exactly what a3_detect_batch returns, including when the device asks for a synchronous re-run
    (first use of a shape; a graph that outgrows the device-side plan)."""
    import torch

    from aruco3_amd import synth

    frames, _ = synth.config_frames(1, 3)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, size=frames.shape, dtype=np.uint8)
    n, h, w, c = frames.shape
    ref = _detector(dicts, "ARUCO_DEFAULT")._context()
    ref.set_debug_taps(False)
    want = {}
    for name, fr in (("clean", frames), ("noise", noise)):
        want[name] = ref.detect_batch(fr.ctypes.data, hip.MEM_HOST, hip.FMT_RGB8, w, h, w * c, h * w * c, n)
    stream = torch.cuda.Stream()
    dev = {"clean": torch.from_numpy(frames).cuda(), "noise": torch.from_numpy(noise).cuda()}
    torch.cuda.synchronize()
    ctxs = [_detector(dicts, "ARUCO_DEFAULT")._context() for _ in range(2)]
    for cx in ctxs:
        cx.set_stream(stream.cuda_stream)
        cx.set_debug_taps(False)
    order = ["clean", "clean", "clean", "noise", "noise", "clean", "clean", "noise", "clean"]
    args = lambda name: (dev[name].data_ptr(), hip.MEM_DEVICE, hip.FMT_RGB8, w, h, w * c, h * w * c, n)
    ctxs[0].submit(*args(order[0]))
    for i, name in enumerate(order):
        if i + 1 < len(order):
            ctxs[(i + 1) % 2].submit(*args(order[i + 1]))
        markers, per = ctxs[i % 2].collect()
        assert np.array_equal(per, want[name][1]) and np.array_equal(markers, want[name][0]), (i, name)
    with pytest.raises(hip.A3Error):
        ctxs[0].collect()            # nothing in flight
    ctxs[0].submit(*args("clean"))
    with pytest.raises(hip.A3Error):
        ctxs[0].submit(*args("clean"))   # one batch per context
    markers, per = ctxs[0].collect()
    assert np.array_equal(markers, want["clean"][0])


def _fuzz_frame(rng, h, w, kind):
    """Structured random content: rectangles / rotated quads / strokes / speckle on a gradient, optional noise."""
    yy, xx = np.mgrid[0:h, 0:w]
    base = (120 + 60 * np.sin(xx / max(w, 1) * rng.uniform(1, 9)) + 40 * np.cos(yy / max(h, 1) * rng.uniform(1, 9))).astype(np.float32)
    img = base.copy()
    for _ in range(int(rng.integers(1, 9))):
        if kind == "quads" and min(h, w) > 24:
            cx, cy = rng.uniform(0, w), rng.uniform(0, h)
            a, s = rng.uniform(0, np.pi), rng.uniform(4, min(h, w) / 2)
            u = (xx - cx) * np.cos(a) + (yy - cy) * np.sin(a)
            v = -(xx - cx) * np.sin(a) + (yy - cy) * np.cos(a)
            m = (np.abs(u) < s) & (np.abs(v) < s * rng.uniform(0.3, 1.0))
            img[m] = rng.choice([15.0, 240.0])
            if rng.random() < 0.6:
                img[(np.abs(u) < s * 0.6) & (np.abs(v) < s * 0.3)] = rng.choice([15.0, 240.0])
        elif kind == "strokes":
            x0, y0, x1, y1 = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0, w), rng.uniform(0, h)
            t = np.linspace(0, 1, 4 * max(h, w))
            xs, ys = np.clip((x0 + (x1 - x0) * t).astype(int), 0, w - 1), np.clip((y0 + (y1 - y0) * t).astype(int), 0, h - 1)
            img[ys, xs] = 20.0
        else:
            x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
            x1, y1 = min(w, x0 + int(rng.integers(1, max(2, w // 2)))), min(h, y0 + int(rng.integers(1, max(2, h // 2))))
            img[y0:y1, x0:x1] = rng.choice([10.0, 245.0])
    if rng.random() < 0.5:
        img += rng.normal(0, rng.uniform(1, 25), img.shape)
    if rng.random() < 0.3:
        sp = rng.random(img.shape) < 0.02
        img[sp] = rng.choice([0.0, 255.0])
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("seed", range(6))
def test_randomised_structured_frames_full_parity(dicts, oracle, seed):
    """Fuzz: odd sizes (not multiples of 16 / 64, narrower than a lane, taller than wide), 1 / 3 / 4 channels, shapes cut by
    every image edge, thin strokes, speckle and noise -- every stage against the oracle, product path and tapped path."""
    rng = np.random.default_rng(1000 + seed)
    det = _detector(dicts, ["ARUCO", "ARUCO_MIP_36H12", "APRILTAG_16H5"][seed % 3])
    for case in range(5):
        h = int(rng.choice([1, 2, 7, 16, 33, 64, 65, 127, 200, 311]))
        w = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 130, 257, 400]))
        c = int(rng.choice([1, 3, 4]))
        n = int(rng.integers(1, 5))
        kind = ["rects", "quads", "strokes"][int(rng.integers(0, 3))]
        frames = np.stack([np.repeat(_fuzz_frame(rng, h, w, kind)[..., None], c, axis=2) for _ in range(n)])
        if c >= 3:   # decorrelate the channels a little so that the luma weights matter
            frames[..., 1] = np.clip(frames[..., 1].astype(np.int16) + rng.integers(-20, 21), 0, 255).astype(np.uint8)
        _check(det, oracle, frames if c > 1 else frames[..., 0][..., None])


@pytest.mark.parametrize("shape", [(96, 16448), (16448, 96)])
def test_frames_wider_or_taller_than_16384(dicts, oracle, shape):
    """Coordinates beyond 2^14: k_contour_quads then takes its 64-bit distance numerators (the 24-bit multiply-adds are only exact
    below), k_dart_count its several-waves-per-row shape; 257 packed words per row.  Every stage against the oracle."""
    h, w = shape
    rng = np.random.default_rng(16448)
    det = _detector(dicts, "ARUCO")
    frames = np.stack([np.repeat(_fuzz_frame(rng, h, w, kind)[..., None], 3, axis=2) for kind in ("quads", "rects")])
    # a few big quads that reach into the far end, so that candidates with coordinates > 16384 exist
    for f in range(2):
        for k in range(3):
            if w > h:
                x0 = w - 110 - 150 * k; frames[f, 20:76, x0:x0 + 90] = 15 if (f + k) % 2 else 240
                frames[f, 30:66, x0 + 15:x0 + 75] = 240 if (f + k) % 2 else 15
            else:
                y0 = h - 110 - 150 * k; frames[f, y0:y0 + 90, 20:76] = 15 if (f + k) % 2 else 240
                frames[f, y0 + 15:y0 + 75, 30:66] = 240 if (f + k) % 2 else 15
    ctx = _check(det, oracle, frames, check_patches=False)
    cands = np.concatenate([np.asarray(ctx.candidates(f, before_discard=True)).reshape(-1, 8) for f in range(2)])
    assert cands.size and cands.max() > 16384      # the test is about those


def test_many_small_frames_in_one_batch(dicts, oracle):
    """Per-frame bookkeeping under load: 600 frames of 96x80 (several per wave everywhere), a sample checked in full,
    every frame's marker list against the oracle."""
    from aruco3_amd import synth

    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    det = _detector(dicts, "ARUCO_DEFAULT")
    spec = synth.SynthSpec(96, 80, n_markers=(0, 1), side=(40.0, 60.0), min_center_sep=60.0)
    frames = np.stack([synth.render_frame(spec, d.code_list, d.num_bits, 5000 + i)[0] for i in range(600)])
    ctx, markers, per = _run(det, frames, populate=False)
    pos, found = 0, 0
    for f in range(len(frames)):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        assert markers_of_hip(markers[pos: pos + int(per[f])]) == markers_of_oracle(res), f
        pos += int(per[f]); found += int(per[f])
    assert pos == len(markers) and found > 100


def test_device_side_frame_generator(dicts, oracle):
    """frames rendered on the GPU (a3_synth_render) from the host renderer's layouts.  They are
    (nearly) the host renderer's frames, decode to the ids that were drawn, and -- downloaded -- give the oracle exactly
    what the HIP path finds in them in place."""
    import torch

    from aruco3_amd import synth

    for config, count in ((1, 2), (2, 2), (4, 2)):
        spec, name = synth.config_spec(config)
        d = dicts.new_from_named_dict(name)
        seeds = [synth.frame_seed(config, i) for i in range(count)]
        dev, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds)
        frames = dev.cpu().numpy()
        host = np.stack([synth.render_frame(spec, d.code_list, d.num_bits, s)[0] for s in seeds])
        if spec.noise_sigma == 0.0:
            diff = np.abs(frames.astype(np.int16) - host.astype(np.int16))
            # f32 painting vs f64 painting: a sub-sample on a cell edge may fall on the other side (one ninth of black-white)
            assert diff.max() <= 30 and (diff > 1).mean() < 0.002 and (diff > 0).mean() < 0.02
        det = _detector(dicts, name)
        in_place = det.detect_batch(dev)                              # device-resident tensor, no copy
        for f in range(count):
            res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
            assert [(m.id, m.corners) for m in in_place[f].markers] == [(m["id"], [tuple(c) for c in m["corners"]]) for m in res["markers"]]
            drawn = sorted(t.id for t in truths[f])
            found = sorted(m.id for m in in_place[f].markers)
            assert len(set(found) & set(drawn)) >= len(set(drawn)) - 1     # (noisy frames may add a false positive: the oracle's too)
            if spec.noise_sigma == 0.0:
                assert set(found) <= set(drawn)


def test_entry_resolution_per_frame_and_global(dicts, oracle):
    """Cross-tile border pieces are joined per frame in LDS when a frame has few of them (clean frames) and by the global
    doubling rounds otherwise; a batch that mixes both kinds must take the global path, and going back and forth between
    the two must not change any result."""
    from aruco3_amd import synth

    det = _detector(dicts, "ARUCO")
    clean, _ = synth.config_frames(2, 2)                     # 1920x1080: borders cross many 2048-dart tiles
    rng = np.random.default_rng(21)
    noise = rng.integers(0, 256, size=clean.shape, dtype=np.uint8)
    mixed = np.stack([clean[0], noise[0]])
    for frames in (clean, mixed, clean, noise, clean):
        _check(det, oracle, frames, check_patches=False)


def test_baseline_config2_full_batch_against_oracle(dicts, oracle):
    """BASELINE config 2 at its full size -- 256 frames of 1920x1080, rendered on the device -- every frame's marker list
    (ids, codes, corner integers and order, rotation, Hamming distance) against the oracle, plus idempotence of the batch
    call and equality of the synchronous and the split entry points on the same device-resident frames."""
    from aruco3_amd import synth

    spec, name = synth.config_spec(2)
    d = dicts.new_from_named_dict(name)
    seeds = [synth.frame_seed(2, i) for i in range(256)]
    dev, truths = synth.render_frames_device(spec, d.code_list, d.num_bits, seeds)
    det = _detector(dicts, name)
    ctx = det._context()
    ctx.set_debug_taps(False)
    n, h, w, c = dev.shape
    args = (dev.data_ptr(), 1, 0, w, h, w * c, h * w * c, n)   # MEM_DEVICE, FMT_RGB8
    markers, per = ctx.detect_batch(*args)
    again, per2 = ctx.detect_batch(*args)
    assert np.array_equal(per, per2) and np.array_equal(markers, again)
    ctx.submit(*args)
    split, per3 = ctx.collect()
    assert np.array_equal(per, per3) and np.array_equal(markers, split)
    frames = dev.cpu().numpy()
    pos, all_found = 0, 0
    for f in range(n):
        res = oracle.detect(frames[f], d.code_list, d.num_bits, d._tau)
        got = markers_of_hip(markers[pos: pos + int(per[f])])
        pos += int(per[f])
        assert got == markers_of_oracle(res), f
        all_found += sorted(set(m[0] for m in got)) == sorted(set(t.id for t in truths[f]))
    assert pos == len(markers) and all_found >= 230      # the remaining frames lose a marker to quirks Q2/Q3, in the oracle too
