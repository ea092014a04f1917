"""Pins the CPU oracle against every known-answer test the reference holds for the path.

Each test re-expresses one reference test (file:line in the docstring) with the same
inputs, expected values and tolerances.  CPU only.
"""
import numpy as np
import pytest


# ---------------------------------------------------------------- src/lib.rs:28-40
def test_hamming_distance(oracle):
    """src/lib.rs:28-40 test_hamming_distance"""
    for i in range(255):
        assert oracle.hamming_distance(i, i) == 0
    assert oracle.hamming_distance(0xFFFFFFFF, 0x0) == 32
    assert oracle.hamming_distance(0x0, 0xFFFFFFFFFFFFFFFF) == 64
    assert oracle.hamming_distance(0b10000000_00000000_00000000_00000000, 0b01000000_00000000_00000000_00000000) == 2


# ---------------------------------------------------------------- src/dictionaries.rs:239-281
def test_tau_sanity(oracle, dicts):
    """src/dictionaries.rs:239-243: ARUCO_DEFAULT.tau == 3 (and the table agrees with its own min distance)"""
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    assert d._tau == 3
    assert oracle.calculate_tau(d.code_list) == 3


def test_find_nearest_aruco_default(oracle, dicts):
    """src/dictionaries.rs:245-269"""
    codes = dicts.new_from_named_dict("ARUCO_DEFAULT").code_list
    assert oracle.find_nearest(codes, 0x1084210) == (0, 0)
    assert oracle.find_nearest(codes, 0x1084209) == (2, 0)
    assert oracle.find_nearest(codes, 0b00000001_00001000_01000010_00001001) == (2, 0)
    assert oracle.find_nearest(codes, 0b00000001_00001000_01000010_10001001) == (2, 1)
    assert oracle.find_nearest(codes, 0x1084217) == (1, 0)


def test_try_find_nearest_aruco_default(oracle, dicts):
    """src/dictionaries.rs:271-281: dist < tau accepts a 2-bit error, rejects an 8-bit one"""
    d = dicts.new_from_named_dict("ARUCO_DEFAULT")
    idx, dist = oracle.find_nearest(d.code_list, 0b01100001_00001000_01000010_00001001)
    assert dist < 3 and idx == 2
    # the reference literal 0b11111111_0000100_01000010_00001001 has a 7-digit group; keep it verbatim
    idx, dist = oracle.find_nearest(d.code_list, int("11111111" "0000100" "01000010" "00001001", 2))
    assert not dist < 3


def test_dictionary_tables(dicts):
    """src/dictionaries.rs:5-19,30-113: counts / bits / tau of every table, first and last codes"""
    expect = {
        "ARUCO": (1023, 25, 3), "ARUCO_DEFAULT": (1023, 25, 3), "ARUCO_MIP_16H3": (250, 16, 3),
        "ARUCO_MIP_25H7": (100, 25, 7), "ARUCO_MIP_36H12": (250, 36, 12), "APRILTAG_16H5": (30, 16, 5),
        "APRILTAG_25H7": (242, 25, 7), "APRILTAG_25H9": (35, 25, 9), "APRILTAG_36H9": (5329, 36, 9),
        "APRILTAG_36H10": (2320, 36, 10), "APRILTAG_36H11": (587, 36, 11), "ARTAG": (1024, 36, 0),
        "ARTOOLKITPLUS": (512, 36, 0), "ARTOOLKITPLUSBCH": (4096, 36, 0), "CHILITAGS": (1024, 64, 5),
    }
    assert sorted(dicts.get_dictionary_names()) == sorted(expect)
    for name, (count, bits, tau) in expect.items():
        d = dicts.new_from_named_dict(name.lower())  # to_ascii_uppercase, src/dictionaries.rs:141
        assert (d.code_list.size, d.num_bits, d._tau) == (count, bits, tau), name
    a = dicts.new_from_named_dict("ARUCO").code_list
    assert (int(a[0]), int(a[-1])) == (0x1084210, 0xE739C9)
    t = dicts.new_from_named_dict("APRILTAG_36H11").code_list
    assert (int(t[0]), int(t[-1])) == (0xD5D628584, 0xE83BE4B73)
    with pytest.raises(KeyError):
        dicts.new_from_named_dict("NOPE")


def test_mark_size(oracle, dicts):
    """src/dictionaries.rs:154-156: 16->6, 25->7, 36->8, 64->10"""
    for bits, size in ((16, 6), (25, 7), (36, 8), (64, 10)):
        assert oracle.mark_size(bits) == size
    assert dicts.new_from_named_dict("CHILITAGS").get_mark_size() == 10


def test_make_binary_image(oracle, dicts):
    """src/dictionaries.rs:212-232: LSB-first cells inside a one-cell black frame (quirk Q6)"""
    d = dicts.new_from_named_dict("ARUCO")
    for mid in (0, 1, 500, 1022):
        w, cells = oracle.make_binary_image(int(d.code_list[mid]), d.num_bits)
        w2, bits = d.make_binary_image(mid)
        assert w == w2 == 7 and len(bits) == 49
        assert np.array_equal(cells.reshape(-1), np.array(bits, dtype=np.uint8))
        assert not cells[0].any() and not cells[-1].any() and not cells[:, 0].any() and not cells[:, -1].any()
        code = int(d.code_list[mid])
        for i in range(25):
            assert cells[1 + i // 5, 1 + i % 5] == (code >> i) & 1


# ---------------------------------------------------------------- src/aruco.rs:400-459
def test_enforce_clockwise(oracle):
    """src/aruco.rs:400-412: both windings come out identical"""
    a = [(0, 0), (0, 1), (1, 1), (1, 0)]
    b = [(0, 0), (1, 0), (1, 1), (0, 1)]
    out = oracle.enforce_clockwise_corners(np.array([a, b], dtype=np.uint32))
    assert np.array_equal(out[0], out[1])
    assert out[1].tolist() == [list(p) for p in b]  # positive cross product: untouched


def test_bit_rotate(oracle):
    """src/aruco.rs:414-444: 90 degrees counter-clockwise"""
    pre = np.array([[1, 1, 1], [1, 0, 0], [0, 1, 0]], dtype=np.uint8)
    post = np.array([[1, 0, 0], [1, 0, 1], [1, 1, 0]], dtype=np.uint8)
    assert np.array_equal(oracle.rotate_bit_matrix(pre), post)
    pre = np.array([[1, 1, 1, 1], [1, 1, 1, 0], [1, 1, 0, 0], [1, 0, 0, 0]], dtype=np.uint8)
    post = np.array([[1, 0, 0, 0], [1, 1, 0, 0], [1, 1, 1, 0], [1, 1, 1, 1]], dtype=np.uint8)
    assert np.array_equal(oracle.rotate_bit_matrix(pre), post)


def test_drop_too_near(oracle):
    """src/aruco.rs:446-459: four near-identical quads, min_distance 10 -> one survives"""
    pts = np.array([
        [(0, 0), (10, 0), (10, 10), (0, 10)],
        [(1, 0), (10, 0), (10, 10), (0, 10)],
        [(0, 0), (10, 2), (10, 10), (0, 10)],
        [(0, 0), (10, 0), (10, 10), (3, 10)],
    ], dtype=np.uint32)
    kept, idx = oracle.discard_too_near(pts, 10.0)
    assert len(kept) == 1


# ---------------------------------------------------------------- src/pose.rs:379-598
def test_marker_transforms(oracle):
    """src/pose.rs:379-392"""
    rot = np.eye(3, dtype=np.float32)
    rot[0, 0] = 0.0; rot[0, 2] = 1.0; rot[2, 0] = 1.0; rot[2, 2] = 0.0
    out = oracle.apply_transform(rot, [1.0, 2.0, 3.0], [(0, 0, 0), (7, 11, 13)])
    assert out.tolist() == [[1.0, 2.0, 3.0], [14.0, 13.0, 10.0]]


def test_marker_identity_random(oracle):
    """src/pose.rs:394-439: inverse(forward(p)) within 1e-5 L1 for 100 poses x 100 points (seeded here)"""
    rng = np.random.default_rng(1234)
    failures = 0
    for _ in range(100):
        t = rng.random(3, dtype=np.float32)
        row1 = np.array([1.0 + rng.random(), 1.0 + rng.random(), 0.0], dtype=np.float32)
        row1 /= np.linalg.norm(row1)
        row2 = np.array([0.0, 1.1 + rng.random(), 1.0 + rng.random()], dtype=np.float32)
        row2 /= np.linalg.norm(row2)
        row3 = np.cross(row1, row2)
        row3 /= np.linalg.norm(row3)
        for _ in range(10):
            row2 = np.cross(row1, row3)
            row1 = np.cross(row3, row2)
        rot = np.stack([row1, row2, row3], axis=1).astype(np.float32)  # set_column(0..2)
        pts = rng.random((100, 3), dtype=np.float32)
        fwd = oracle.apply_transform(rot, t, pts)
        back = oracle.apply_transform(rot, t, fwd, inverse=True)
        failures += int((np.abs(pts - back).sum(axis=1) > 1e-5).sum())
    assert failures == 0


def test_gen_marker_square(oracle):
    """src/pose.rs:441-455"""
    sq = oracle.make_marker_square(11.0)
    assert sq[:, :2].tolist() == [[-5.5, 5.5], [5.5, 5.5], [5.5, -5.5], [-5.5, -5.5]]
    assert not sq[:, 2].any()


SQUARE_PTS = [(0.1, 0.1), (0.3, 0.1), (0.3, 0.3), (0.1, 0.3)]


def test_homography_solve(oracle):
    """src/pose.rs:457-474"""
    expected = np.array([
        [0.01818181818181819, 0.0, 0.2],
        [9.856383386231859e-19, -0.01818181818181819, 0.2000000000000001],
        [1.577021341797097e-17, -1.577021341797097e-17, 1.0]])
    h = oracle.compute_homography_from_marker_square(11.0, SQUARE_PTS)
    assert np.abs(h - expected).sum() < 1e-5


def test_canonical_solve(oracle):
    """src/pose.rs:476-512"""
    square = oracle.make_marker_square(11.0)
    h = oracle.compute_homography_from_marker_square(11.0, SQUARE_PTS)
    (ea, ra, ta), (eb, rb, tb) = oracle.solve_canonical_form(square, SQUARE_PTS, h)
    pose_a = np.array([
        [1.0, -2.775557561562891e-17, 1.02695629777827e-15, 10.99999999999999],
        [7.632783294297951e-17, -1.0, 1.02695629777827e-15, 11.0],
        [1.02695629777827e-15, -9.992007221626409e-16, -1.0, 54.99999999999996]])
    pose_b = np.array([
        [0.9259259259259256, 0.07407407407407443, -0.3703703703703712, 10.79629629629629],
        [-0.0740740740740744, -0.9259259259259256, -0.3703703703703713, 10.79629629629629],
        [-0.3703703703703712, 0.3703703703703713, -0.8518518518518512, 54.99999999999999]])
    assert np.abs(ra - pose_a[:, :3]).sum() < 1e-5
    assert np.abs(rb - pose_b[:, :3]).sum() < 1e-5
    assert np.abs(ta - pose_a[:, 3]).sum() < 1e-4
    assert np.abs(tb - pose_b[:, 3]).sum() < 1e-4


PA_ROT = np.array([
    [0.07313995850727262, 0.2953796077825095, 0.9525762089070907],
    [0.9973210134149258, -0.02055233410014844, -0.07020254813082821],
    [-0.001158736630905738, 0.9551588814795613, -0.2960914866390682]])
PB_ROT = np.array([
    [0.05174977302896467, 0.1311239186581316, -0.9900143832021767],
    [0.9667844474723887, -0.2550432732960733, 0.01675592050389792],
    [-0.2502994069448807, -0.957997623536802, -0.1399669967559523]])


def test_e2e_pose(oracle):
    """src/pose.rs:514-552"""
    (ea, ra, ta), (eb, rb, tb) = oracle.solve_with_undistorted_points([(90, 89), (95, 150), (80, 170), (75, 90)], 17.0, (1000, 1000))
    assert np.abs(ra - PA_ROT).sum() < 2e-5
    assert np.abs(rb - PB_ROT).sum() < 2e-5
    assert np.abs(ta - np.array([20.32196265994096, 29.69316666108512, 238.3658341694123])).sum() < 0.0005
    assert np.abs(tb - np.array([19.85146615649354, 29.20013946746331, 234.3277337340188])).sum() < 0.0005
    assert ea <= eb  # src/pose.rs:76-80


def test_e2e_pose2(oracle):
    """src/pose.rs:554-598"""
    pts = [(-0.090, -0.089), (-0.095, -0.150), (-0.080, -0.170), (-0.075, -0.090)]
    h = oracle.compute_homography_from_marker_square(19.0, pts)
    expected_h = np.array([
        [0.0001197249881460392, -0.00193812233285917, -0.08585585585585585],
        [-0.003084400189663352, -0.00115457562825984, -0.1225675675675677],
        [-0.004504504504504568, 0.01351351351351346, 1.0]])
    assert np.abs(h - expected_h).max() <= 1e-5
    (ea, ra, ta), (eb, rb, tb) = oracle.solve_with_normalized_points(pts, 19.0)
    sign = np.array([[-1, -1, -1], [-1, -1, -1], [1, 1, 1]])
    assert np.abs(ra - PA_ROT * sign).max() <= 1e-5
    assert np.abs(rb - PB_ROT * sign).max() <= 1e-5
    assert np.abs(ta - np.array([-22.712781796404, -33.18648038591866, 266.408873483460])).max() <= 1e-3
    assert np.abs(tb - np.array([-22.18693276313984, -32.6354499930472, 261.8957024086092])).max() <= 1e-3
