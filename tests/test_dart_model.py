"""The parallel contour formulation the HIP kernels implement (tests/dart_model.py) must
reproduce the oracle's sequential border following: same contours, same order, same start
point -- including the start-selection anomalies of components that touch column 0."""
import numpy as np
import pytest

from tests.dart_model import contours_by_darts


def _isolated(img, c):
    h, w = img.shape
    if len(c) != 1:
        return False
    x, y = c[0]
    return not any(
        (dx or dy) and 0 <= x + dx < w and 0 <= y + dy < h and img[y + dy, x + dx]
        for dy in (-1, 0, 1) for dx in (-1, 0, 1))


def _ref(oracle, img):
    cs, _, _ = oracle.find_contours(img)
    ref = [[tuple(map(int, p)) for p in c] for c in cs]
    # the kernels ignore isolated pixels: their 1-point contours can never become a quad
    return [c for c in ref if not _isolated(img, c)]


def _images(rng, count):
    for trial in range(count):
        h = int(rng.integers(1, 28))
        w = int(rng.integers(1, 28))
        kind = trial % 4
        if kind == 0:
            img = rng.random((h, w)) < rng.choice([0.3, 0.5, 0.6, 0.7, 0.85, 0.95])
        elif kind == 1:
            a = rng.random((h + 4, w + 4))
            s = sum(np.roll(np.roll(a, dy, 0), dx, 1) for dy in (-1, 0, 1) for dx in (-1, 0, 1))[2:-2, 2:-2]
            img = s > rng.uniform(3.5, 5.5)
        elif kind == 2:
            img = np.ones((h, w), bool)
            for _ in range(int(rng.integers(1, 6))):
                x0 = int(rng.integers(0, w)); y0 = int(rng.integers(0, h))
                x1 = int(rng.integers(x0, w)) + 1; y1 = int(rng.integers(y0, h)) + 1
                img[y0:y1, x0:x1] = False
                if rng.random() < 0.5 and x1 - x0 > 2 and y1 - y0 > 2:
                    img[y0 + 1:y1 - 1, x0 + 1:x1 - 1] = True
        else:
            img = rng.random((h, w)) < 0.6
            img[:, 0] = rng.random(h) < 0.8
            if w > 1:
                img[:, 1] = rng.random(h) < 0.5
        yield img.astype(np.uint8) * 255


@pytest.mark.parametrize("rule", ["any8", "border4", "pdart"])
def test_dart_cycles_equal_sequential_border_following(oracle, rule):
    rng = np.random.default_rng(20261003)
    resolved = 0
    for img in _images(rng, 600):
        got, st = contours_by_darts(img, node_rule=rule)
        assert st["chain_events"] == 0
        assert got == _ref(oracle, img), (img > 0).astype(int)
        resolved += st["iters"] > 1
    assert resolved > 0  # the sample must exercise the start-resolution fixpoint


def test_known_anomaly_column0(oracle):
    """A component whose first pixel sits in column 0 gets no outer start there (the x > 0 guard);
    its border is then picked up later, possibly as a 'hole' start."""
    img = np.zeros((5, 6), np.uint8)
    img[0:3, 0:3] = 255
    img[1, 1] = 0
    got, st = contours_by_darts(img)
    assert got == _ref(oracle, img)
    assert len(got) == 2


def test_static_fire_cycles_start_at_their_natural_start():
    """The rule behind the borders k_local_contract finishes early (kDead): a cycle whose smallest event passes the static test --
    a W-event on a pixel that owns exactly one dart, an E-event on a pixel without a W side, or (round 5) an E-event on a pixel
    through which a border passes that itself starts, by the first two rules, before the pixel's W-event -- starts there in the fixpoint of the
    start resolution, whatever happens to its neighbours; and it is a traced border (its key is finite).  Checked on the executable
    model over random, blob-like and stroke-like images, including dense noise where most cycles are a few darts long."""
    rng = np.random.default_rng(20261005)
    n_static = n_short_static = n_moved_elsewhere = n_wide_only = 0
    for img in _images(rng, 400):
        _, st = contours_by_darts(img, node_rule="pdart")
        for c, ok in st["static_ok_wide"].items():
            n_wide_only += ok and not st["static_ok"][c]
            if ok:
                n_static += 1
                n_short_static += st["cycle_len"].get(c, 99) < 21
                assert st["T_final"].get(c) == st["T_natural"][c], (c, st["T_final"].get(c), st["T_natural"][c])
            elif st["T_final"].get(c) != st["T_natural"][c]:
                n_moved_elsewhere += 1          # (borders whose start moves -- the column-0 anomalies -- must all be among the others)
    assert n_static > 2000 and n_short_static > 1000 and n_wide_only > 500   # (E-events with a witness: round 5's extension)


def test_natural_assignment_is_the_fixpoint_when_every_natural_start_fires():
    """k_cycle_select's inline check and k_local_contract's trust_natural rule: evaluate every border's smallest event under the
    assignment "every border starts at its smallest event"; if all of them fire, that assignment is the fixpoint of the start
    resolution (and a short border finished with early was rightly counted as traced); if one does not, some start moves -- the
    case the library answers with a re-run through the fixpoint passes."""
    rng = np.random.default_rng(20261003)      # (the sample of test_dart_cycles_equal_sequential_border_following: it holds cases of both kinds)
    n_all = n_some = 0
    for img in _images(rng, 600):
        _, st = contours_by_darts(img, node_rule="pdart")
        if all(st["natural_fires"].values()):
            n_all += 1
            assert st["T_final"] == st["T_natural"]
        else:
            n_some += 1
            assert st["T_final"] != st["T_natural"]
    assert n_all > 100 and n_some > 0
