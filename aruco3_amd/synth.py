"""Deterministic synthetic frames for the BASELINE.json configurations.

Host-side test/bench utility (numpy), not part of the detection path.  Markers are drawn
with the cell order the decoder reads (row-major, first inner cell = most significant
bit, src/aruco.rs:296-308) so corner 0 of a detection is the marker's true top-left; the
`lsb` cell order reproduces `ARDictionary::make_binary_image` (src/dictionaries.rs:212-232,
quirk Q6: the same marker seen rotated by 180 degrees).  `noise_frame` is the reference
bench's own input recipe (benches/detect_markers.rs:29-45): uniform random RGB.
"""
import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

_M64 = (1 << 64) - 1


class SplitMix64:
    """Tiny seeded generator for the scalar draws, independent of numpy's Generator."""

    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return lo + (hi - lo) * ((self.next() >> 11) / float(1 << 53))

    def randint(self, lo: int, hi: int) -> int:
        """inclusive bounds"""
        return lo + self.next() % (hi - lo + 1)


def frame_seed(config: int, frame_idx: int) -> int:
    """seed = 0xA3C0DE00 + config*1000 + frame_idx"""
    return 0xA3C0DE00 + config * 1000 + frame_idx


@dataclass
class TruthMarker:
    id: int
    corners: np.ndarray  # (4,2) float, TL,TR,BR,BL of the black frame in image coordinates


@dataclass
class SynthSpec:
    width: int
    height: int
    n_markers: Tuple[int, int] = (4, 8)
    side: Tuple[float, float] = (140.0, 280.0)
    rotation_deg: Tuple[float, float] = (0.0, 360.0)
    perspective: float = 0.12   # max relative shrink of one side (0 = pure similarity)
    min_center_sep: float = 300.0
    noise_sigma: float = 0.0
    background: str = "gradient"  # "flat" | "gradient"
    cell_order: str = "msb"       # "msb" (decoder order) | "lsb" (make_binary_image order)
    grid: Tuple[int, int] = (0, 0)  # (cols, rows) > 0: place markers on a regular grid (config C5)
    white: int = 235            # only used when paper=False
    black: int = 25
    paper: bool = True           # True: black cells are "printed" on the background (white = background);
                                 # False: a sticker with its own white level and a one-cell quiet zone
    supersample: int = 3


def marker_cells(code: int, num_bits: int, cell_order: str = "msb") -> np.ndarray:
    """n x n cells (1 = white) including the one-cell black frame."""
    k = int(math.ceil(math.sqrt(float(num_bits))))
    n = k + 2
    cells = np.zeros((n, n), dtype=np.uint8)
    for r in range(k):
        for c in range(k):
            i = r * k + c
            if i >= num_bits:
                continue
            bit = (code >> (num_bits - 1 - i)) & 1 if cell_order == "msb" else (code >> i) & 1
            cells[1 + r, 1 + c] = bit
    return cells


def _homography(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    a = []
    b = []
    for (x, y), (u, v) in zip(src, dst):
        a.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        a.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.array(a, dtype=np.float64), np.array(b, dtype=np.float64))
    return np.append(h, 1.0).reshape(3, 3)


def _draw_marker(gray: np.ndarray, cells: np.ndarray, quad: np.ndarray, spec: SynthSpec) -> None:
    """Paint cells (plus a one-cell white quiet zone) into `gray` (float32 HxW) through the quad's homography."""
    n = cells.shape[0]
    h, w = gray.shape
    # marker coordinates: black frame spans [0,n]^2, quiet zone [-1,n+1]^2
    src = np.array([[0, 0], [n, 0], [n, n], [0, n]], dtype=np.float64)
    H = _homography(src, quad)
    Hinv = np.linalg.inv(H)
    outer = (H @ np.array([[-1, -1, 1], [n + 1, -1, 1], [n + 1, n + 1, 1], [-1, n + 1, 1]], dtype=np.float64).T).T
    outer = outer[:, :2] / outer[:, 2:3]
    x0 = max(int(math.floor(outer[:, 0].min())) - 1, 0)
    x1 = min(int(math.ceil(outer[:, 0].max())) + 2, w)
    y0 = max(int(math.floor(outer[:, 1].min())) - 1, 0)
    y1 = min(int(math.ceil(outer[:, 1].max())) + 2, h)
    if x1 <= x0 or y1 <= y0:
        return
    ss = spec.supersample
    offs = (np.arange(ss, dtype=np.float64) + 0.5) / ss - 0.5
    ys, xs = np.mgrid[y0:y1, x0:x1].astype(np.float64)
    acc = np.zeros(ys.shape, dtype=np.float64)
    cov = np.zeros(ys.shape, dtype=np.float64)
    padded = np.ones((n + 2, n + 2), dtype=np.float64)
    padded[1:-1, 1:-1] = cells
    for oy in offs:
        for ox in offs:
            px, py = xs + ox, ys + oy
            d = Hinv[2, 0] * px + Hinv[2, 1] * py + Hinv[2, 2]
            u = (Hinv[0, 0] * px + Hinv[0, 1] * py + Hinv[0, 2]) / d
            v = (Hinv[1, 0] * px + Hinv[1, 1] * py + Hinv[1, 2]) / d
            inside = (u >= -1) & (u < n + 1) & (v >= -1) & (v < n + 1)
            ui = np.clip(np.floor(u).astype(np.int64) + 1, 0, n + 1)
            vi = np.clip(np.floor(v).astype(np.int64) + 1, 0, n + 1)
            val = padded[vi, ui]
            if spec.paper:
                hit = inside & (val == 0)
                acc += np.where(hit, float(spec.black), 0.0)
                cov += hit
            else:
                acc += np.where(inside, spec.black + (spec.white - spec.black) * val, 0.0)
                cov += inside
    k = float(ss * ss)
    region = gray[y0:y1, x0:x1]
    region[...] = (region * (k - cov) + acc) / k


def frame_layout(spec: SynthSpec, codes: np.ndarray, num_bits: int, seed: int):
    """Every random choice of one frame (the only consumer of the seed): background (base, gx, gy) and, per marker,
    (quad 4x2 in image coordinates, dictionary index, n x n cells).  Shared by the host and the device renderer."""
    rng = SplitMix64(seed)
    w, h = spec.width, spec.height
    if spec.background == "gradient":
        gx = rng.uniform(-12.0, 12.0)
        gy = rng.uniform(-12.0, 12.0)
        base = rng.uniform(185.0, 215.0)
    else:
        gx, gy, base = 0.0, 0.0, 200.0
    markers = []
    centers: List[Tuple[float, float]] = []
    gc, gr = spec.grid
    count = gc * gr if gc and gr else rng.randint(*spec.n_markers)
    for k in range(count):
        side = rng.uniform(*spec.side)
        half_diag = side * 0.5 * math.sqrt(2.0) * (1.0 + 2.0 / (math.ceil(math.sqrt(num_bits)) + 2)) + 4.0
        if gc and gr:
            cx = (k % gc + 0.5) * w / gc
            cy = (k // gc + 0.5) * h / gr
        else:
            ok = False
            for _ in range(200):
                cx = rng.uniform(half_diag, w - half_diag)
                cy = rng.uniform(half_diag, h - half_diag)
                if all((cx - px) ** 2 + (cy - py) ** 2 >= spec.min_center_sep ** 2 for px, py in centers):
                    ok = True
                    break
            if not ok:
                continue
        centers.append((cx, cy))
        ang = math.radians(rng.uniform(*spec.rotation_deg))
        ca, sa = math.cos(ang), math.sin(ang)
        sq = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]], dtype=np.float64) * side
        # mild perspective: shrink the edge opposite to a random side, in the marker's own frame
        if spec.perspective > 0.0:
            f = 1.0 - rng.uniform(0.0, spec.perspective)
            which = rng.randint(0, 3)
            a, b = which, (which + 1) % 4
            mid = (sq[a] + sq[b]) / 2.0
            sq[a] = mid + (sq[a] - mid) * f
            sq[b] = mid + (sq[b] - mid) * f
        rot = np.array([[ca, -sa], [sa, ca]])
        quad = sq @ rot.T + np.array([cx, cy])
        mid_ = rng.randint(0, len(codes) - 1)
        markers.append((quad, mid_, marker_cells(int(codes[mid_]), num_bits, spec.cell_order)))
    return base, gx, gy, markers


def render_frame(spec: SynthSpec, codes: np.ndarray, num_bits: int, seed: int):
    """-> (HxWx3 uint8 RGB frame, [TruthMarker])"""
    w, h = spec.width, spec.height
    base, gx, gy, markers = frame_layout(spec, codes, num_bits, seed)
    if spec.background == "gradient":
        xs = np.linspace(-1.0, 1.0, w, dtype=np.float32)[None, :]
        ys = np.linspace(-1.0, 1.0, h, dtype=np.float32)[:, None]
        gray = (base + gx * xs + gy * ys).astype(np.float32)
    else:
        gray = np.full((h, w), 200.0, dtype=np.float32)
    truth: List[TruthMarker] = []
    for quad, mid_, cells in markers:
        _draw_marker(gray, cells, quad, spec)
        truth.append(TruthMarker(mid_, quad))

    rgb = np.empty((h, w, 3), dtype=np.float32)
    # slight, fixed channel tint so that the luma weights matter
    rgb[..., 0] = gray * 1.00
    rgb[..., 1] = gray * 0.98
    rgb[..., 2] = gray * 0.94
    if spec.noise_sigma > 0.0:
        g = np.random.Generator(np.random.PCG64(seed))
        rgb += g.standard_normal(rgb.shape, dtype=np.float32) * np.float32(spec.noise_sigma)
    np.clip(rgb, 0.0, 255.0, out=rgb)
    return np.rint(rgb).astype(np.uint8), truth


SYNTH_MARKER_DTYPE = np.dtype([("hinv", np.float32, 9), ("x0", np.int32), ("y0", np.int32), ("x1", np.int32), ("y1", np.int32),
                               ("cells", np.uint64, 2), ("n", np.uint32), ("reserved", np.uint32)], align=True)   # C layout: 80 bytes
SYNTH_FRAME_DTYPE = np.dtype([("base", np.float32), ("gx", np.float32), ("gy", np.float32), ("noise_sigma", np.float32),
                              ("first_marker", np.uint32), ("n_markers", np.uint32), ("seed", np.uint64)], align=True)       # 32 bytes


def device_layout(spec: SynthSpec, codes: np.ndarray, num_bits: int, seeds):
    """The layouts of `frame_layout` for a list of seeds as the two record arrays a3_synth_render takes
    (include/aruco3_hip.h) -> (frames, markers, [[TruthMarker]])."""
    w, h = spec.width, spec.height
    frames = np.zeros(len(seeds), dtype=SYNTH_FRAME_DTYPE)
    recs, truths = [], []
    for fi, seed in enumerate(seeds):
        base, gx, gy, markers = frame_layout(spec, codes, num_bits, seed)
        frames[fi] = (base, gx, gy, spec.noise_sigma, len(recs), len(markers), seed & _M64)
        truth = []
        for quad, mid_, cells in markers:
            n = cells.shape[0]
            src = np.array([[0, 0], [n, 0], [n, n], [0, n]], dtype=np.float64)
            H = _homography(src, quad)
            outer = (H @ np.array([[-1, -1, 1], [n + 1, -1, 1], [n + 1, n + 1, 1], [-1, n + 1, 1]], dtype=np.float64).T).T
            outer = outer[:, :2] / outer[:, 2:3]
            x0 = max(int(math.floor(outer[:, 0].min())) - 1, 0); x1 = min(int(math.ceil(outer[:, 0].max())) + 2, w)
            y0 = max(int(math.floor(outer[:, 1].min())) - 1, 0); y1 = min(int(math.ceil(outer[:, 1].max())) + 2, h)
            bits = 0
            for r in range(n):
                for c in range(n):
                    bits |= int(cells[r, c]) << (r * n + c)
            recs.append((np.linalg.inv(H).astype(np.float32).reshape(9), x0, y0, x1, y1, (bits & _M64, bits >> 64), n, 0))   # n <= 11
            truth.append(TruthMarker(mid_, quad))
        truths.append(truth)
    marr = np.zeros(max(len(recs), 1), dtype=SYNTH_MARKER_DTYPE)
    for i, r in enumerate(recs):
        marr[i] = r
    return frames, marr[: len(recs)] if recs else marr[:0], truths


def render_frames_device(spec: SynthSpec, codes: np.ndarray, num_bits: int, seeds, out=None, device: int = 0):
    """Render the frames of `seeds` on the GPU (a3_synth_render) into a CUDA uint8 tensor (N,H,W,3) -> (tensor, truths).
    Same layouts (markers, ids, positions) as render_frame; pixel values may differ by a grey level (f32 vs f64 painting) and
    the noise, if any, comes from a different generator."""
    import torch

    from . import _lib

    frames, markers, truths = device_layout(spec, codes, num_bits, seeds)
    if out is None:
        out = torch.empty((len(seeds), spec.height, spec.width, 3), dtype=torch.uint8, device=torch.device("cuda", device))
    _lib.synth_render(device, frames, markers, spec.width, spec.height, spec.paper, float(spec.black), float(spec.white), spec.supersample,
                      out.data_ptr(), spec.width * 3, spec.width * spec.height * 3)
    return out, truths


def noise_frame(width: int, height: int, seed: int) -> np.ndarray:
    """benches/detect_markers.rs:36-45: every channel of every pixel uniform random u8."""
    g = np.random.Generator(np.random.PCG64(seed))
    return g.integers(0, 256, size=(height, width, 3), dtype=np.uint8)


# ---- the BASELINE.json configurations ----------------------------------------------------

def config_spec(config: int) -> Tuple[SynthSpec, str]:
    """-> (spec, dictionary name) for C1..C5 of SURVEY.md section 8d."""
    if config == 1:
        return SynthSpec(640, 480, n_markers=(4, 4), side=(84.0, 84.0), rotation_deg=(0.0, 0.0), perspective=0.0,
                         min_center_sep=150.0), "ARUCO_DEFAULT"
    if config in (2, 3):
        return SynthSpec(1920, 1080, n_markers=(4, 8), side=(140.0, 280.0), min_center_sep=330.0), "ARUCO"
    if config == 4:
        return SynthSpec(1280, 720, n_markers=(4, 4), side=(120.0, 200.0), rotation_deg=(-15.0, 15.0), perspective=0.0,
                         min_center_sep=260.0, noise_sigma=8.0), "APRILTAG_36H11"
    if config == 5:
        return SynthSpec(3840, 2160, side=(260.0, 380.0), rotation_deg=(-20.0, 20.0), perspective=0.2, grid=(4, 4)), "ARUCO"
    raise ValueError(config)


def config_frames(config: int, count: int, first: int = 0):
    """Generate `count` frames of one configuration -> (N,H,W,3) uint8, [[TruthMarker]]"""
    from .dictionaries import ARDictionary

    spec, name = config_spec(config)
    d = ARDictionary.new_from_named_dict(name)
    frames = np.empty((count, spec.height, spec.width, 3), dtype=np.uint8)
    truths = []
    for i in range(count):
        frames[i], t = render_frame(spec, d.code_list, d.num_bits, frame_seed(config, first + i))
        truths.append(t)
    return frames, truths
