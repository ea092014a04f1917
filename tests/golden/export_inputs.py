#!/usr/bin/env python3
"""Writes the input images of tests/golden/*.npz as raw interleaved bytes (RGB8 or RGBA8, row-major, no header) into
tests/fixtures/inputs/<name>.raw, plus manifest.txt (name width height channels dictionary per line): the form the Rust-side
fixture dumper (integration/dump_fixtures.rs) reads.  Inputs only -- the expected outputs come from a `cargo test` run of the
reference crate elsewhere ("Pinning the image stages")."""
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
OUT = HERE.parent / "fixtures" / "inputs"
OUT.mkdir(parents=True, exist_ok=True)
lines = []
for path in sorted(HERE.glob("*.npz")):
    z = np.load(path)
    img = np.ascontiguousarray(z["image"])
    h, w, c = img.shape
    (OUT / f"{path.stem}.raw").write_bytes(img.tobytes())
    lines.append(f"{path.stem} {w} {h} {c} {str(z['dictionary'])}")
(OUT / "manifest.txt").write_text("\n".join(lines) + "\n")
print("\n".join(lines))
