#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz: small input frames and the CPU oracle's outputs for them.

The reference (Rust) cannot be built or run in this environment (no cargo/rustc, dependencies not
vendored), so these vectors come from oracle/a3_oracle.c, whose pose/dictionary/helper stages are pinned by the
reference's own known-answer tests (tests/test_oracle_kat.py) and whose image stages are "parity unpinned"
(see oracle/a3_oracle.h).  They freeze the oracle's behaviour so that a change to it is noticed, and they
travel to the GPU box where the HIP path must reproduce them bit for bit.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))

from aruco3_amd import synth  # noqa: E402
from aruco3_amd.dictionaries import ARDictionary  # noqa: E402
from oracle import a3oracle  # noqa: E402

OUT = Path(__file__).resolve().parent


def pack(name, img, dict_name, extra=None):
    d = ARDictionary.new_from_named_dict(dict_name)
    r = a3oracle.detect(img, d.code_list, d.num_bits, d._tau)
    m = r["markers"]
    np.savez_compressed(
        OUT / f"{name}.npz",
        image=img,
        dictionary=np.array(dict_name),
        grey=r["grey"], thresholded=np.packbits(r["thresholded"] > 0, axis=1), thr_shape=np.array(r["thresholded"].shape),
        candidates_pre=r["candidates_pre"], candidates=r["candidates"], homographies=r["homographies"],
        homography_ok=r["homography_ok"], decode_ok=r["decode_ok"], codes=r["codes"],
        n_contours=np.array(r["n_contours"]),
        marker_id=np.array([x["id"] for x in m], dtype=np.uint32),
        marker_code=np.array([x["code"] for x in m], dtype=np.uint64),
        marker_corners=np.array([x["corners"] for x in m], dtype=np.uint32).reshape(-1, 4, 2),
        marker_hamming=np.array([x["hamming_distance"] for x in m], dtype=np.uint8),
        marker_rotation=np.array([x["rotation"] for x in m], dtype=np.uint8),
        **(extra or {}),
    )
    print(name, img.shape, "contours", r["n_contours"], "cand", len(r["candidates_pre"]), "->", len(r["candidates"]), "markers", [x["id"] for x in m])


def main():
    # C1: the reference's own CPU-runnable case, 640x480, 4 ARUCO_DEFAULT markers
    frames, truth = synth.config_frames(1, 1)
    pack("c1_640x480_aruco", frames[0], "ARUCO_DEFAULT", {"truth_ids": np.array([t.id for t in truth[0]])})
    # a reduced C4: APRILTAG_36H11, rotation +-15 degrees, gaussian noise sigma 8 -> noisy binary image, many tiny contours
    spec = synth.SynthSpec(480, 360, n_markers=(2, 2), side=(110.0, 140.0), rotation_deg=(-15.0, 15.0), perspective=0.0,
                           min_center_sep=200.0, noise_sigma=8.0)
    d = ARDictionary.new_from_named_dict("APRILTAG_36H11")
    img, truth = synth.render_frame(spec, d.code_list, d.num_bits, synth.frame_seed(4, 0))
    pack("c4_480x360_apriltag_noise", img, "APRILTAG_36H11", {"truth_ids": np.array([t.id for t in truth])})
    # C0: the reference bench's input recipe (uniform random RGB), reduced to 192x160
    pack("c0_192x160_noise", synth.noise_frame(192, 160, synth.frame_seed(0, 0)), "ARUCO")
    # odd geometry: width not a multiple of 4, markers rotated, RGBA input
    spec = synth.SynthSpec(333, 251, n_markers=(1, 1), side=(120.0, 120.0), rotation_deg=(20.0, 40.0), min_center_sep=10.0)
    d = ARDictionary.new_from_named_dict("ARUCO")
    img, truth = synth.render_frame(spec, d.code_list, d.num_bits, synth.frame_seed(9, 3))
    rgba = np.concatenate([img, np.full(img.shape[:2] + (1,), 255, np.uint8)], axis=2)
    pack("odd_333x251_rgba", rgba, "ARUCO", {"truth_ids": np.array([t.id for t in truth])})


if __name__ == "__main__":
    main()
