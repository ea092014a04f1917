"""Shared comparison helpers for the parity tests."""
import numpy as np


def markers_of_oracle(res):
    return [(m["id"], m["code"], tuple(m["corners"]), m["hamming_distance"], m["rotation"]) for m in res["markers"]]


def markers_of_hip(arr):
    out = []
    for m in arr:
        c = m["corners"]
        out.append((int(m["id"]), int(m["code"]), tuple((int(c[2 * i]), int(c[2 * i + 1])) for i in range(4)), int(m["hamming_distance"]),
                    int(m["rotation"])))
    return out


def assert_frame_parity(ctx, frame_idx, img, res, w, h, check_patches=True):
    """Every stage the C ABI exposes for one frame of the last batch against the oracle's dict `res`."""
    grey = ctx.download_grey(frame_idx, w, h)
    assert np.array_equal(grey, res["grey"]), "grey differs"
    thr = ctx.download_grey(frame_idx, w, h, thresholded=True)
    bad = np.argwhere(thr != res["thresholded"])
    assert bad.size == 0, f"thresholded differs at {bad[:5].tolist()} ({len(bad)} px)"
    pre = ctx.candidates(frame_idx, before_discard=True)
    assert pre.tolist() == res["candidates_pre"].tolist(), "candidates before discard_too_near differ"
    fin = ctx.candidates(frame_idx)
    assert fin.tolist() == res["candidates"].tolist(), "candidates differ"
    patches, ok, codes, dec = ctx.homographies(frame_idx, with_patches=check_patches)
    assert ok.tolist() == res["homography_ok"].tolist()
    assert dec.tolist() == res["decode_ok"].tolist()
    assert codes.tolist() == res["codes"].tolist()
    if check_patches:
        assert np.array_equal(patches, res["homographies"]), "warped patches differ"


def marker_tuples(m):
    """a3_marker records as plain tuples (field values only: the 4 padding bytes of the 56-byte record are not data)"""
    return [(int(r["frame"]), int(r["id"]), int(r["code"]), tuple(int(v) for v in r["corners"]), int(r["hamming_distance"]),
             int(r["rotation"]), int(r["candidate_index"])) for r in m]


BENCH_LINE_BUDGET = 4096


def bench_output(p):
    """(line, detail) of a finished `bench.py` child: stdout must hold exactly ONE line -- the compact JSON line the driver parses,
    at most BENCH_LINE_BUDGET bytes -- and stderr a `bench_detail {...}` line with everything else."""
    import json

    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    # (RCCL prints its version banner on stdout: lines of other libraries may precede ours, never a second JSON line, and ours is last)
    assert len([ln for ln in lines if ln.startswith("{")]) == 1 and lines[-1].startswith("{"), p.stdout[-2000:]
    assert len(lines[-1].encode()) <= BENCH_LINE_BUDGET, len(lines[-1])
    line = json.loads(lines[-1])
    det = [ln for ln in p.stderr.splitlines() if ln.startswith("bench_detail ")]
    assert len(det) == 1, p.stderr[-2000:]
    return line, json.loads(det[0][len("bench_detail "):])
