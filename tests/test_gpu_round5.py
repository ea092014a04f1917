"""Round-5 GPU tests: the burst stepping is the library's behaviour behind the public header (fresh process, defaults untouched,
a3_stats.stepping proves the hold path was taken); contexts that share a stream get the deferred decode by themselves; the bench's
per-rotation gather survives a lagging collective (write-after-read guard).  Everything goes through the C ABI.  GPU only."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from tests.util import bench_output, marker_tuples

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

# The documented rotation (include/aruco3_hip.h, a3_order_after), public calls only, in a process that never touches a switch.
_FRESH = r'''
import json, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
from aruco3_amd import _lib, synth
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

d = ARDictionary.new_from_named_dict("ARUCO_DEFAULT")
N = 4
bufs, args = [], []
for j in range(N):                       # every context steps a batch of its own
    f, _ = synth.config_frames(1, 6, first=6 * j)
    t = torch.from_numpy(f).cuda(); bufs.append(t)
    n, h, w, c = f.shape
    args.append((t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n))
ctxs = [Detector(DetectorConfig.default(), d)._context() for _ in range(N)]
tup = lambda m: [(int(x["id"]), int(x["code"]), tuple(int(v) for v in x["corners"]), int(x["hamming_distance"]), int(x["rotation"])) for x in m]
want = []
for k, cx in enumerate(ctxs):
    r = cx.detect_batch(*args[k]); cx.detect_batch(*args[k])          # (first batches of a shape are planned by the host)
    want.append((tup(r[0]), r[1].tolist()))
sync_stepping = [cx.stats()["stepping"] for cx in ctxs]

def submit(k):
    for m in range(k + 1, N):
        ctxs[k].order_after(ctxs[m])
    ctxs[k].submit(*args[k])

seen, equal = [], True
for k in range(N):
    submit(k)
for i in range(12):
    k = i %% N
    r = ctxs[k].collect()
    st = ctxs[k].stats()
    seen.append((k, st["stepping"], st["released_others"]))
    equal &= (tup(r[0]), r[1].tolist()) == want[k]
    if i + N < 12:
        submit(k)
print(json.dumps({"seen": seen, "equal": equal, "sync": sync_stepping, "markers": [len(w[0]) for w in want]}))
'''


def _fresh_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "A3_HIP_LIB"):
        env.pop(k, None)
    return env


def test_default_library_holds_chains_in_the_documented_rotation():
    """fresh process, no a3_debug_* call anywhere in it: contexts 0..2 of every rotation report 'held_released_by_last', context 3
    'burst_last' with three chains released; results equal the synchronous calls'"""
    assert "a3_debug" not in _FRESH and "debug_" not in _FRESH
    p = subprocess.run([sys.executable, "-c", _FRESH % {"root": str(ROOT)}], cwd=ROOT, env=dict(_fresh_env(), GPU_MAX_HW_QUEUES="8"), capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-4000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["equal"] is True and min(out["markers"]) > 0
    assert out["sync"] == ["whole"] * 4
    for k, stepping, released in out["seen"]:
        if k < 3:
            assert (stepping, released) == ("held_released_by_last", 0), out["seen"]
        else:
            assert (stepping, released) == ("burst_last", 3), out["seen"]


def test_shared_stream_contexts_get_the_deferred_decode_by_themselves(dicts):
    """library defaults: two contexts on ONE caller stream -> 'decode_deferred'; the same two contexts each on a caller stream of its
    own -> 'whole' without gates, held with gates; one context alone on a caller's stream -> 'whole'"""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig

    fa, _ = synth.config_frames(1, 5)
    da = torch.from_numpy(fa).cuda()
    n, h, w, c = fa.shape
    aa = (da.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    c0, c1 = (Detector(DetectorConfig(), dicts.new_from_named_dict("ARUCO_DEFAULT"))._context() for _ in range(2))
    want = c0.detect_batch(*aa); c0.detect_batch(*aa); c1.detect_batch(*aa); c1.detect_batch(*aa)
    same = lambda got: marker_tuples(got[0]) == marker_tuples(want[0]) and np.array_equal(got[1], want[1])
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    c0.set_stream(s0.cuda_stream)
    c0.submit(*aa); assert same(c0.collect()) and c0.stats()["stepping"] == "whole"           # alone on a caller's stream
    c1.set_stream(s0.cuda_stream)
    c0.submit(*aa); c1.submit(*aa)
    assert same(c0.collect()) and same(c1.collect())
    assert c0.stats()["stepping"] == "decode_deferred" and c1.stats()["stepping"] == "decode_deferred"
    c0.order_after(c1); c0.submit(*aa); c1.submit(*aa)                                           # a gate between them is a no-op
    assert same(c0.collect()) and same(c1.collect()) and c0.stats()["stepping"] == "decode_deferred"
    c1.set_stream(s1.cuda_stream)                                                                 # a caller stream each
    c0.submit(*aa); c1.submit(*aa)
    assert same(c0.collect()) and same(c1.collect())
    assert c0.stats()["stepping"] == "whole" and c1.stats()["stepping"] == "whole"
    c0.order_after(c1); c0.submit(*aa); c1.submit(*aa)                                           # ... and now the gate makes a burst of two
    assert same(c0.collect()) and same(c1.collect())
    assert c0.stats()["stepping"] == "held_released_by_last" and (c1.stats()["stepping"], c1.stats()["released_others"]) == ("burst_last", 1)
    c0.order_after(c1); c0.submit(*aa)                                                           # nobody comes: collect releases
    assert same(c0.collect()) and c0.stats()["stepping"] == "held_released_early"
    # the switches refuse to change while a chain is held or a decode stage deferred
    L = _lib.load()
    c0.order_after(c1); c0.submit(*aa)
    assert L.a3_debug_set_overlap(2) == _lib.ERR_INVALID and L.a3_debug_set_hold(0) == _lib.ERR_INVALID
    assert same(c0.collect())
    assert L.a3_debug_set_overlap(-1) == 0 and L.a3_debug_set_hold(1) == 0


def test_first_batch_of_a_shape_is_not_held(dicts):
    """a gated submit whose batch needs a host-side plan (first batch of a shape on the context) is enqueued whole: nothing that waits
    for the device ever runs under the burst lock"""
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig

    fa, _ = synth.config_frames(1, 3)
    da = torch.from_numpy(fa).cuda()
    n, h, w, c = fa.shape
    aa = (da.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)
    c0, c1 = (Detector(DetectorConfig(), dicts.new_from_named_dict("ARUCO_DEFAULT"))._context() for _ in range(2))
    c1.detect_batch(*aa)
    c0.order_after(c1); c0.submit(*aa)
    r = c0.collect()
    assert c0.stats()["stepping"] == "whole" and len(r[0]) > 0
    c0.order_after(c1); c0.submit(*aa); c1.submit(*aa)
    assert marker_tuples(c0.collect()[0]) == marker_tuples(r[0]) and c0.stats()["stepping"] == "held_released_by_last"
    assert marker_tuples(c1.collect()[0]) == marker_tuples(r[0])


def _bench(extra, timeout=900):
    cmd = [sys.executable, str(ROOT / "bench.py"), "--frames", "8", "--steps", "24", "--warmup", "4", "--repeats", "2", "--isolated-launches", "2",
           "--device-synth", "--no-other-workloads", "--no-cpu-baseline", "--gpus", "1", "--force-dist", "--backend", "nccl",
           "--verify-gathers", "--gather-delay-us", "4000"] + extra
    env = _fresh_env()
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    line, out = bench_output(p)
    assert line["gathered"]["collectives_with_wrong_records"] == out["gathered"]["collectives_with_wrong_records"]
    return out


def test_gather_survives_a_lagging_collective():
    """the RCCL branch with one rank, every collective held up for 4 ms (several rotations of these small batches) on the side stream,
    the batches changing hands every rotation so that consecutive rotations' records differ: with the write-after-read guard every
    collective delivers the records of ITS rotation; without it (round 4's bench) records are overwritten before they are sent --
    the check must be able to see that"""
    good = _bench([])
    g = good["gathered"]
    assert g["verified_collectives"] >= 12 and g["collectives_with_wrong_records"] == 0, g
    assert g["global_frame_indices_in_order"] is True and good["config"]["distinct_batches_in_flight"] == 4
    assert good["library"]["internal_switches_used"] == []
    bad = _bench(["--no-gather-backpressure"])
    gb = bad["gathered"]
    assert gb["collectives_with_wrong_records"] > 0, "the race check cannot see an overwritten record"


def test_borders_finished_early_are_counted_once_and_change_nothing(dicts, oracle):
    """Dense graphs (noise-like frames) take the round-5 path on which short borders that close inside a tile and start unconditionally
    are finished with in k_local_contract (kDead), windows that wrapped drop out of the doubling rounds, and the tighter length bound
    prunes: a3_stats.contours_traced of the PRODUCT path -- borders finished early + borders listed -- must equal the number of borders
    the reference follows (1-pixel specks aside), exactly as the tapped run (no early finish) counts them; candidates and markers
    must be those of the oracle.  Uniform noise (the reference's bench recipe), denser and sparser noise (many column-0 starts, where
    natural starts do not fire), and sigma-8 noise over rendered markers; small frames (one tile) up to 1080p (global entry rounds)."""
    from aruco3_amd import synth

    rng = np.random.default_rng(55)
    d = dicts.new_from_named_dict("ARUCO")
    det = _round5_detector(dicts)
    cases = []
    for (w, h, n), p in (((192, 160, 6), None), ((640, 360, 3), None), ((1920, 1080, 1), None), ((333, 217, 4), 0.62), ((256, 256, 4), 0.38)):
        if p is None:
            cases.append(rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8))
        else:   # grey levels arranged so that a fraction p of the pixels thresholds to foreground in blobs of a few pixels
            a = (rng.random((n, h, w)) < p).astype(np.uint8) * 255
            cases.append(np.repeat(a[..., None], 3, axis=3))
    spec4, name4 = synth.config_spec(4)
    d4 = dicts.new_from_named_dict(name4)
    cases.append(np.stack([synth.render_frame(spec4, d4.code_list, d4.num_bits, synth.frame_seed(4, 100 + i))[0] for i in range(2)]))
    # ... and a DENSE graph that needs the fixpoint passes (found with tests/dart_model.py): a white frame -- one component, first pixel
    # (0, 0) -- whose outer border has events only at two dark pixels touching column 0 diagonally, where its natural start does not fire,
    # plus 7 % single dark pixels away from the frame's edge (0.3 darts per pixel: the dense path).  The batch is re-run through the
    # fixpoint passes, on which k_local_contract must not trust the natural assignment.
    anomaly = np.full((2, 96, 128), 255, np.uint8)
    specks = rng.random((2, 96, 128)) < 0.07
    specks[:, :10, :10] = False
    specks[:, :2, :] = False; specks[:, -2:, :] = False; specks[:, :, :2] = False; specks[:, :, -2:] = False
    anomaly[specks] = 0
    anomaly[:, 3, 1] = 0; anomaly[:, 4, 0] = 0
    cases.insert(1, np.repeat(anomaly[..., None], 3, axis=3))
    resolved = []
    for ci, frames in enumerate(cases):
        dd = d4 if ci == len(cases) - 1 else d
        detector = _round5_detector(dicts, name4) if ci == len(cases) - 1 else det
        want_traced, want_markers = 0, []
        for f in range(frames.shape[0]):
            ref = oracle.detect(frames[f], dd.code_list, dd.num_bits, dd._tau)
            cs, _, _ = oracle.find_contours(ref["thresholded"])
            img = ref["thresholded"] > 0
            want_traced += sum(1 for c in cs if not _speck(img, c))
            want_markers.append([(m["id"], m["code"], tuple(v for c in m["corners"] for v in c)) for m in ref["markers"]])
        got = {}
        for taps in (False, True):
            ctx = detector._context()
            ctx.set_debug_taps(taps)
            n, h, w, c = frames.shape
            a = np.ascontiguousarray(frames)
            m, per = ctx.detect_batch(a.ctypes.data, _lib_mod().MEM_HOST, _lib_mod().FMT_RGB8, w, h, w * c, h * w * c, n)
            st = ctx.stats()
            got[taps] = (st["contours_traced"], st["contours_materialised"], [(int(x["id"]), int(x["code"]), tuple(int(v) for v in x["corners"])) for x in m], per.tolist())
            if not taps:
                resolved.append(st["resolve_iterations"])
        ctx.set_debug_taps(False)
        assert got[False][0] == want_traced == got[True][0], (ci, got[False][0], got[True][0], want_traced)
        assert got[True][1] == want_traced                                  # taps: every traced border is materialised
        assert got[False][1] < got[True][1] or want_traced < 50             # the product path prunes
        flat = [t for fr in want_markers for t in fr]
        assert got[False][2] == flat == got[True][2], ci
        assert got[False][3] == got[True][3] == [len(fr) for fr in want_markers]
    assert resolved[1] >= 1 and resolved[0] == 0, resolved      # the anomaly case went through the fixpoint passes, plain noise did not


def _speck(img, c):
    """a 1-point contour of a pixel without any foreground 8-neighbour (the kernels never build darts for those)"""
    if len(c) != 1:
        return False
    x, y = (int(v) for v in c[0])
    h, w = img.shape
    return not any((dx or dy) and 0 <= x + dx < w and 0 <= y + dy < h and img[y + dy, x + dx] for dy in (-1, 0, 1) for dx in (-1, 0, 1))


def _round5_detector(dicts, name="ARUCO"):
    from aruco3_amd.aruco import Detector, DetectorConfig

    return Detector(DetectorConfig(), dicts.new_from_named_dict(name))


def _lib_mod():
    from aruco3_amd import _lib

    return _lib


@pytest.mark.parametrize("gates", [False, True])
def test_python_batch_queue_matches_detect_batch(dicts, gates):
    """aruco3_amd.aruco.BatchQueue (the twin of the Rust shim's): batches of device tensors and of host arrays through a rotation of
    four contexts, free-running and with burst gates: every batch equals Detector.detect_batch; the library reports the stepping"""
    import torch

    from aruco3_amd import synth
    from aruco3_amd.aruco import BatchQueue, Detector, DetectorConfig

    det = Detector(DetectorConfig(), dicts.new_from_named_dict("ARUCO_DEFAULT"))
    batches = [synth.config_frames(1, 5, first=5 * j)[0] for j in range(6)]
    want = [[[(m.id, m.code, tuple(m.corners), m.hamming_distance) for m in d.markers] for d in det.detect_batch(b)] for b in batches]
    assert sum(len(f) for w in want for f in w) > 50
    q = BatchQueue(det, depth=4, gates=gates)
    inputs = [torch.from_numpy(b).cuda() if j % 2 == 0 else b for j, b in enumerate(batches)]     # device and host frames in turn
    seen = []
    for rep in range(3):
        order = list(range(6))
        submitted = []
        for j in order:
            if q.full:
                k = submitted.pop(0)
                got = q.collect()
                seen.append(q.last_stepping)
                assert [[(m.id, m.code, tuple(m.corners), m.hamming_distance) for m in d.markers] for d in got] == want[k], (rep, k)
            q.submit(inputs[j]); submitted.append(j)
        while len(q):
            k = submitted.pop(0)
            got = q.collect()
            seen.append(q.last_stepping)
            assert [[(m.id, m.code, tuple(m.corners), m.hamming_distance) for m in d.markers] for d in got] == want[k], (rep, k)
    q.close()
    if gates:
        assert "held_released_by_last" in seen and "burst_last" in seen
    else:
        assert set(seen) == {"whole"}
