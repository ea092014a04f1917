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

from tests.util import marker_tuples

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
    return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])


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
