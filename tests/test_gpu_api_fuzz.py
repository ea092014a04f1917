"""Random call sequences through the public stepping API: several contexts -- some on streams of their own, some moved onto shared or
private caller streams mid-way -- submit, collect (in any order), order_after (any pair, including holders and contexts with a deferred
decode stage), synchronous calls in between, pose and plain batches mixed, a context destroyed with a batch in flight.  Whatever the
library decides to do with each batch (held, released early or by a last member, decode deferred, whole), every collected result must
equal the synchronous result for the same frames, no call may fail, and nothing may hang (the test runs under a timeout).  GPU only."""
import numpy as np
import pytest

from tests.util import marker_tuples

pytestmark = pytest.mark.gpu


def _ctx(dicts):
    from aruco3_amd.aruco import Detector, DetectorConfig

    return Detector(DetectorConfig(), dicts.new_from_named_dict("ARUCO_DEFAULT"))._context()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("seed", list(range(1, 13)))
def test_random_stepping_sequences_equal_synchronous_results(dicts, seed):
    import torch

    from aruco3_amd import _lib, synth

    rng = np.random.default_rng(1000 + seed)
    # three batches of different content (and two sizes: a context that changes shape re-plans on the host: no hold for that batch)
    sets = []
    for j, (cfg, cnt) in enumerate(((1, 5), (1, 3), (1, 5))):
        f, _ = synth.config_frames(cfg, cnt, first=7 * j)
        t = torch.from_numpy(f).cuda()
        n, h, w, c = f.shape
        sets.append((t, (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)))
    ref = _ctx(dicts)
    want = [ref.detect_batch(*a) for _, a in sets]
    want_pose = [ref.detect_batch_pose(*a, 40.0, None, 256) for _, a in sets]
    same = lambda got, w: marker_tuples(got[0]) == marker_tuples(w[0]) and np.array_equal(got[1], w[1])
    streams = [torch.cuda.Stream() for _ in range(3)]
    ctxs = [_ctx(dicts) for _ in range(5)]
    pending = {}          # context index -> (batch index, with pose)
    seen = set()
    for step in range(260):
        op = rng.choice(["submit", "submit", "collect", "collect", "order", "order", "sync", "stream", "destroy"], p=[0.24, 0.12, 0.2, 0.12, 0.12, 0.06, 0.06, 0.06, 0.02])
        k = int(rng.integers(len(ctxs)))
        cx = ctxs[k]
        if op == "submit" and k not in pending:
            j = int(rng.integers(len(sets))); pose = bool(rng.integers(4) == 0)
            if pose:
                cx.submit_pose(*sets[j][1], 40.0, None, 256)
            else:
                cx.submit(*sets[j][1], out_cap=256)
            pending[k] = (j, pose)
        elif op == "collect" and pending:
            kk = int(rng.choice(sorted(pending)))            # any context with a batch in flight, not necessarily the oldest
            j, pose = pending.pop(kk)
            if pose:
                m, p, poses = ctxs[kk].collect_pose()
                assert same((m, p), want_pose[j]) and np.array_equal(poses.view(np.uint32), want_pose[j][2].view(np.uint32)), (seed, step, kk, j)
            else:
                assert same(ctxs[kk].collect(), want[j]), (seed, step, kk, j)
            seen.add(ctxs[kk].stats()["stepping"])
        elif op == "order":
            m = int(rng.integers(len(ctxs)))
            if k not in pending:                              # gates are declared before a submit
                cx.order_after(ctxs[m])
        elif op == "sync" and k not in pending:
            j = int(rng.integers(len(sets)))
            assert same(cx.detect_batch(*sets[j][1]), want[j]), (seed, step, k, j)
        elif op == "stream":                                  # move a context (also one that holds a chain: it goes out on the old stream first)
            s = int(rng.integers(len(streams) + 1))
            cx.set_stream(streams[s].cuda_stream if s < len(streams) else 0)
        elif op == "destroy" and len(ctxs) > 3:
            cx.close()                                        # with or without a batch in flight
            pending.pop(k, None)
            ctxs[k] = _ctx(dicts)
    for kk, (j, pose) in sorted(pending.items()):
        got = ctxs[kk].collect_pose() if pose else ctxs[kk].collect()
        assert same(got[:2], (want_pose if pose else want)[j]), (seed, "drain", kk, j)
    torch.cuda.synchronize()
    assert "whole" in seen and len(seen) >= 3, seen           # the sequences do reach the held / deferred paths


@pytest.mark.timeout(600)
def test_random_stepping_from_three_threads(dicts):
    """three host threads, a context each (streams of their own), random submit / collect / synchronous calls, and gates declared on
    the OTHER threads' contexts (which makes the library release their held chains or deferred decode stages from this thread):
    every result equals the synchronous one, nothing deadlocks"""
    import threading

    import torch

    from aruco3_amd import _lib, synth

    sets = []
    for j in range(3):
        f, _ = synth.config_frames(1, 4, first=4 * j)
        t = torch.from_numpy(f).cuda()
        n, h, w, c = f.shape
        sets.append((t, (t.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n)))
    ref = _ctx(dicts)
    want = [ref.detect_batch(*a) for _, a in sets]
    same = lambda got, w: marker_tuples(got[0]) == marker_tuples(w[0]) and np.array_equal(got[1], w[1])
    ctxs = [_ctx(dicts) for _ in range(3)]
    for cx in ctxs:
        for _, a in sets:
            cx.detect_batch(*a)
    errs = []

    def worker(k):
        try:
            rng = np.random.default_rng(77 + k)
            cx = ctxs[k]
            for step in range(400):
                r = rng.random()
                if r < 0.55:
                    for m in range(3):
                        if m != k and rng.random() < 0.5:
                            cx.order_after(ctxs[m])
                    j = int(rng.integers(3))
                    cx.submit(*sets[j][1], out_cap=256)
                    if rng.random() < 0.3:
                        for _ in range(int(rng.integers(200))):
                            pass
                    assert same(cx.collect(), want[j]), (k, step, j)
                else:
                    j = int(rng.integers(3))
                    assert same(cx.detect_batch(*sets[j][1]), want[j]), (k, step, j)
        except Exception as e:   # noqa: BLE001
            errs.append((k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=500)
    assert not any(t.is_alive() for t in ts), "a thread hangs"
    assert not errs, errs
