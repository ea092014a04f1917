#!/usr/bin/env python3
"""A/B of stepping arrangements inside ONE process (boxes of the pool differ by more than the effects): BASELINE config 2,
contexts in submit / collect.  An arrangement = (streams: shared | own, contexts, K1 waves per SIMD, overlap mode).  Arrangements
are interleaved ROUNDS times; prints the median ms per step of each, the threshold kernel's duration in company (sampled events)
and its duration alone in the same geometry.
  python tools/ab_streams.py [frames] [steps] [rounds] [spec,spec,...]      spec = streams:contexts:k1waves:overlap[:group[:hold]]  (overlap -1: the library decides per batch, its default; 0/1/2: forced through the internal switch)  e.g. own:2:1:0, own:8:2:0:4 (two bursts of
  four, submitted together), own:4:2:0:-1 (rotation with burst gates), own:4:2:0:-1:0 (the same, chains not held back);
  further fields KEY=VALUE are environment knobs of a -DA3_TUNING build (A3_HIP_LIB), set for that arrangement only"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import os
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

DEFAULT = "shared:2:2:2,own:2:2:0,own:2:1:0,own:3:1:0,own:3:2:0,shared:2:1:2"


def main():
    import torch

    from aruco3_amd import _lib, synth
    from aruco3_amd.aruco import Detector, DetectorConfig
    from aruco3_amd.dictionaries import ARDictionary

    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    specs = (sys.argv[4] if len(sys.argv) > 4 else DEFAULT).split(",")
    d = ARDictionary.new_from_named_dict("ARUCO")
    spec, _ = synth.config_spec(2)
    max_ctx = max(int(s.split(":")[1]) for s in specs)
    # every context steps a batch of its OWN (round 5: four contexts re-reading ONE batch share it in the Infinity Cache, which flattered
    # every "own" arrangement of round 4); `same=1` in a spec restores the shared batch for that arrangement
    bufs = [synth.render_frames_device(spec, d.code_list, d.num_bits, [synth.frame_seed(2, j * frames + i) for i in range(frames)])[0] for j in range(max_ctx)]
    n, h, w, c = bufs[0].shape
    args_of = [(b.data_ptr(), _lib.MEM_DEVICE, _lib.FMT_RGB8, w, h, w * c, h * w * c, n) for b in bufs]
    a = args_of[0]
    same_batch = [False]
    arg = lambda k: args_of[0] if same_batch[0] else args_of[k % len(args_of)]
    L = _lib.load()
    if os.environ.get("A3_K1_STREAM", "0") != "0":     # 1 / 2: the threshold kernels on one device-wide stream of the lowest / highest priority
        assert L.a3_debug_set_k1_stream(int(os.environ["A3_K1_STREAM"])) == 0
    if os.environ.get("A3_PARTITION", "0") != "0":     # "k1_cus[:pattern]": CU partition, set before any context exists
        pp = os.environ["A3_PARTITION"].split(":")
        assert L.a3_debug_set_partition(int(pp[0]), int(pp[1]) if len(pp) > 1 else 0) == 0
    stream = torch.cuda.Stream()
    pools = {"shared": [], "own": []}
    for kind in pools:
        if not any(s.startswith(kind) for s in specs):
            continue
        for _ in range(max_ctx):
            cx = Detector(DetectorConfig.default(), d)._context()
            if kind == "shared":
                cx.set_stream(stream.cuda_stream)
            for j in range(len(args_of)):      # (every batch once: the pools grow to the largest before anything is timed)
                cx.detect_batch(*args_of[j], out_cap=n * 64)
            for _ in range(2):
                cx.detect_batch(*arg(len(pools[kind])), out_cap=n * 64)
            pools[kind].append(cx)
    refs = [pools[next(k for k in pools if pools[k])][0].detect_batch(*args_of[j], out_cap=n * 64) for j in range(len(args_of))]

    def run(k, ctxs, group=1):
        nc = len(ctxs)
        host_us.clear(); host_us.extend([] for _ in range(nc))
        if group <= 1:       # rotation: the context just collected submits the batch nc ahead
            def sub(i):      # group < 0: gated rotation (a3_order_after): context k waits for contexts k+1 .. nc-1
                if group < 0:   # -1: every later context; -2: all but the last; -3: only the next one
                    k0 = i % nc
                    later = list(range(k0 + 1, nc))
                    if group == -2: later = later[:-1]
                    if group == -3: later = later[:1]
                    for mth in later:
                        ctxs[k0].order_after(ctxs[mth])
                t_s = time.perf_counter()
                ctxs[i % nc].submit(*arg(i % nc), out_cap=n * 64)
                host_us[i % nc].append((time.perf_counter() - t_s) * 1e6)
            for i in range(min(nc, k)):
                sub(i)
            for i in range(k):
                m, per = ctxs[i % nc].collect()
                if i + nc < k:
                    sub(i)
            return m, per, (k - 1) % nc
        # bursts: the contexts form nc / group groups; a group's batches are submitted together (their threshold kernels back to
        # back), collected together, and submitted again
        groups = [ctxs[g * group:(g + 1) * group] for g in range(nc // group)]
        left = k
        pending = []
        for grp in groups:
            if left >= group:
                for cx in grp: cx.submit(*arg(ctxs.index(cx)), out_cap=n * 64)
                pending.append(grp); left -= group
        while pending:
            grp = pending.pop(0)
            for cx in grp: m, per = cx.collect()
            last = ctxs.index(grp[-1])
            if left >= group:
                for cx in grp: cx.submit(*arg(ctxs.index(cx)), out_cap=n * 64)
                pending.append(grp); left -= group
        return m, per, last

    host_us = []
    seen = {}
    knobs_seen = set()
    res = {s: [] for s in specs}
    host = {}
    k1c = {s: [0.0, 0] for s in specs}
    k1a = {}
    for r in range(rounds):
        for s in specs:
            kind, nc, kw, ov = s.split(":")[:4]
            group = int(s.split(":")[4]) if len(s.split(":")) > 4 else 1
            assert L.a3_debug_set_hold(int(s.split(":")[5]) if len(s.split(":")) > 5 else 1) == 0
            for kv in knobs_seen:                      # tuning builds read their knobs from the environment at every launch
                os.environ.pop(kv, None)
            same_batch[0] = False
            for kv in s.split(":")[6:]:
                kname, kval = kv.split("=")
                if kname == "same":
                    same_batch[0] = kval == "1"
                    continue
                os.environ[kname] = kval; knobs_seen.add(kname)
            ctxs = pools[kind][: int(nc)]
            assert L.a3_debug_set_overlap(int(ov)) == 0
            assert L.a3_debug_set_k1_waves(int(kw)) == 0
            if s not in k1a:   # the threshold kernel alone in this geometry (synchronous calls, nothing else on the GPU)
                cx = ctxs[0]
                cx.set_profiling(_lib.PROFILE_THRESHOLD_ONLY)
                cx.profile(_lib.STAGE_THRESHOLD, reset=True)
                for _ in range(8):
                    cx.detect_batch(*arg(0), out_cap=n * 64)
                ms, cnt = cx.profile(_lib.STAGE_THRESHOLD, reset=True)
                k1a[s] = ms / max(cnt, 1)
            for cx in ctxs:
                cx.set_profiling(_lib.PROFILE_THRESHOLD_SAMPLED)
                cx.profile(_lib.STAGE_THRESHOLD, reset=True)
            run(8 * max(group, 1), ctxs, group)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            m, per, last = run(steps, ctxs, group)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            ref = refs[0 if same_batch[0] else last % len(refs)]
            assert len(m) == len(ref[0]) and np.array_equal(per, ref[1])
            seen[s] = [cx.stats()["stepping"] for cx in ctxs]
            done_steps = steps if group <= 1 else (steps // group) * group
            res[s].append(dt / done_steps * 1e3)
            host[s] = [round(sum(v) / max(len(v), 1)) for v in host_us]
            for cx in ctxs:
                ms, cnt = cx.profile(_lib.STAGE_THRESHOLD, reset=True)
                k1c[s][0] += ms; k1c[s][1] += cnt
                cx.set_profiling(0)
    for s in specs:
        v = sorted(res[s])
        med = v[len(v) // 2]
        print(f"{s:16s} median {med:.4f} ms/step  ({n / med * 1e3:8.0f} frames/s)  K1 in company {k1c[s][0] / max(k1c[s][1], 1):.4f} ms, alone {k1a[s]:.4f} ms  "
              f"all {[round(x, 4) for x in res[s]]}  host us per submit by context {host.get(s)}  library stepping {seen.get(s)}", flush=True)
    L.a3_debug_set_overlap(-1)
    L.a3_debug_set_k1_waves(2)
    L.a3_debug_set_hold(1)


if __name__ == "__main__":
    main()
