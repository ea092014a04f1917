"""Hardware-queue-aware stream choice for callers that keep several contexts in flight beside other stream work (a collective).

The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (include/aruco3_hip.h, "Hardware queues"),
and two streams that share a queue run IN ORDER whatever their events say -- a stream that merely WAITS (a collective waiting for
its peers, a stream waiting for an event) holds up every other stream of its queue.  Which streams share a queue is not something
the runtime tells; it can be measured: put something slow on one stream and see which other streams' work is held up behind it.

Measured on MI355X / ROCm 7.2 with GPU_MAX_HW_QUEUES=8 (profiles/r05_queue_collisions.txt): PyTorch creates its pool of 32 streams
per priority at once, the library's own context streams come later, and the context stream of a burst's LAST member landed on
the queue of the bench's side stream -- every collective then stalled that context for its whole duration.

`pick_streams` chooses, among candidate streams, `n` that are held up neither by the caller's blockers nor by one another; the
contexts are then moved onto them with a3_set_stream (contexts on DISTINCT caller streams are stepped exactly like contexts on
streams of their own: burst gates, held chains).
"""
from typing import Callable, List, Sequence

import torch


def _delays_ms(blocker: Callable[[], None], streams: Sequence["torch.cuda.Stream"], device) -> List[float]:
    """how long a trivial kernel enqueued on each stream AFTER `blocker()` has been enqueued takes to complete"""
    scratch = torch.zeros(64, dtype=torch.int32, device=device)
    torch.cuda.synchronize(device)
    starts = [torch.cuda.Event(enable_timing=True) for _ in streams]
    ends = [torch.cuda.Event(enable_timing=True) for _ in streams]
    for s, e in zip(streams, starts):
        e.record(s)
    blocker()
    for s, e in zip(streams, ends):
        with torch.cuda.stream(s):
            scratch.add_(1)
        e.record(s)
    torch.cuda.synchronize(device)
    return [a.elapsed_time(b) for a, b in zip(starts, ends)]


def sleep_on(stream: "torch.cuda.Stream", ms: float = 1.5) -> Callable[[], None]:
    """a blocker: a kernel that stays resident for about `ms` milliseconds on `stream`"""
    def go():
        with torch.cuda.stream(stream):
            torch.cuda._sleep(int(ms * 2.4e6))
    return go


def held_up_by(blocker: Callable[[], None], streams: Sequence["torch.cuda.Stream"], device, threshold_ms: float = 0.5, repeats: int = 2) -> List[bool]:
    """per stream: is work enqueued on it behind `blocker` held up (in every one of `repeats` trials)?"""
    held = [True] * len(streams)
    for _ in range(repeats):
        d = _delays_ms(blocker, streams, device)
        held = [h and x > threshold_ms for h, x in zip(held, d)]
    return held


def pick_streams(n: int, candidates: Sequence["torch.cuda.Stream"], blockers: Sequence[Callable[[], None]], device, blocker_ms: float = 1.5):
    """-> (chosen streams, report).  Chooses `n` of `candidates` such that none is held up by any of `blockers` (callables that enqueue
    something slow -- `sleep_on(side)`, or a sleep followed by a collective) nor by a sleep on another chosen stream.  Falls back
    to the first candidates when fewer than `n` free ones exist (the report says so)."""
    cands = list(candidates)
    blocked = [False] * len(cands)
    for b in blockers:
        blocked = [x or y for x, y in zip(blocked, held_up_by(b, cands, device, threshold_ms=blocker_ms / 3))]
    free = [i for i, x in enumerate(blocked) if not x]
    chosen: List[int] = []
    for i in free:
        if len(chosen) == n:
            break
        if chosen:
            # does a sleep on candidate i hold up a stream already chosen (= same hardware queue)?
            held = held_up_by(sleep_on(cands[i], blocker_ms), [cands[j] for j in chosen], device, threshold_ms=blocker_ms / 3)
            if any(held):
                continue
        chosen.append(i)
    report = {"candidates": len(cands), "held_up_by_blockers": [i for i, x in enumerate(blocked) if x], "chosen": list(chosen),
              "complete": len(chosen) == n}
    if len(chosen) < n:
        chosen = (chosen + [i for i in range(len(cands)) if i not in chosen])[:n]
        report["chosen"] = list(chosen)
    return [cands[i] for i in chosen], report
