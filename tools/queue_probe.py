"""Which streams of a bench-like process share a hardware queue?  (profiles/r05_queue_collisions.txt)

  GPU_MAX_HW_QUEUES=8 python tools/queue_probe.py [nccl]

Creates what bench.py creates -- (optionally) a one-rank NCCL process group, torch's explicit stream and side stream, four contexts
on streams of their own, twelve more torch pool streams -- then, for every stream S, puts a 1.5 ms sleep on S and measures how long
a trivial kernel on each other stream takes: a stream that is held up shares S's hardware queue (streams of one queue run in order)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import torch.distributed as dist

from aruco3_amd import streams as a3s
from aruco3_amd.aruco import Detector, DetectorConfig
from aruco3_amd.dictionaries import ARDictionary

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
with_nccl = len(sys.argv) > 1 and sys.argv[1] == "nccl"
if with_nccl:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
d = ARDictionary.new_from_named_dict("ARUCO")
ctxs = [Detector(DetectorConfig.default(), d)._context() for _ in range(4)]
stream = torch.cuda.Stream(device=dev)
own = [torch.cuda.ExternalStream(cx.stream_ptr, device=dev) for cx in ctxs]
side = torch.cuda.Stream(device=dev)
pool = [torch.cuda.Stream(device=dev) for _ in range(12)]
names = ["explicit"] + [f"ctx{k}" for k in range(4)] + ["side"] + [f"pool{k}" for k in range(12)]
streams = [stream] + own + [side] + pool
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}  nccl={'yes' if with_nccl else 'no'}  streams probed: {len(streams)}")
groups = {}
for i, s in enumerate(streams):
    held = a3s.held_up_by(a3s.sleep_on(s), streams, dev, threshold_ms=0.5)
    key = tuple(j for j, h in enumerate(held) if h)
    groups.setdefault(key, []).append(i)
    print(f"sleep on {names[i]:9s} holds up: {[names[j] for j in key if j != i]}")
if with_nccl:
    t_in = torch.zeros(64, dtype=torch.uint8, device=dev); t_out = torch.zeros(64, dtype=torch.uint8, device=dev)

    def coll():
        with torch.cuda.stream(side):
            torch.cuda._sleep(int(1.5 * 2.4e6))
            dist.all_gather_into_tensor(t_out, t_in)

    held = a3s.held_up_by(coll, streams, dev, threshold_ms=0.5)
    print(f"sleep + all_gather issued from side holds up: {[names[j] for j, h in enumerate(held) if h]}   (side's queue + the backend's internal stream's)")
    chosen, rep = a3s.pick_streams(4, own + pool, [coll], dev)
    print("pick_streams(4, own + pool, [collective behind a sleep]) ->", rep)
print("queues seen (sets of streams that hold one another up):", len(groups))
if with_nccl:
    dist.destroy_process_group()
