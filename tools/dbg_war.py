"""does a sleep on a side stream delay a copy behind it while another stream overwrites the source?  (diagnosis of the gather test)"""
import time, torch
dev = torch.device("cuda", 0)
a = torch.cuda.Stream(); side = torch.cuda.Stream()
buf = torch.zeros(1 << 20, dtype=torch.uint8, device=dev); out = torch.empty_like(buf)
torch.cuda.synchronize()
for trial in range(3):
    with torch.cuda.stream(a):
        buf.fill_(1)
        ev = torch.cuda.Event(); ev.record(a)
    side.wait_event(ev)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        torch.cuda._sleep(8_400_000)
        out.copy_(buf, non_blocking=True)
    t1 = time.perf_counter()
    with torch.cuda.stream(a):
        buf.fill_(2)          # the overwrite, no guard
        e2 = torch.cuda.Event(enable_timing=False); e2.record(a)
    e2.synchronize(); t2 = time.perf_counter()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"trial {trial}: enqueue {1e3*(t1-t0):.3f} ms, overwrite done at {1e3*(t2-t0):.3f} ms, all done {1e3*(t3-t0):.3f} ms, copy saw {int(out[0])} (1 = old, 2 = overwritten)")
