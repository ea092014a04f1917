import torch, time
s = torch.cuda.Stream()
for cyc in (1_000_000, 8_400_000):
    torch.cuda.synchronize(); t0=time.perf_counter()
    with torch.cuda.stream(s):
        torch.cuda._sleep(cyc)
    t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
    print(cyc, "enqueue", (t1-t0)*1e3, "ms; total", (t2-t0)*1e3, "ms")
