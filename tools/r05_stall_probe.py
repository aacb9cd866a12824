"""Is there a periodic runtime stall (per N launches / per N events / per allocation pattern)?  Times batches of small launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
torch.cuda.synchronize()
spikes = []
t_all = time.perf_counter()
for batch in range(600):
    t0 = time.perf_counter()
    for _ in range(100):
        x.add_(1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 3.0:
        spikes.append((batch * 100, round(dt, 1)))
print("60000 launches in", round(time.perf_counter() - t_all, 2), "s; batches of 100 over 3 ms:", spikes)
# with events recorded per launch (bench's pattern) and pinned-host copies
spikes = []
h = torch.zeros(8).pin_memory()
for batch in range(300):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
    t0 = time.perf_counter()
    for i in range(100):
        evs[i].record()
        x.add_(1.0)
        h.copy_(x[:8], non_blocking=True)
    evs[100].record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    if dt > 6.0:
        spikes.append((batch * 100, round(dt, 1)))
print("with events + D2H copies: batches over 6 ms:", spikes)
