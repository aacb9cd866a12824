import torch, time
dev = torch.device("cuda", 0)
x = torch.empty(1_000_000_000, dtype=torch.int32, device=dev).random_(0, 100)   # 4 GB
y = torch.empty_like(x)
def t(fn, nbytes, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    return nbytes / ms / 1e9
print("sum int32 (read 4 GB): %.2f TB/s" % t(lambda: x.sum(), x.numel() * 4))
print("max int32 (read 4 GB): %.2f TB/s" % t(lambda: x.max(), x.numel() * 4))
xf = x.view(torch.float32)
print("sum f32   (read 4 GB): %.2f TB/s" % t(lambda: xf.sum(), x.numel() * 4))
print("copy (read 4 + write 4 GB): %.2f TB/s" % t(lambda: y.copy_(x), x.numel() * 8))
print("fill (write 4 GB): %.2f TB/s" % t(lambda: y.fill_(1), x.numel() * 4))
