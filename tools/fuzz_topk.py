import sys, random, torch
sys.path.insert(0, "/root/repo")
from rails_amd import engine as E
from oracle import mol_oracle as O
dev = torch.device("cuda", 0)
random.seed(7); g = torch.Generator().manual_seed(7)
bad = 0
cases = []
for _ in range(160):
    regime = random.choice(["small", "mid", "two", "radix", "ties", "bf16"])
    rows = random.choice([1, 2, 3, 5, 8, 17, 32])
    if regime == "small": n = random.randint(1, 1100)
    elif regime == "mid": n = random.randint(1025, 49152)
    elif regime == "two": n = random.randint(49153, 600000)
    elif regime == "radix": n = random.randint(49153, 300000)
    else: n = random.randint(2000, 200000)
    kmax = min(n, 4096 if regime != "radix" else 6000)
    k = random.choice([1, 2, 7, 64, 200, 512, 513, 1000, kmax]) 
    k = max(1, min(k, n, 16384))
    if regime == "radix": k = max(513, min(k, n))
    if rows * n > 40_000_000: rows = 1
    x = torch.randn((rows, n), generator=g)
    if regime == "ties": x = torch.round(x * 2) / 2
    if regime == "bf16": x = x.bfloat16().float()
    if random.random() < 0.2: x[:, ::5] = float("-inf")
    ld_pad = random.choice([0, 0, 1, 3, 4])
    buf = torch.zeros((rows, n + ld_pad)); buf[:, :n] = x
    view = buf.to(dev)[:, :n]
    s, i = E.topk(view, k)
    rs, ri = O.select_topk_deterministic(x, k)
    ok = torch.equal(s.cpu(), rs) and torch.equal(i.cpu(), ri)
    if not ok:
        bad += 1
        print("MISMATCH", regime, rows, n, k, ld_pad)
print("fuzz done, mismatches:", bad)
