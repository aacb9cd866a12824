#!/usr/bin/env python3
"""Time rails_topk alone on a (rows, n) fp32 logit-like matrix: median of repeated calls."""
import argparse, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rails_amd import engine as E

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=32)
ap.add_argument("--n", type=int, default=695762)
ap.add_argument("--k", type=int, default=200)
ap.add_argument("--rounds", type=int, default=20)
ap.add_argument("--dist", default="normal", choices=["normal", "narrow"], help="narrow: 0.1 + 0.02 * randn, the spread of MoL logits (few distinct exponents)")
a = ap.parse_args()
g = torch.Generator().manual_seed(0)
x = torch.randn((a.rows, a.n), generator=g)
x = (x * 2.0 if a.dist == "normal" else 0.1 + 0.02 * x).cuda()
for _ in range(3):
    E.topk(x, a.k)
ts = []
for _ in range(a.rounds):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); E.topk(x, a.k); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3)
print(f"topk rows={a.rows} n={a.n} k={a.k} dist={a.dist}: median {statistics.median(ts):.1f} us  min {min(ts):.1f} us")
