#!/usr/bin/env python3
"""rails_topk vs rails_topk_filtered on small rows (what the seen-id filter inside the selection launch costs):
  python tools/filter_fusion_probe.py    -> microseconds per call, events around 200 back-to-back calls"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rails_amd import engine as E

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for n, kp, k in ((3883, 200, 120), (27278, 200, 120), (86971, 200, 120)):
    scores = torch.randn((32, n), generator=g).to(dev)
    ids = torch.arange(1, n + 1, dtype=torch.int64, device=dev)
    ws = torch.empty(E._lib.load().rails_topk_workspace_bytes(32, n, kp), dtype=torch.uint8, device=dev)
    row = []
    for width in (0, 61, 211):
        inv = ids[torch.randint(0, n, (32, max(width, 1)), generator=g).to(dev)] if width else None
        fn = (lambda: E.topk(scores, kp, ids=ids, workspace=ws)) if inv is None else (lambda: E.topk_filtered(scores, kp, ids, inv, k, workspace=ws))
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            fn()
        e1.record()
        torch.cuda.synchronize()
        row.append(f"width {width:3d}: {e0.elapsed_time(e1) / 200 * 1e3:6.1f} us")
    print(f"n = {n:6d}, k' = {kp}: " + "   ".join(row))
