#!/usr/bin/env python3
"""Stress fuzz of the f16 scoring kernels against the fp32 ones: random per-layer scalings of the gate weights and biases (far
from the reference's initialisation), several shapes; reports non-finite logits and the largest deviation.  Found the
near-overflow bug of the un-shifted softmax.   python tools/f16_stress_fuzz.py [--cases 40]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)
from tests.test_gpu_parity import build_module  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(123)
    worst = {}
    for case in range(args.cases):
        name = ("amzn-books", "ml-20m", "ml-1m", "synthetic-16x16x64")[case % 4]
        cfg = O.CONFIGS[name]
        w = dict(O.synthetic_weights(cfg, seed=100 + case))
        scales = {}
        for key in w:
            if "_gating_fn" in key and torch.is_floating_point(w[key]):
                sc = float(10 ** (torch.rand(1, generator=g) * 1.5 - 0.25)) if torch.rand(1, generator=g) < 0.6 else 1.0   # 0.56 .. 17.8
                if key.endswith("bias"):
                    w[key] = w[key] + torch.randn(w[key].shape, generator=g) * 0.3 * sc
                else:
                    w[key] = w[key] * sc
                scales[key.split("_gating_fn.")[1]] = round(sc, 2)
        N, B = 20_000, 24
        X = torch.from_numpy(O.hash_item_table(50 + case, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
        ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
        q = O.synthetic_queries(cfg, B, seed=200 + case).to(dev)
        kw = {"user_ids": torch.arange(B, dtype=torch.int64, device=dev)} if cfg.uid_embedding_hash_sizes else {}
        with torch.inference_mode():
            ref = rails_amd.MoLBruteForceTopK(build_module(cfg, w, dev, None), X, ids).all_logits(q, **kw)
            line = f"case {case:2d} {name:18s} fp32 finite {bool(torch.isfinite(ref).all())} |max| {float(ref.abs().max()):6.2f}"
            for pr in ("f16x3", "f16x1"):
                try:
                    got = rails_amd.MoLBruteForceTopK(build_module(cfg, w, dev, pr), X, ids).all_logits(q, **kw)
                except NotImplementedError as e:
                    line += f"  {pr}: refused ({str(e)[:40]})"
                    continue
                bad = int((~torch.isfinite(got)).sum())
                d = float((got - ref)[torch.isfinite(got)].abs().max())
                line += f"  {pr}: nonfinite {bad} max|d| {d:.2e}"
                worst[pr] = max(worst.get(pr, 0.0), float("inf") if bad else d)
            print(line, scales if bad or case < 3 else "")
    print("worst deviation:", worst)


if __name__ == "__main__":
    main()
