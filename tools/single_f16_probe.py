#!/usr/bin/env python3
"""Experiment: how far is a ONE-product f16 scoring pass (library built with -DRAILS_F16_SINGLE=1, selected through
RAILS_AMD_LIBRARY) from the fp32 path, and how many candidates would a speculate-then-verify top-k need on top of it?
Prints kernel times, the error distribution over the whole corpus and, per query, how many items lie within the error of the
k-th score.   RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_single.so python tools/single_f16_probe.py [--workload amzn-books]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="amzn-books")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--items", type=int, default=0)
    args = ap.parse_args()
    cfg_key, N, _ = bench.WORKLOADS[args.workload]
    N = args.items or N
    cfg = O.CONFIGS[cfg_key]
    dev = torch.device("cuda:0")
    w = O.synthetic_weights(cfg, seed=0)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).to(dev)
    q = O.synthetic_queries(cfg, args.batch).to(dev)
    uid = torch.arange(args.batch, dtype=torch.int64, device=dev) if cfg.uid_embedding_hash_sizes else None
    out = {}
    with torch.inference_mode():
        for pr in ("fp32", "f16x3"):
            mol.precision = None if pr == "fp32" else pr
            eng = mol.engine()
            index = eng.build_index(X)
            qpack, _, _ = eng.query_pack(q, uid)
            o = eng.score_dense(qpack, args.batch, index)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                eng.score_dense(qpack, args.batch, index, out=o)
            e1.record()
            torch.cuda.synchronize()
            out[pr] = o.clone()
            print(f"{pr}: kernel {e0.elapsed_time(e1) / 10:.3f} ms")
        d = (out["f16x3"] - out["fp32"]).abs()
        qs = torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999])
        flat = d.flatten()
        samp = flat[torch.randint(0, flat.numel(), (2_000_000,), device=dev)]
        print("abs error: max %.3e  mean %.3e  quantiles(0.5,0.9,0.99,0.999,0.9999) %s" % (float(d.max()), float(d.mean()),
              [f"{float(v):.2e}" for v in torch.quantile(samp, qs.to(dev))]))
        kk = min(args.k + 1024, N)
        s32, _ = torch.topk(out["fp32"], kk, dim=1)
        top = out["fp32"] >= s32[:, args.k - 1 : args.k]
        print("abs error among the fp32 top-%d: max %.3e" % (args.k, float(d[top].max())))
        gaps = s32[:, args.k - 1 : args.k] - s32
        for eps in (1e-3, 1e-2, 3e-2, 0.1, 0.2, 0.5):
            within = (gaps[:, args.k :] < 2 * eps).sum(1)
            print(f"items ranked below k = {args.k} but within 2*eps = {2 * eps:g} of the k-th fp32 score: max over queries {int(within.max())}, mean {float(within.float().mean()):.1f}")
        print("score range: top-1 %.3f  k-th %.3f  %d-th %.3f  median %.3f" % (float(s32[:, 0].mean()), float(s32[:, args.k - 1].mean()), kk,
              float(s32[:, -1].mean()), float(out["fp32"].median())))


if __name__ == "__main__":
    main()
