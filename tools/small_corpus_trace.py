#!/usr/bin/env python3
"""The get_top_k_outputs step on a small corpus (default ML-1M, 3 883 items), for `rocprofv3 --kernel-trace`: which launches the
78 us step consists of and what separates them.   python tools/small_corpus_trace.py [--workload ml-20m] [--precision f16x3]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="ml-1m")
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--steps", type=int, default=200)
    args = ap.parse_args()
    cfg_key, N, width = bench.WORKLOADS[args.workload]
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
    mol.precision = None if args.precision == "fp32" else args.precision
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
    ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, 32).to(dev)
    kw = {"user_ids": torch.arange(32, device=dev)} if cfg.uid_embedding_hash_sizes else {}
    inv = ids[0, torch.randint(0, N, (32, max(width, 1)), device=dev)]
    with torch.inference_mode():
        tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
        cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
        for _ in range(10):
            cand.get_top_k_outputs(q, 120, kw, tk, inv)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cand.get_top_k_outputs(q, 120, kw, tk, inv)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"{args.workload} {args.precision}: host enqueue {1e6 * (t1 - t0) / args.steps:.1f} us/step; step {1e6 * (t2 - t0) / args.steps:.1f} us")


if __name__ == "__main__":
    main()
