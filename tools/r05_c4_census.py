#!/usr/bin/env python3
"""How many first-pass candidates would a PER-PAIR a-priori bound need on a BASELINE config-4 shard (16x16x64, 12.5 M items)?
Dense fp32 logits of a few queries over the whole shard (the product kernels), the T best items per query, their cross logits on the
CPU (oracle stage functions), eps(c) = f16x3_bound.first_pass_bound(cl_max = c) with c = max_l |cl_l| of the pair, and the count of
items with s + eps(pair) >= the k'-th score -- against the same count with the one a-priori eps of the shape.
  python tools/r05_c4_census.py [--workload synthetic-16x16x64] [--items 12500000] [--queries 8] [--top 60000] [--k-prime 200]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (stage functions for the cross logits of the sampled pairs)
from rails_amd import engine as E  # noqa: E402
from rails_amd import f16x3_bound as FB  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="synthetic-16x16x64")
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--queries", type=int, default=8)
    ap.add_argument("--top", type=int, default=60000)
    ap.add_argument("--k-prime", type=int, default=200)
    ap.add_argument("--out", default="")
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
    X = E.hash_item_table(1, 0, N, cfg.item_embedding_dim, dev).unsqueeze(0)
    ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    B = args.queries
    q_cpu = O.synthetic_queries(cfg, B)
    kw, uid = {}, None
    if cfg.uid_embedding_hash_sizes:
        uid = torch.randint(0, cfg.uid_embedding_hash_sizes[0], (B,), generator=torch.Generator().manual_seed(3), dtype=torch.int64)
        kw["user_ids"] = uid.to(dev)
    tk = bench.brute_force_module(mol, X, ids, "fp32")
    with torch.inference_mode():
        logits = tk.all_logits(q_cpu.to(dev), **kw)
        T = min(args.top, N)
        top_s, top_i = torch.topk(logits, T, dim=1)
    top_s, top_i = top_s.cpu().double().numpy(), top_i.cpu().numpy()
    ks = sorted(k for k in w if "_qi_partial_module" in k)
    W1, b1, W2, b2 = (torch.as_tensor(w[k]) for k in (ks[1], ks[0], ks[3], ks[2]))
    grid = np.linspace(0.0, 1.0 / cfg.temperature * 1.001, 65)
    eps_grid = np.array([FB.first_pass_bound(W1, b1, W2, b2, cfg.temperature, cfg.dot_product_dimension, cfg.query_dot_product_groups,
                                             cfg.item_dot_product_groups, cl_max=float(c))["eps"] for c in grid])
    eps_all = float(eps_grid[-1])
    eq = O.query_component_embeddings(cfg, w, q_cpu, uid)                 # (B, P_Q, d)
    rows = []
    for b in range(B):
        cmax = np.empty(T)
        for lo in range(0, T, 20000):
            items = torch.from_numpy(O.hash_item_rows(1, top_i[b, lo:lo + 20000].astype(np.int64), cfg.item_embedding_dim))
            ex = O.item_component_embeddings(cfg, w, items)                # (n, P_X, d)
            cl = torch.einsum("pd,nmd->npm", eq[b], ex) / cfg.temperature
            cmax[lo:lo + len(items)] = cl.abs().amax((1, 2)).double().numpy()
        j = np.minimum(np.searchsorted(grid, cmax, side="left"), len(grid) - 1)     # eps at the grid point above c: non-decreasing bound
        eps_pair = eps_grid[j]
        e_k = top_s[b, args.k_prime - 1]
        need_pair = int((top_s[b] + eps_pair >= e_k).sum())
        need_all = int((top_s[b] + eps_all >= e_k).sum())
        last_ok = bool(top_s[b, -1] + eps_pair.max() < e_k)
        rows.append({"query": b, "kth_score": float(e_k), "candidates_per_pair_bound": need_pair, "candidates_one_eps": need_all, "one_eps_saturated": need_all >= T,
                     "c_max_median_of_top": float(np.median(cmax)), "c_max_p99": float(np.quantile(cmax, 0.99)), "eps_pair_median": float(np.median(eps_pair)),
                     "eps_pair_max": float(eps_pair.max()), "top_T_covers_per_pair": last_ok, "score_at_T": float(top_s[b, -1])})
        print(rows[-1], flush=True)
    out = {"workload": args.workload, "items": N, "k_prime": args.k_prime, "top_T": T, "eps_a_priori": eps_all, "eps_of_c": {f"{c:.2f}": float(e) for c, e in zip(grid[::8], eps_grid[::8])},
           "rows": rows, "max_candidates_per_pair_bound": max(r["candidates_per_pair_bound"] for r in rows)}
    print(json.dumps(out))
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
