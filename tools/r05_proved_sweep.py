#!/usr/bin/env python3
"""The proved exact path against the dense fp32 kernels over many batches: weight seeds x table seeds x query seeds on the amzn-books shape at
full N (and, with --workload synthetic-16x16x64 --items N, the per-pair form): per combination the number of calls proved / redone and whether
every output (scores, ids, order) was bit-identical to the dense module's.
  python tools/r05_proved_sweep.py [--workload amzn-books] [--items 0] [--weights 3] [--tables 2] [--batches 10] [--out f.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generators only)
from rails_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="amzn-books")
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--weights", type=int, default=3)
    ap.add_argument("--tables", type=int, default=2)
    ap.add_argument("--batches", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--k", type=int, default=200)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    cfg_key, N, _ = bench.WORKLOADS[args.workload]
    N = args.items or N
    cfg = O.CONFIGS[cfg_key]
    dev = torch.device("cuda:0")
    rows = []
    total = {"calls": 0, "proved": 0, "redone": 0, "identical": 0, "violations": 0}
    for ws in range(args.weights):
        w = O.synthetic_weights(cfg, seed=ws)
        mol, _ = rails_amd.create_mol_interaction_module(
            cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
            cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
            cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
            query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
        mol.load_state_dict(w, strict=True)
        mol = mol.to(dev).eval()
        for ts in range(1, args.tables + 1):
            X = E.hash_item_table(ts, 0, N, cfg.item_embedding_dim, dev).unsqueeze(0)
            ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
            with torch.inference_mode():
                proved = bench.brute_force_module(mol, X, ids, "proved")
                dense = bench.brute_force_module(mol, X, ids, "fp32")
                same = 0
                for qs in range(args.batches):
                    q = O.synthetic_queries(cfg, args.batch, seed=100 * ws + 10 * ts + qs).to(dev)
                    kw = {}
                    if cfg.uid_embedding_hash_sizes:
                        kw["user_ids"] = torch.randint(0, cfg.uid_embedding_hash_sizes[0], (args.batch,), generator=torch.Generator().manual_seed(qs), dtype=torch.int64).to(dev)
                    s, i = proved(q, k=args.k, **kw)
                    r_s, r_i = dense(q, k=args.k, **kw)
                    same += int(torch.equal(s, r_s) and torch.equal(i, r_i))
                st = proved.stats()
            row = {"weights_seed": ws, "table_seed": ts, "calls": st["calls"], "proved_calls": st.get("proved_calls", 0), "fallbacks": st["fallbacks"],
                   "bound_violations": st.get("bound_violations", 0), "identical_outputs": same, "batches": args.batches, "kc": st.get("kc"),
                   "eps_rigorous": st.get("eps_rigorous"), "bound": st.get("bound_kind", "one a-priori eps"), "guard_max": st.get("guard_max")}
            print(row, flush=True)
            rows.append(row)
            total["calls"] += st["calls"]; total["proved"] += st.get("proved_calls", 0); total["redone"] += st["fallbacks"]
            total["identical"] += same; total["violations"] += st.get("bound_violations", 0)
            del proved, dense, X, ids
            torch.cuda.empty_cache()
    out = {"workload": args.workload, "items": N, "batch": args.batch, "k": args.k, "rows": rows, "total": total,
           "all_identical": total["identical"] == args.weights * args.tables * args.batches}
    print(json.dumps(out["total"]), out["all_identical"])
    if args.out:
        json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
