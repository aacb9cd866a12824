#!/usr/bin/env python3
"""Step time of MoLBruteForceTopK in precisions fp32 / proved / f16x3 / f16x3-exact through get_top_k_outputs on amzn-books (B = 32, k = 120,
k' = 200), for `rocprofv3 --kernel-trace --stats` (per-kernel times of the exact path's extra launches) or on its own (wall time).
  python tools/exact_step_profile.py [--precisions f16x3-exact] [--steps 50]
"""
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


def _stats(tk):
    """the module's counters brought up to date with the device verdicts (stats() synchronises), without the bound's long breakdown"""
    st = tk.stats() if hasattr(tk, "stats") else getattr(tk, "rescore_stats", {})
    keep = ("calls", "proved_calls", "fallbacks", "bound_violations", "kc", "eps", "guard_max", "mismatches")
    return {k: st[k] for k in keep if k in st}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precisions", default="fp32,f16x3,f16x3-exact,f16-exact")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--k", type=int, default=120)
    ap.add_argument("--width", type=int, default=80)
    ap.add_argument("--k-prime", type=int, default=200, help="truncate_k_prime_to (the harness's timing protocol: 200)")
    ap.add_argument("--workload", default="amzn-books")
    ap.add_argument("--items", type=int, default=0, help="corpus size (default: the workload's)")
    ap.add_argument("--min-items", type=int, default=-1, help="override MoLBruteForceTopK.SPECULATE_MIN_ITEMS (where the proved flow starts)")
    ap.add_argument("--force-per-pair", action="store_true", help="proved rows: per-pair upper bounds even where one eps proves the calls (PROVED_MAX_EPS = 0)")
    ap.add_argument("--per-pair-pad", type=int, default=0, help="candidate floor beyond k under per-pair bounds (default: the module's 1848)")
    ap.add_argument("--min-batch", type=int, default=0, help="speculate from this batch size on whatever the corpus (PROVED_MIN_BATCH = it, PROVED_MIN_PAIRS = 0)")
    ap.add_argument("--no-rows-copy", action="store_true", help="re-score the candidates from the tile-packed fp32 index (no row-major copy)")
    args = ap.parse_args()
    if args.min_batch:
        rails_amd.MoLBruteForceTopK.PROVED_MIN_BATCH = args.min_batch
        rails_amd.MoLBruteForceTopK.PROVED_MIN_PAIRS = 0
    if args.no_rows_copy:
        rails_amd.MoLBruteForceTopK.ROWS_COPY_MAX_BYTES = 0
    if args.force_per_pair:
        rails_amd.MoLBruteForceTopK.PROVED_MAX_EPS = 0.0
    if args.per_pair_pad:
        rails_amd.MoLBruteForceTopK.PAD_PER_PAIR = (args.per_pair_pad, 1)
    cfg_key, N, _ = bench.WORKLOADS[args.workload]
    N = args.items or N
    if args.min_items >= 0:
        rails_amd.MoLBruteForceTopK.SPECULATE_MIN_ITEMS = args.min_items
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
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
    ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, args.batch).to(dev)
    inv = ids[0, torch.randint(0, N, (args.batch, args.width), device=dev)]
    cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
    kw = {}
    if cfg.uid_embedding_hash_sizes:
        kw["user_ids"] = torch.randint(0, cfg.uid_embedding_hash_sizes[0], (args.batch,), dtype=torch.int64).to(dev)
    with torch.inference_mode():
        for pr in args.precisions.split(","):
            tk = bench.brute_force_module(mol, X, ids, pr)      # "fp32" = dense fp32 kernels, "proved" = the default exact path
            for _ in range(5):
                cand.get_top_k_outputs(q, args.k, kw, tk, inv, truncate_k_prime_to=args.k_prime)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cand.get_top_k_outputs(q, args.k, kw, tk, inv, truncate_k_prime_to=args.k_prime)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / args.steps * 1e3
            print(f"{args.workload} N={N} B={args.batch} {pr:12s} {ms:7.4f} ms per step  {args.batch / ms * 1e3:9.1f} queries/s  {_stats(tk)}")


if __name__ == "__main__":
    main()
