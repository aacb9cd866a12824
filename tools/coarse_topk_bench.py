#!/usr/bin/env python3
"""The fused coarse top-K' (config 5, pass 1) on its own: a bf16 table of N x d random rows (no MoL index is built, so a 125 M-item
shard needs 8 GB), B queries, K' = --avg-top-k.  Prints ms per call (HIP events over --reps calls), the table bytes / ms, the
candidate counts, and -- with --check -- compares (scores, positions) with the materialising path on the first --check items.
  python tools/coarse_topk_bench.py --items 125000000 --batch 32,128 --avg-top-k 1000
Under rocprofv3 --kernel-trace --stats this gives the per-kernel split of the pass (sample, threshold, select scan, key selection)."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from rails_amd import engine as E  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (weights generator only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--items", type=int, default=125_000_000)
    ap.add_argument("--batch", default="32")
    ap.add_argument("--avg-top-k", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--check", type=int, default=0, help="compare with coarse_scores + topk on a table of this many items first")
    ap.add_argument("--prefilter", default="both", choices=["off", "on", "both"], help="the int8 pre-filter of the streaming pass (rails_mol_coarse_prefilter_build)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = O.CONFIGS["synthetic-8x8x32"]
    w = O.synthetic_weights(cfg, seed=0)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    d, pq = cfg.dot_product_dimension, cfg.query_dot_product_groups
    g = torch.Generator(device=dev).manual_seed(1)
    with torch.inference_mode():
        eng = mol.engine()

        def table_of(n):
            t = torch.empty((n, d), dtype=torch.bfloat16, device=dev)
            step = 1 << 24
            for lo in range(0, n, step):   # rows of norm ~ 1 / sqrt(P_X): what averaging P_X unit vectors gives
                m = min(step, n - lo)
                t[lo : lo + m] = (torch.randn((m, d), device=dev, generator=g) * (1.0 / (d * cfg.item_dot_product_groups) ** 0.5)).to(torch.bfloat16)
            return t

        def queries(B):
            eq = torch.randn((B, pq, d), device=dev, generator=g)
            return eq / eq.norm(dim=-1, keepdim=True)

        if args.check:
            t = table_of(args.check)
            pre = eng.build_coarse_prefilter(t)
            for B in (1, 32, 77):
                eq = queries(B)
                sc, pos, counts = eng.coarse_topk(eq, t, False, args.avg_top_k)
                ref_s, ref_p = E.topk(eng.coarse_scores(eq, t, False), args.avg_top_k)
                cap = eng.coarse_topk_capacity(args.avg_top_k)
                ok = bool(((counts >= args.avg_top_k) & (counts <= cap)).all())
                print(f"check N={args.check} B={B}: counts {int(counts.min())}..{int(counts.max())} (cap {cap}) in range {ok}; "
                      f"scores equal {torch.equal(sc, ref_s)}, positions equal {torch.equal(pos, ref_p)}")
                s8, p8, c8 = eng.coarse_topk(eq, t, False, args.avg_top_k, prefilter=pre)
                print(f"      with the int8 pre-filter: scores equal {torch.equal(s8, ref_s)}, positions equal {torch.equal(p8, ref_p)}, counts equal {torch.equal(c8, counts)}")
            del t, pre
        table = table_of(args.items)
        pre_all = eng.build_coarse_prefilter(table) if args.prefilter != "off" else None
        for B in [int(b) for b in args.batch.split(",")]:
            eq = queries(B)
            for mode in (["off", "on"] if args.prefilter == "both" else [args.prefilter]):
                pre = pre_all if mode == "on" else None
                for _ in range(3):
                    out = eng.coarse_topk(eq, table, False, args.avg_top_k, prefilter=pre)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    out = eng.coarse_topk(eq, table, False, args.avg_top_k, prefilter=pre)
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / args.reps
                counts = out[2]
                print(f"N={args.items} B={B} K'={args.avg_top_k} pre-filter {mode}: {ms:.4f} ms per call = {table.numel() * 2 / ms / 1e9:.2f} TB/s of bf16-table bytes; "
                      f"candidates per query {int(counts.min())}..{int(counts.max())} mean {float(counts.float().mean()):.0f} of {eng.coarse_topk_capacity(args.avg_top_k)} slots")


if __name__ == "__main__":
    main()
