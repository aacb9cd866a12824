#!/usr/bin/env python3
"""A/B harness for the scoring kernel: interleaved rounds of every variant in ONE process
(cdna_hip_programming.md section 5.4 rule 24), median/min ms and TFLOP/s per variant, max |diff| vs variant 0.
Variants are selected by the RAILS_SCORE_VARIANT env var read at launch time by librails_amd.so; a trailing "n" (e.g. "2n")
also sets RAILS_F16_OVERLAP=0 (f16x3 kernels without the cross-query overlap of stage X).
  python tools/score_bench.py --variants 0,1,2 --workload amzn-books --batch 32 --rounds 7
"""
import argparse
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402  (input generator only)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0")
    ap.add_argument("--workload", default="amzn-books")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--items", type=int, default=0)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", default=None, help="fp32 | f16x3 (default: RAILS_PRECISION or fp32)")
    ap.add_argument("--check-items", type=int, default=0, help="compare this many sampled columns with the CPU oracle")
    ap.add_argument("--upper", action="store_true", help="precision f16x3: the UPPER first pass (rails_mol_score_dense_upper, the product's per-pair bound added to the logit)")
    args = ap.parse_args()
    variants = [v for v in args.variants.split(",")]
    cfg_key, N, _ = bench.WORKLOADS[args.workload]
    if args.items:
        N = args.items
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
    mol.precision = args.precision
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).to(dev)
    q = O.synthetic_queries(cfg, args.batch).to(dev)
    uid = None
    if cfg.uid_embedding_hash_sizes:
        uid = torch.arange(args.batch, dtype=torch.int64, device=dev)
    with torch.inference_mode():
        eng = mol.engine()
        index = eng.build_index(X)
        qpack, _, _ = eng.query_pack(q, uid)
        outs, times = {}, {v: [] for v in variants}
        score = eng.score_dense
        if args.upper:
            from rails_amd import f16x3_bound as FB

            p_ = "_gating_fn._qi_partial_module."
            poly = FB.upper_bound_poly(w[p_ + "1.weight"], w[p_ + "1.bias"], w[p_ + "3.weight"], w[p_ + "3.bias"], cfg.temperature, cfg.dot_product_dimension,
                                       cfg.query_dot_product_groups, cfg.item_dot_product_groups)["poly"]
            score = lambda qp, b, idx, out=None: eng.score_dense_upper(qp, b, idx, poly, out=out)      # noqa: E731

        def select(v):
            os.environ["RAILS_SCORE_VARIANT"] = v.rstrip("n")
            os.environ["RAILS_F16_OVERLAP"] = "0" if v.endswith("n") else "1"

        for v in variants:
            select(v)
            outs[v] = score(qpack, args.batch, index).clone()
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for v in variants:
                select(v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    score(qpack, args.batch, index, out=outs[v])
                e1.record()
                torch.cuda.synchronize()
                times[v].append(e0.elapsed_time(e1) / args.reps)
    if args.check_items:
        g = torch.Generator().manual_seed(0)
        cols = torch.randperm(N, generator=g)[: args.check_items]
        ref = O.mol_logits(cfg, w, q.cpu(), X[cols.to(dev)].cpu().unsqueeze(0), None if uid is None else uid.cpu())
        err = (outs[variants[0]][:, cols.to(dev)].cpu() - ref).abs()
        print(f"precision {eng.precision}: max|logit - oracle| over {args.check_items} sampled items x {args.batch} queries = {float(err.max()):.3e}  (mean {float(err.mean()):.3e})")
    flops = args.batch * N * bench.flops_per_pair(cfg)
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        diff = float((outs[v] - outs[variants[0]]).abs().max())
        print(f"variant {v}: median {med:.3f} ms  min {mn:.3f} ms  {flops / med / 1e9:.1f} TFLOP/s (median)  "
              f"{flops / mn / 1e9 / bench.PEAK_F32_MFMA_TFLOPS * 100:.1f}% of peak (min)  max|d vs v{variants[0]}| {diff:.2e}")


if __name__ == "__main__":
    main()
