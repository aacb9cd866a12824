#!/usr/bin/env python3
"""HR@k / NDCG@k / MRR parity of the GPU path with the CPU oracle (BASELINE.json's "HR@10/50 parity"): full ML-20M-size and
amzn-books-shaped corpora of synthetic items, random-init MoL weights, and targets PLANTED at known oracle ranks (uniform in
1..100, or absent) so that the metrics are non-trivial.  Both sides run the reference's harness arithmetic
(data/eval.py:194-243) on their own top-k ids; the ids themselves are compared as well."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import eval_harness as H
from oracle import mol_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=128)
a = ap.parse_args()
dev = torch.device("cuda", 0)
rows = []
for name, N in (("ml-20m", 27278), ("amzn-books", 60000), ("ml-1m", 3883)):
    cfg = O.CONFIGS[name]
    w = O.synthetic_weights(cfg, seed=1)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    B, k = a.batch, 120
    X = torch.from_numpy(O.hash_item_table(2, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
    ids = (torch.arange(N, dtype=torch.int64) * 3 + 7).unsqueeze(0)          # sparse ids, as the datasets have
    q = O.synthetic_queries(cfg, B, seed=5)
    g = torch.Generator().manual_seed(6)
    uid = torch.randint(0, 5000, (B,), generator=g) if len(cfg.uid_embedding_hash_sizes) else None
    seen = torch.zeros((B, 40), dtype=torch.int64)
    t0 = time.time()
    rs, ri, _ = O.brute_force_topk(cfg, w, q, X, ids, O.k_prime(k, seen, N, 200), uid)
    oracle_s = time.time() - t0
    # seen ids: 20 of each row's own top-60 (so the filter really removes candidates), rest padding
    for b in range(B):
        seen[b, :20] = ri[b, torch.randperm(60, generator=g)[:20]]
    ref_ids, _ = O.filter_seen_ids(ri, rs, seen, k)
    # targets: the id at a planted oracle rank (1..100) for 85 % of the rows, an id that is not retrieved for the rest
    planted = torch.randint(0, 100, (B,), generator=g)
    target = ref_ids[torch.arange(B), planted].clone()
    absent = torch.rand(B, generator=g) < 0.15
    target[absent] = ids[0, -1]          # the lowest-ranked corner of the corpus: practically never in a top-120
    target = target.unsqueeze(1)
    ref = O.eval_metrics(ref_ids, target, 120)

    class Enc:      # the encoder is upstream of this check: replay the query embeddings
        def encode(self, **kw): return q.to(dev)
        def get_item_embeddings(self, item_ids): return X.to(dev)[0][(item_ids.to(dev) - 7) // 3]
    model = Enc(); model._ndp_module = mol
    feats = H.SequentialFeatures(torch.full((B,), 40), seen.to(dev), None, {"user_ids": uid.to(dev)} if uid is not None else {})
    with torch.inference_mode():
        state = H.get_eval_state(model, ids[0].tolist(), None, lambda e, i: rails_amd.MoLBruteForceTopK(mol, e, i), dev)
        out = H.eval_metrics_v2_from_tensors(state, model, feats, target.to(dev), include_eval_time=True, include_eval_top_k_ids=True)
    got_ids = out["eval_top_k_ids"].cpu()
    row = {"workload": f"{name} shape, N={N}, B={B}, k=120, k'=160, 20 seen ids per row", "ids_identical_fraction": float((got_ids == ref_ids).float().mean()),
           "rows_with_identical_ids": int((got_ids == ref_ids).all(1).sum()), "oracle_cpu_seconds": oracle_s}
    for key in ("hr@1", "hr@10", "hr@50", "hr@100", "ndcg@10", "ndcg@50", "mrr"):
        mine, theirs = out[key].float().cpu(), ref[key].float()
        row[key] = {"gpu": float(mine.mean()), "oracle": float(theirs.mean()), "rows_differing": int(((mine - theirs).abs() > 1e-6).sum()),
                    "max_abs_diff": float((mine - theirs).abs().max())}
    rows.append(row)
print(json.dumps({"what": "HR / NDCG / MRR of the HIP path vs the CPU oracle on the same inputs (targets planted at known oracle ranks)", "rows": rows}, indent=1))
