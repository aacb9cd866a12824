#!/usr/bin/env python3
"""Randomised check of MoLSimilarity.forward (prologue + index build + scoring kernel auto-selection) against the CPU oracle:
random batch sizes (1..70), ragged corpus sizes, every built shape, both precision modes."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from oracle import mol_oracle as O

dev = torch.device("cuda", 0)
random.seed(11)
worst = {"fp32": 0.0, "f16x3": 0.0}
fails = 0
for case in range(36):
    name = random.choice(["amzn-books", "ml-1m", "ml-20m", "synthetic-16x16x64"])
    cfg = O.CONFIGS[name]
    B = random.choice([1, 2, 3, 4, 7, 8, 16, 31, 32, 33, 48, 64, 70])
    n = random.choice([1, 31, 32, 33, 500, 2047, 4096, 4097, 9000, random.randint(100, 12000)])
    prec = "fp32" if name == "synthetic-16x16x64" or random.random() < 0.6 else "f16x3"
    w = O.synthetic_weights(cfg, seed=case)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    mol.precision = prec
    X = torch.from_numpy(O.hash_item_table(case, 0, n, cfg.item_embedding_dim)).unsqueeze(0)
    q = O.synthetic_queries(cfg, B, seed=case)
    uid = torch.randint(0, 5000, (B,), generator=torch.Generator().manual_seed(case)) if len(cfg.uid_embedding_hash_sizes) else None
    ref = O.mol_logits(cfg, w, q, X, uid)
    kw = {"user_ids": uid.to(dev)} if uid is not None else {}
    with torch.inference_mode():
        got, _ = mol(q.to(dev), X.to(dev), **kw)
    err = float((got.cpu() - ref).abs().max())
    worst[prec] = max(worst[prec], err)
    if not (err <= 1e-4):
        fails += 1
        print("FAIL", name, B, n, prec, err)
print(f"scoring fuzz done: {fails} failures; max |logit - oracle| fp32 {worst['fp32']:.2e}, f16x3 {worst['f16x3']:.2e} (bar 1e-4)")
