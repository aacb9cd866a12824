#!/usr/bin/env python3
"""Randomised check: rails_mol_coarse_topk / rails_mol_component_topk == (materialised scores + rails_topk) bit for bit,
whenever the returned candidate counts are in range (and that they ARE in range on untied random data)."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O

dev = torch.device("cuda", 0)
random.seed(3)
bad = out_of_range = 0
for case in range(24):
    name = random.choice(["amzn-books", "ml-1m", "ml-20m"])
    cfg = O.CONFIGS[name]
    n = random.randint(262144, 900000)
    B = random.choice([1, 5, 16, 32, 40])
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(O.synthetic_weights(cfg, seed=case), strict=True)
    mol = mol.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(case)
    X = (torch.fmod(torch.randn((1, n, cfg.item_embedding_dim), generator=g, device=dev), 2.0) * 0.02)
    ids = torch.arange(1, n + 1, dtype=torch.int64, device=dev).unsqueeze(0)
    q = O.synthetic_queries(cfg, B, seed=case).to(dev)
    uid = torch.arange(B, dtype=torch.int64, device=dev) if len(cfg.uid_embedding_hash_sizes) else None
    with torch.inference_mode():
        at = rails_amd.MoLAvgTopK(mol, X, ids, avg_top_k=100)
        eng = at._bind()
        _, eq, _ = eng.query_pack(q, uid, want_plain=True)
        kp = random.choice([1, 10, 100, 777, 2000, 4096])
        avg = random.random() < 0.5
        fs, fp, cnt = eng.coarse_topk(eq, at._table(), avg, kp)
        rs, rp = E.topk(eng.coarse_scores(eq, at._table(), avg), kp)
        in_range = int(cnt.min()) >= kp and int(cnt.max()) <= eng.coarse_topk_capacity(kp)
        out_of_range += not in_range
        if in_range and not (torch.equal(fs, rs) and torch.equal(fp, rp)):
            bad += 1; print("COARSE MISMATCH", name, n, B, kp, avg)
        # the int8 pre-filter of the streaming pass, on the table as it is and on a copy whose rows are rescaled at random (x 1e-3 .. 1e3,
        # a few rows x 1e4: the single scale then serves some rows badly): same candidates, counts and output
        for tbl in (at._table(), (at._table().float() * torch.pow(10.0, torch.randint(-3, 4, (at._table().shape[0], 1), device=dev).float())).to(torch.bfloat16)):
            pre = eng.build_coarse_prefilter(tbl)
            if pre is None:
                continue
            a0, b0, c0 = eng.coarse_topk(eq, tbl, avg, kp)
            a1, b1, c1 = eng.coarse_topk(eq, tbl, avg, kp, prefilter=pre)
            ok_rows = (c0 >= kp) & (c0 <= eng.coarse_topk_capacity(kp))
            if not torch.equal(c0, c1) or not (torch.equal(a0[ok_rows], a1[ok_rows]) and torch.equal(b0[ok_rows], b1[ok_rows])):
                bad += 1; print("PREFILTER MISMATCH", name, n, B, kp, avg)
        if B <= 16 and case % 3 == 0:
            nt = rails_amd.MoLNaiveTopK(mol, X, ids, k_per_group=5)
            kg = random.choice([1, 5, 50, 100])
            tab = nt._component_table()
            fs, fp, cnt = eng.component_topk(eq, tab, kg)
            rs, rp = E.topk(eng.component_scores(eq, tab), kg)
            in_range = int(cnt.min()) >= kg and int(cnt.max()) <= eng.coarse_topk_capacity(kg)
            out_of_range += not in_range
            if in_range and not (torch.equal(fs, rs) and torch.equal(fp, rp)):
                bad += 1; print("COMPONENT MISMATCH", name, n, B, kg)
    del X, at, mol
    torch.cuda.empty_cache()
print(f"fused-scan fuzz done: mismatches {bad}, count-out-of-range cases {out_of_range}")
