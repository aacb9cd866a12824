"""Timing of rails_mol_score_indexed (per-row candidates read in place): 8x8x32, 1 000 candidates per row, B = 32 / 128, per
RAILS_SCORE_VARIANT.  Round 4 tried an indexed instantiation of the small-unit shell with it (2 016 units of 16 candidates at four
waves per SIMD instead of 1 024 of 32 at two): 63.3 vs 56.7 us at B = 32, 221 vs 217 us at B = 128 -- not kept.  The launch is bound
by the gather itself: a candidate's operands are 64+ float4 pieces 1 KiB apart in the tile-packed index, one cache line each."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O
dev = torch.device("cuda:0")
cfg = O.CONFIGS["synthetic-8x8x32"]
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
N = 4_000_000
with torch.inference_mode():
    eng = mol.engine()
    X = E.hash_item_table(1, 0, N, cfg.item_embedding_dim, dev)
    index = eng.build_index(X)
    for B in (32, 128):
        q = O.synthetic_queries(cfg, B).to(dev)
        qpack, _, _ = eng.query_pack(q, None)
        pos = torch.randint(0, N, (B, 1000), device=dev)
        for v in ("0", "1"):
            os.environ["RAILS_SCORE_VARIANT"] = v
            for _ in range(3): out = eng.score_indexed(qpack, B, index, pos)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): out = eng.score_indexed(qpack, B, index, pos)
            e1.record(); torch.cuda.synchronize()
            print(f"B={B} variant {v}: {e0.elapsed_time(e1)/50*1e3:.1f} us per call", float(out.sum()))
