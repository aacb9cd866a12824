#!/usr/bin/env python3
"""Golden vectors for the model VARIANTS and SHAPES beyond the five BASELINE configs: runs the
REFERENCE (imported from /root/reference, build container only) and writes tests/golden/variants.npz.

TEST INFRASTRUCTURE ONLY: arrays in, arrays out; nothing of the reference is copied.
  python oracle/gen_golden_variants.py

Cases (create_mol_interaction_module arguments, modeling/similarity_utils.py:41-245):
  v_plainq_8x8x64      query_hidden_dim = -1 (plain Linear query projection), shape 8x8x64
  v_itemglu_8x4x32     item_hidden_dim = 96 (GLU item projection, swiglu), shape 8x4x32
  v_none_16x4x32       gating_combination_type = "none" with the pair gate only (gating_query_fn = gating_item_fn = False), shape
                       16x4x32.  (With a query-only or item-only part the reference's in-place `gating_inputs += ...` cannot
                       broadcast (B,1,L) / (1,X,L) up to (B,X,L) and raises, similarity_fn.py:187-197: pair-only is the reachable form.)
  v_h64_8x4x64         gating_qi_hidden_dim = 64, shape 8x4x64
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG  # noqa: E402  (sets up the reference import + shims)

import json  # noqa: E402

import numpy as np  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _npz import savez_deterministic  # noqa: E402
import torch  # noqa: E402

from modeling.similarity_utils import create_mol_interaction_module  # noqa: E402  (reference)
from oracle.mol_oracle import MoLConfig, hash_item_table, synthetic_queries  # noqa: E402

CASES = {
    "v_plainq_8x8x64": (MoLConfig(64, 64, 64, 8, 8, query_hidden_dim=-1), {}),
    "v_itemglu_8x4x32": (MoLConfig(48, 40, 32, 8, 4, item_hidden_dim=96, item_nonlinearity="swiglu"), {}),
    "v_none_16x4x32": (MoLConfig(64, 64, 32, 16, 4, gating_combination_type="none", gating_query_fn=False, gating_item_fn=False), {}),
    "v_h64_8x4x64": (MoLConfig(64, 64, 64, 8, 4, gating_qi_hidden_dim=64), {}),
    "v_nohid_8x8x32": (MoLConfig(64, 64, 32, 8, 8, gating_qi_hidden_dim=-1), {}),    # pair gate = one Linear(L, L): similarity_utils.py:199-206
}


def build(cfg: MoLConfig, seed: int):
    torch.manual_seed(seed)
    mol, _ = create_mol_interaction_module(
        query_embedding_dim=cfg.query_embedding_dim, item_embedding_dim=cfg.item_embedding_dim,
        dot_product_dimension=cfg.dot_product_dimension, query_dot_product_groups=cfg.query_dot_product_groups,
        item_dot_product_groups=cfg.item_dot_product_groups, temperature=cfg.temperature, query_dropout_rate=0.0,
        query_hidden_dim=cfg.query_hidden_dim, item_dropout_rate=0.1, item_hidden_dim=cfg.item_hidden_dim,
        gating_query_hidden_dim=cfg.gating_query_hidden_dim, gating_qi_hidden_dim=cfg.gating_qi_hidden_dim,
        gating_item_hidden_dim=cfg.gating_item_hidden_dim, softmax_dropout_rate=cfg.softmax_dropout_rate, bf16_training=False,
        gating_query_fn=cfg.gating_query_fn, gating_item_fn=cfg.gating_item_fn, query_nonlinearity=cfg.query_nonlinearity,
        item_nonlinearity=cfg.item_nonlinearity, gating_combination_type=cfg.gating_combination_type, eps=cfg.eps)
    mol.eval()
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():   # non-zero biases, so that a port which drops one cannot pass
        for name, p in mol.named_parameters():
            if name.endswith("bias") or name.endswith("_b"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return mol


def main():
    out = {}
    for i, (name, (cfg, _)) in enumerate(CASES.items()):
        mol = build(cfg, 40 + i)
        N, B = 517, 6
        X = torch.from_numpy(hash_item_table(30 + i, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
        q = synthetic_queries(cfg, B, seed=50 + i)
        with torch.inference_mode():
            logits, aux = mol(q, X)
            assert aux == {}
            eq, _ = mol.get_query_component_embeddings(q)
            ex, _ = mol.get_item_component_embeddings(X)
            cand = X.squeeze(0)[torch.randint(0, N, (B, 40), generator=torch.Generator().manual_seed(7))]
            rows, _ = mol(q, cand)
        out[f"{name}/cfg_json"] = np.array(json.dumps(cfg.to_dict()))
        out[f"{name}/q"], out[f"{name}/X"], out[f"{name}/cand"] = q.numpy(), X.numpy(), cand.numpy()
        out[f"{name}/logits"], out[f"{name}/Eq"], out[f"{name}/Ex"], out[f"{name}/row_logits"] = logits.numpy(), eq.numpy(), ex.numpy(), rows.numpy()
        for k, v in mol.state_dict().items():
            out[f"{name}/w/{k}"] = v.detach().numpy()
    savez_deterministic(os.path.join(GG.OUT, "variants.npz"), **out)
    print("wrote variants.npz:", {k: v.shape for k, v in out.items() if k.endswith("logits")})


if __name__ == "__main__":
    main()
