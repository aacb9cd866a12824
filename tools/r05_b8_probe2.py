"""Replays bench.measurement_matrix's proved leg with host timers around every engine call: which call stalls?"""
import os, sys, time, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench, rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O
gc.disable()
rails_amd.MoLBruteForceTopK.EXACT_MODE = "dense"
cfg = O.CONFIGS["amzn-books"]; N = 695762; dev = torch.device("cuda:0")
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups, cfg.item_dot_product_groups,
    cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim, cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim,
    cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False, query_nonlinearity=cfg.query_nonlinearity)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, 32).to(dev)
inv = ids[0, torch.randint(0, N, (32, 61), device=dev)]
log = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); dt = time.perf_counter() - t0
        if dt > 2e-3: log.append((name, round(dt * 1e3, 2)))
        return r
    setattr(obj, name, g)
for n in ("topk", "rescore_select", "rescore_verdict", "filter_seen_ids", "topk_filtered"):
    wrap(E, n)
for n in ("score_dense", "score_indexed", "query_pack_both", "query_pack", "build_index", "gather_index"):
    wrap(E.MolEngine, n)
wrap(rails_amd.MoLBruteForceTopK, "_absorb_state"); wrap(rails_amd.MoLBruteForceTopK, "_bind"); wrap(rails_amd.MoLBruteForceTopK, "_proved_eps")
wrap(torch.cuda, "synchronize")
with torch.inference_mode():
    pts = bench.measurement_matrix(mol, X, ids, q, {}, inv, cfg, N, 5, dev)
for p in pts:
    print(p["precision"], p["batch"], p["k_prime"], round(p["ms_per_step"], 3), round(p["ms_per_step_stdev"], 3))
print("calls over 2 ms:", log)
