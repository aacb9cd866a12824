"""cProfile of the Python side of a two-pass call (MoLAvgTopK through CandidateIndex.get_top_k_outputs, the speculative route of large
shards forced on a 2 M-item corpus): where the host time of a batch goes -- it sits on the critical path of the plain (unpipelined) call."""
import cProfile, pstats, sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rails_amd
from rails_amd import engine as E
from oracle import mol_oracle as O
dev = torch.device("cuda", 0)
cfg = O.CONFIGS["synthetic-8x8x32"]; N = 2_000_000; B = 32
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True); mol = mol.to(dev).eval()
X = E.hash_item_table(1, 0, N, cfg.item_embedding_dim, dev).unsqueeze(0)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
inv = torch.zeros((B, 61), dtype=torch.int64, device=dev)
rails_amd.MoLAvgTopK.DEVICE_REDO_BYTES = 0
rails_amd.MoLAvgTopK.PREFILTER_MIN_ITEMS = 1
with torch.inference_mode():
    at = rails_amd.MoLAvgTopK(mol, X, ids, avg_top_k=1000)
    cand = rails_amd.CandidateIndex(ids=ids, embeddings=X)
    step = lambda: cand.get_top_k_outputs(q, 120, {}, at, inv, truncate_k_prime_to=200)
    for _ in range(20): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): step()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"plain call: {1e6*(t1-t0)/500:.1f} us per batch (GPU + host round trip)")
    pr = cProfile.Profile(); pr.enable()
    for _ in range(500): step()
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
