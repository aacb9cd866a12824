#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE implementation (imported from /root/reference,
build container only) on seeded inputs and writes plain-array fixtures to tests/golden/*.npz.

TEST INFRASTRUCTURE ONLY.  Nothing from the reference (source, bytecode, pickled modules) is
copied: fixtures hold input arrays, the module's weights as arrays and the reference's outputs.
Re-run with:  python oracle/gen_golden.py   (needs /root/reference; CPU only, ~1 min)

Fixture families:
  F1  every stage of MoLSimilarity.forward (Eq, Ex, gq, gi, cl, gqi, w, pi, logits)
  F2  MoLBruteForceTopK (scores, ids) for k in {10, 200, N}
  F3  CandidateIndex.get_top_k_outputs with seen-id filtering, back-fill and truncate_k_prime_to
  F4  MoLAvgTopK.forward / topk_ids, incl. the k > avg_top_k ValueError
  F5  eval_metrics_v2_from_tensors (rank, HR@k, NDCG@k, MRR) driven through the reference harness
  F6  per-row candidates branch of MoLSimilarity.forward (B' == B)
  F7  one full-size case for ML-1M (N=3883) and ML-20M (N=27278): B=32, k=200
  F9  MIPSBruteForceTopK + DotProductSimilarity (all three shape branches)
  F10 MoLNaiveTopK / MoLCombTopK outputs and the candidate union the reference reranked
"""
import os
import sys
import types

os.environ.setdefault("TORCH_COMPILE_DISABLE", "1")  # similarity_fn.py:31 is @torch.compile; eager == compiled to 1.2e-6

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# harness-only shims for packages the container lacks; they never ship
gin = types.ModuleType("gin")
gin.configurable = lambda f=None, **kw: (f if f is not None else (lambda g: g))
sys.modules["gin"] = gin
tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = tb
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _npz import savez_deterministic  # noqa: E402
import torch  # noqa: E402

from modeling.similarity_utils import create_mol_interaction_module  # noqa: E402  (reference)
from rails.indexing.mol_top_k import MoLAvgTopK, MoLBruteForceTopK  # noqa: E402  (reference)
from rails.similarities.mol.similarity_fn import _softmax_dropout_combiner_fn  # noqa: E402  (reference)
from indexing.candidate_index import CandidateIndex  # noqa: E402  (reference)

from oracle.mol_oracle import CONFIGS, MoLConfig, hash_item_table, synthetic_queries  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def build_reference_module(cfg: MoLConfig, seed: int):
    torch.manual_seed(seed)
    mol, _ = create_mol_interaction_module(
        query_embedding_dim=cfg.query_embedding_dim,
        item_embedding_dim=cfg.item_embedding_dim,
        dot_product_dimension=cfg.dot_product_dimension,
        query_dot_product_groups=cfg.query_dot_product_groups,
        item_dot_product_groups=cfg.item_dot_product_groups,
        temperature=cfg.temperature,
        query_dropout_rate=0.0,
        query_hidden_dim=cfg.query_hidden_dim,
        item_dropout_rate=0.1,
        item_hidden_dim=cfg.item_hidden_dim,
        gating_query_hidden_dim=cfg.gating_query_hidden_dim,
        gating_qi_hidden_dim=cfg.gating_qi_hidden_dim,
        gating_item_hidden_dim=cfg.gating_item_hidden_dim,
        softmax_dropout_rate=cfg.softmax_dropout_rate,
        bf16_training=False,
        query_nonlinearity=cfg.query_nonlinearity,
        item_nonlinearity=cfg.item_nonlinearity,
        uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None,
        gating_combination_type=cfg.gating_combination_type,
        eps=cfg.eps,
    )
    mol.eval()
    # the gate/bias parameters are zero-initialised by the reference; give them seeded non-zero
    # values so that a port which drops a bias cannot pass
    g = torch.Generator().manual_seed(seed + 1000)
    with torch.no_grad():
        for name, p in mol.named_parameters():
            if name.endswith("bias") or name.endswith("_b"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return mol


def slim_weights(mol, cfg: MoLConfig, user_ids):
    """state_dict as numpy; uid tables are sliced to the rows the fixture touches."""
    sd = {k: v.detach().clone() for k, v in mol.state_dict().items()}
    remap = None
    if len(cfg.uid_embedding_hash_sizes) > 0:
        assert len(cfg.uid_embedding_hash_sizes) == 1
        hs = cfg.uid_embedding_hash_sizes[0]
        rows = torch.unique((user_ids % hs) + 1)
        key = "_query_embeddings_fn._uid_embeddings_0.weight"
        sd[key + ".rows"] = rows
        sd[key] = sd[key][rows]
    return {k: v.numpy() for k, v in sd.items()}


def stage_outputs(mol, q, X, kw):
    """All intermediates of the reference forward, captured with hooks on its own sub-modules."""
    cap = {}
    g = mol._gating_fn
    hooks = [
        g.register_forward_pre_hook(lambda m, a, k: cap.__setitem__("cl", k["logits"].clone()), with_kwargs=True),
        g._qi_partial_module.register_forward_hook(lambda m, a, o: cap.__setitem__("gqi", o.clone())),
        g._query_only_partial_module.register_forward_hook(lambda m, a, o: cap.__setitem__("gq", o.clone())),
        g._item_only_partial_module.register_forward_hook(lambda m, a, o: cap.__setitem__("gi", o.clone())),
        g._normalization_fn.register_forward_pre_hook(lambda m, a: cap.__setitem__("w", a[0].clone())),
    ]
    with torch.inference_mode():
        logits, aux = mol(q, X, **kw)
        assert aux == {}
        cap["Eq"], _ = mol.get_query_component_embeddings(q, **kw)
        cap["Ex"], _ = mol.get_item_component_embeddings(X)
        cap["pi"], comb = _softmax_dropout_combiner_fn(
            x=cap["w"], y=cap["cl"], dropout_pr=g._normalization_fn._dropout_rate, eps=g._normalization_fn._eps, training=False
        )
        assert torch.equal(comb, logits)
    for h in hooks:
        h.remove()
    cap["logits"] = logits
    return {k: v.numpy() for k, v in cap.items()}


def sparse_item_ids(n: int, seed: int) -> torch.Tensor:
    """1-based, strictly increasing, sparse ids (ML ids are sparse: position != id)."""
    g = torch.Generator().manual_seed(seed)
    gaps = torch.randint(1, 4, (n,), generator=g)
    return torch.cumsum(gaps, 0).to(torch.int64).unsqueeze(0)


def make_invalid_ids(top_ids: torch.Tensor, width: int, seed: int, heavy_rows=()) -> torch.Tensor:
    """(B, W) int64: a sample of each row's own top ids (so the filter has work) + zero padding."""
    g = torch.Generator().manual_seed(seed)
    B, kp = top_ids.shape
    inv = torch.zeros((B, width), dtype=torch.int64)
    for b in range(B):
        n = min(width if b in heavy_rows else width // 2, kp - 3)
        perm = torch.randperm(kp, generator=g)[:n]
        inv[b, :n] = top_ids[b, perm]
    return inv


def per_config_fixture(name: str, cfg: MoLConfig, seed: int, B: int = 6, N: int = 1024, n_stage: int = 200):
    mol = build_reference_module(cfg, seed)
    q = synthetic_queries(cfg, B, seed=seed + 2)
    X = torch.from_numpy(hash_item_table(seed + 1, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
    ids = sparse_item_ids(N, seed + 3)
    kw = {}
    if len(cfg.uid_embedding_hash_sizes) > 0:
        g = torch.Generator().manual_seed(seed + 4)
        kw["user_ids"] = torch.randint(0, 200000, (B,), generator=g, dtype=torch.int64)
    out = {"cfg_json": np.array(__import__("json").dumps(cfg.to_dict())), "torch_version": np.array(torch.__version__)}
    for k, v in slim_weights(mol, cfg, kw.get("user_ids")).items():
        out["w/" + k] = v
    out["q"], out["X"], out["item_ids"] = q.numpy(), X.numpy(), ids.numpy()
    if "user_ids" in kw:
        out["user_ids"] = kw["user_ids"].numpy()

    # F1
    for k, v in stage_outputs(mol, q, X[:, :n_stage], kw).items():
        out["F1/" + k] = v
    out["F1/n"] = np.array(n_stage)

    with torch.inference_mode():
        # F2
        bf = MoLBruteForceTopK(mol, X, ids)
        all_logits, _ = mol(q, X, **kw)
        out["F2/all_logits"] = all_logits.numpy()
        for k in (10, 200, N):
            s, i = bf(q, k=k, sorted=True, **kw)
            out[f"F2/k{k}/scores"], out[f"F2/k{k}/ids"] = s.numpy(), i.numpy()

        # F3: timing protocol (k=120, truncate 200), accuracy protocol (no truncate), back-fill rows
        ci = CandidateIndex(ids=ids, embeddings=X)
        _, top_ids_300 = bf(q, k=300, **kw)
        cases = [
            ("timing", 120, 61, 200, ()),
            ("accuracy", 100, 40, None, ()),
            ("backfill", 20, 30, 25, (0, 3)),  # k'=25: heavy rows lose >5 ids to the filter -> back-fill
            ("nofilter", 50, 0, None, ()),
        ]
        for cname, k, width, trunc, heavy in cases:
            inv = None
            if width > 0:
                kp = min(k + width, N) if trunc is None else min(k + width, N, trunc)
                inv = make_invalid_ids(top_ids_300[:, :kp], width, seed + 5, heavy)
                out[f"F3/{cname}/invalid_ids"] = inv.numpy()
            r_ids, r_scores, r_emb = ci.get_top_k_outputs(
                query_embeddings=q, k=k, aux_payloads=kw, top_k_module=bf, invalid_ids=inv,
                return_embeddings=False, truncate_k_prime_to=trunc,
            )
            assert r_emb is None
            out[f"F3/{cname}/k"] = np.array(k)
            out[f"F3/{cname}/truncate"] = np.array(-1 if trunc is None else trunc)
            out[f"F3/{cname}/ids"], out[f"F3/{cname}/scores"] = r_ids.numpy(), r_scores.numpy()

        # F4
        for avg_k in (100, 500):
            at = MoLAvgTopK(mol, X, ids, avg_top_k=avg_k)
            s, i = at(q, k=50, sorted=True, **kw)
            out[f"F4/a{avg_k}/scores"], out[f"F4/a{avg_k}/ids"] = s.numpy(), i.numpy()
            out[f"F4/a{avg_k}/coarse_topk_idx_sorted"] = at.topk_ids(q, sorted=True, **kw).numpy()
            # what pass 1 saw (bf16 mm): recorded so that the coarse stage can be checked in isolation
            eqs, _ = mol.get_query_component_embeddings(q, decoupled_inference=True, **kw)
            coarse = torch.mm(eqs.sum(1).to(torch.bfloat16), at._avg_mol_item_embeddings_t)
            out[f"F4/a{avg_k}/coarse_scores_bf16_as_f32"] = coarse.float().numpy()
            _, cidx = torch.topk(coarse, k=avg_k, dim=1, sorted=False)
            out[f"F4/a{avg_k}/coarse_idx_forward"] = cidx.numpy()
            try:
                at(q, k=avg_k + 1, **kw)
                raised = False
            except ValueError:
                raised = True
            out[f"F4/a{avg_k}/raises_when_k_gt_avg"] = np.array(raised)

        # F6: per-row candidates (B' == B)
        g = torch.Generator().manual_seed(seed + 6)
        cand_idx = torch.stack([torch.randperm(N, generator=g)[:48] for _ in range(B)])
        cand = X.squeeze(0)[cand_idx]
        rows, _ = mol(q, cand, **kw)
        out["F6/cand_idx"], out["F6/logits"] = cand_idx.numpy(), rows.numpy()
    savez_deterministic(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: wrote {len(out)} arrays")


def harness_fixture(seed: int = 11):
    """F5: drive the reference's own eval harness (data/eval.py:76-268) with a stub encoder so that
    rank / HR / NDCG / MRR come from the reference's code, not from a restatement."""
    from data.eval import EvalState, eval_metrics_v2_from_tensors  # reference
    from modeling.sequential.features import SequentialFeatures  # reference

    cfg = CONFIGS["amzn-books"]
    mol = build_reference_module(cfg, seed)
    B, N = 16, 512
    q = synthetic_queries(cfg, B, seed=seed + 2)
    X = torch.from_numpy(hash_item_table(seed + 1, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
    ids = sparse_item_ids(N, seed + 3)

    class StubModel:
        def encode(self, **kw):
            return q

        def get_item_embeddings(self, item_ids):
            raise AssertionError("not used: past_embeddings is computed from a stub")

    stub = StubModel()
    stub.get_item_embeddings = lambda item_ids: torch.zeros(item_ids.shape + (cfg.item_embedding_dim,))
    out = {}
    with torch.inference_mode():
        bf = MoLBruteForceTopK(mol, X, ids)
        state = EvalState(all_item_ids=set(ids.view(-1).tolist()), candidate_index=CandidateIndex(ids=ids, embeddings=X), top_k_module=bf)
        _, top_ids = bf(q, k=300)
        past_ids = make_invalid_ids(top_ids[:, :150], 61, seed + 5)
        # targets: a mix of ranks (hit early, hit late, miss, seen-and-filtered)
        g = torch.Generator().manual_seed(seed + 7)
        pos = torch.tensor([0, 1, 4, 9, 10, 49, 50, 99, 100, 119, 150, 250, 299, 7, 30, 75])
        target_ids = top_ids[torch.arange(B), pos].unsqueeze(1).clone()
        feats = SequentialFeatures(
            past_lengths=torch.full((B,), 61), past_ids=past_ids, past_embeddings=None,
            past_payloads={},
        )
        for mode, timing in (("accuracy", False), ("timing", True)):
            import random
            random.seed(1)  # timing branch samples 10 % of batches
            res = eval_metrics_v2_from_tensors(
                state, stub, feats, target_ids=target_ids, filter_invalid_ids=True,
                include_eval_time=timing, include_eval_top_k_ids=True,
            )
            for k, v in res.items():
                if k == "eval_time":
                    continue
                out[f"F5/{mode}/{k}"] = (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
    out["cfg_json"] = np.array(__import__("json").dumps(cfg.to_dict()))
    for k, v in slim_weights(mol, cfg, None).items():
        out["w/" + k] = v
    out["q"], out["X"], out["item_ids"] = q.numpy(), X.numpy(), ids.numpy()
    out["past_ids"], out["target_ids"] = past_ids.numpy(), target_ids.numpy()
    savez_deterministic(os.path.join(OUT, "harness.npz"), **out)
    print(f"harness: wrote {len(out)} arrays")


def full_size_fixture(name: str, cfg: MoLConfig, N: int, seed: int, B: int = 32, k: int = 200, first_row: bool = True):
    """F7: inputs by recipe (hash table + seeded module), outputs (scores, ids) + logits summary.
    first_row = False (the amzn-books-sized corpus: 695 762 items): without the 2.7 MB row of logits; its B is kept at 8 because the
    reference materialises (B, N, 128) intermediates (11 GB at B = 32)."""
    mol = build_reference_module(cfg, seed)
    q = synthetic_queries(cfg, B, seed=seed + 2)
    X = torch.from_numpy(hash_item_table(seed + 1, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
    ids = sparse_item_ids(N, seed + 3)
    kw = {}
    if len(cfg.uid_embedding_hash_sizes) > 0:
        g = torch.Generator().manual_seed(seed + 4)
        kw["user_ids"] = torch.randint(0, 200000, (B,), generator=g, dtype=torch.int64)
    with torch.inference_mode():
        bf = MoLBruteForceTopK(mol, X, ids)
        logits, _ = mol(q, X, **kw)
        s, i = bf(q, k=k, **kw)
    out = {"cfg_json": np.array(__import__("json").dumps(cfg.to_dict())), "N": np.array(N), "seed": np.array(seed),
           "q": q.numpy(), "item_ids_seed": np.array(seed + 3), "table_seed": np.array(seed + 1)}
    for kk, v in slim_weights(mol, cfg, kw.get("user_ids")).items():
        out["w/" + kk] = v
    if "user_ids" in kw:
        out["user_ids"] = kw["user_ids"].numpy()
    out["scores"], out["ids"] = s.numpy(), i.numpy()
    out["logits_rowsum_f64"] = logits.double().sum(1).numpy()
    out["logits_quantiles"] = torch.quantile(logits, torch.tensor([0.0, 0.01, 0.5, 0.99, 1.0]), dim=1).numpy()
    if first_row:
        out["logits_first_row"] = logits[0].numpy()
    savez_deterministic(os.path.join(OUT, f"{name}.npz"), **out)
    print(f"{name}: wrote {len(out)} arrays")


def mips_fixture(seed: int = 707):
    """F9: MIPSBruteForceTopK and the three branches of DotProductSimilarity."""
    from rails.indexing.mips_top_k import MIPSBruteForceTopK  # reference
    from rails.similarities.dot_product_similarity_fn import DotProductSimilarity  # reference

    out = {}
    g = torch.Generator().manual_seed(seed)
    for tag, D, N, B in (("d50", 50, 1000, 6), ("d64", 64, 20000, 33)):
        q = torch.randn((B, D), generator=g)
        X = torch.from_numpy(hash_item_table(seed + D, 0, N, D, sigma=1.0)).unsqueeze(0)
        ids = sparse_item_ids(N, seed + 3)
        with torch.inference_mode():
            tk = MIPSBruteForceTopK(X, ids)
            dp = DotProductSimilarity()
            logits, aux = dp(q, X)
            assert aux == {}
            # X and ids by recipe (hash_item_table(seed + D, 0, N, D, sigma=1.0), sparse_item_ids(N, seed + 3)); logits: 2 rows
            out[f"{tag}/q"], out[f"{tag}/N"], out[f"{tag}/table_seed"], out[f"{tag}/ids_seed"] = q.numpy(), np.array(N), np.array(seed + D), np.array(seed + 3)
            out[f"{tag}/logits_head"] = logits[:2].numpy()
            for k in (10, 200):
                s, i = tk(q, k=k)
                out[f"{tag}/k{k}/scores"], out[f"{tag}/k{k}/ids"] = s.numpy(), i.numpy()
    # per-row candidates (B, X, D) x (B, D), and the (B*r, D) x (B, X, D) branch
    with torch.inference_mode():
        dp = DotProductSimilarity()
        Xr = torch.randn((4, 37, 24), generator=g)
        q1 = torch.randn((4, 24), generator=g)
        q3 = torch.randn((12, 24), generator=g)
        out["rows/X"], out["rows/q1"], out["rows/q3"] = Xr.numpy(), q1.numpy(), q3.numpy()
        out["rows/out1"] = dp(q1, Xr)[0].numpy()
        out["rows/out3"] = dp(q3, Xr)[0].numpy()
    savez_deterministic(os.path.join(OUT, "mips.npz"), **out)
    print(f"mips: wrote {len(out)} arrays")


def union_fixture(seed: int = 808):
    """F10: MoLNaiveTopK and MoLCombTopK.  The candidate union the reference built is
    captured by spying on its torch.sort call, so the rerank half can be checked exactly on the same candidates."""
    from rails.indexing.mol_top_k import MoLCombTopK, MoLNaiveTopK  # reference

    out = {}
    for cname, cfg, s0 in (("c1", CONFIGS["ml-1m"], seed), ("c3", CONFIGS["amzn-books"], seed + 50)):
        mol = build_reference_module(cfg, s0)
        B, N = 5, 1024
        q = synthetic_queries(cfg, B, seed=s0 + 2)
        X = torch.from_numpy(hash_item_table(s0 + 1, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
        ids = sparse_item_ids(N, s0 + 3)
        kw = {}
        if len(cfg.uid_embedding_hash_sizes) > 0:
            g = torch.Generator().manual_seed(s0 + 4)
            kw["user_ids"] = torch.randint(0, 200000, (B,), generator=g, dtype=torch.int64)
            out[f"{cname}/user_ids"] = kw["user_ids"].numpy()
        out[f"{cname}/cfg_json"] = np.array(__import__("json").dumps(cfg.to_dict()))
        for k, v in slim_weights(mol, cfg, kw.get("user_ids")).items():
            out[f"{cname}/w/" + k] = v
        out[f"{cname}/q"], out[f"{cname}/X"], out[f"{cname}/item_ids"] = q.numpy(), X.numpy(), ids.numpy()
        mods = {"naive5": lambda: MoLNaiveTopK(mol, X, ids, k_per_group=5),
                "comb5_100": lambda: MoLCombTopK(mol, X, ids, avg_top_k=100, k_per_group=5)}
        for mname, make in mods.items():
            rec = {}
            orig_sort = torch.sort

            def spy(*a, **k):
                r = orig_sort(*a, **k)
                rec["sorted"] = r[0].clone()
                return r

            with torch.inference_mode():
                mod = make()
                torch.sort = spy
                try:
                    s, i = mod(q, k=10, sorted=True, **kw)   # k is ignored by the reference: all candidates come back
                finally:
                    torch.sort = orig_sort
            out[f"{cname}/{mname}/scores"], out[f"{cname}/{mname}/ids"] = s.numpy(), i.numpy()
            out[f"{cname}/{mname}/sorted_all_indices"] = rec["sorted"].numpy()
    savez_deterministic(os.path.join(OUT, "union.npz"), **out)
    print(f"union: wrote {len(out)} arrays")


ALL = {
    "c1_ml1m": lambda: per_config_fixture("c1_ml1m", CONFIGS["ml-1m"], seed=101),
    "c2_ml20m": lambda: per_config_fixture("c2_ml20m", CONFIGS["ml-20m"], seed=202),
    "c3_books": lambda: per_config_fixture("c3_books", CONFIGS["amzn-books"], seed=303),
    "c4_16x16x64": lambda: per_config_fixture("c4_16x16x64", CONFIGS["synthetic-16x16x64"], seed=404, B=5, N=512, n_stage=96),
    "harness": harness_fixture,
    "full_c1_ml1m": lambda: full_size_fixture("full_c1_ml1m", CONFIGS["ml-1m"], N=3883, seed=505),
    "full_c2_ml20m": lambda: full_size_fixture("full_c2_ml20m", CONFIGS["ml-20m"], N=27278, seed=606),
    "full_c3_books": lambda: full_size_fixture("full_c3_books", CONFIGS["amzn-books"], N=695762, seed=808, B=8, first_row=False),
    "mips": mips_fixture,
    "union": union_fixture,
}

if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    for name in (sys.argv[1:] or list(ALL)):   # `python oracle/gen_golden.py mips` regenerates one fixture
        ALL[name]()
