"""CPU oracle for the Mixture-of-Logits (MoL) retrieval hot path.

TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path: only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
Nothing under `rails_amd/` imports, calls or falls back to anything in `oracle/`.

What it is: a plain torch-CPU fp32 restatement, op for op and in the same order, of the
reference's eval-mode algorithm for the path (citations are `path:line` inside the
reference repo bailuding/rails @ 2025-02-12):

  step 1  query component embeddings  rails/similarities/mol/query_embeddings_fns.py:175-254
                                      (GLU: rails/similarities/layers.py:19-74)
  step 2  item component embeddings   rails/similarities/mol/item_embeddings_fns.py:149-183
  step 3  item-only gate              modeling/similarity_utils.py:169-185, similarity_fn.py:170-171
  step 4  query-only gate             modeling/similarity_utils.py:153-168, similarity_fn.py:166-169
  step 5  cross logits / temperature  rails/similarities/mol/similarity_fn.py:389-405
  step 6  pair gate MLP               modeling/similarity_utils.py:186-207, similarity_fn.py:172-173
  step 7  glu_silu combination        rails/similarities/mol/similarity_fn.py:175-179
  step 8  softmax + eval-time renorm  rails/similarities/mol/similarity_fn.py:31-46
  step 9  exact top-k                 rails/indexing/mol_top_k.py:99-130
  step 10 seen-id filter              indexing/candidate_index.py:116-185
  two-pass approximate top-k          rails/indexing/mol_top_k.py:296-429
  rank / HR / NDCG / MRR              data/eval.py:194-243

How it is pinned: the reference has no tests, so parity is pinned by golden
vectors produced by importing the reference itself in the build container
(`oracle/gen_golden.py`, fixtures under `tests/golden/`); `tests/test_oracle_golden.py`
checks every function here against them.

Weights are passed as a flat dict keyed by the reference's `state_dict()` names
(SURVEY.md §8b), values torch tensors or numpy arrays.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclasses.dataclass
class MoLConfig:
    """Field names follow `create_mol_interaction_module` (modeling/similarity_utils.py:42-70)."""

    query_embedding_dim: int
    item_embedding_dim: int
    dot_product_dimension: int
    query_dot_product_groups: int
    item_dot_product_groups: int
    temperature: float = 0.05
    query_hidden_dim: int = 512
    item_hidden_dim: int = -1
    gating_query_hidden_dim: int = 128
    gating_qi_hidden_dim: int = 128
    gating_item_hidden_dim: int = 128
    softmax_dropout_rate: float = 0.2
    query_nonlinearity: str = "geglu"
    item_nonlinearity: str = "geglu"
    uid_embedding_hash_sizes: Sequence[int] = ()
    gating_combination_type: str = "glu_silu"
    dot_product_l2_norm: bool = True
    eps: float = 1e-6
    gating_query_fn: bool = True    # modeling/similarity_utils.py:147-161: False -> no query-only gate part
    gating_item_fn: bool = True     # modeling/similarity_utils.py:162-179: False -> no item-only gate part

    @property
    def num_logits(self) -> int:
        return self.query_dot_product_groups * self.item_dot_product_groups

    def to_dict(self) -> dict:
        d = dataclasses.asdict(self)
        d["uid_embedding_hash_sizes"] = list(self.uid_embedding_hash_sizes)
        return d


# The five shapes BASELINE.json names.
CONFIGS: Dict[str, MoLConfig] = {
    "ml-1m": MoLConfig(50, 50, 64, 8, 4, query_nonlinearity="swiglu", uid_embedding_hash_sizes=(6040,)),
    "ml-20m": MoLConfig(256, 256, 128, 8, 4, query_nonlinearity="swiglu", uid_embedding_hash_sizes=(16384,)),
    "amzn-books": MoLConfig(64, 64, 32, 8, 8),
    "synthetic-16x16x64": MoLConfig(64, 64, 64, 16, 16),
    "synthetic-8x8x32": MoLConfig(64, 64, 32, 8, 8),
}


def _t(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        return x.detach().to("cpu")
    return torch.from_numpy(np.asarray(x))


def _weights(w: Dict[str, object]) -> Dict[str, torch.Tensor]:
    return {k: _t(v) for k, v in w.items()}


def _l2norm(x: torch.Tensor, eps: float) -> torch.Tensor:
    # query_embeddings_fns.py:244-253 / item_embeddings_fns.py:173-182
    return x / torch.clamp(torch.linalg.norm(x, ord=None, dim=-1, keepdim=True), min=eps)


def _glu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, kind: str) -> torch.Tensor:
    # rails/similarities/layers.py:36-43 (GeGLU) / :66-73 (SwiGLU)
    out_features = w.shape[1] // 2
    bs = x.size()[:-1]
    lhs, rhs = torch.split(torch.mm(x.reshape(-1, w.shape[0]), w) + b, [out_features, out_features], dim=-1)
    act = F.gelu(lhs) if kind == "geglu" else F.silu(lhs)
    return (act * rhs).reshape(bs + (out_features,))


def query_component_embeddings(
    cfg: MoLConfig, w: Dict[str, object], q: torch.Tensor, user_ids: Optional[torch.Tensor] = None
) -> torch.Tensor:
    """Step 1 -> (B, P_Q, d).  query_embeddings_fns.py:175-254."""
    w = _weights(w)
    q = _t(q)
    pre = "_query_embeddings_fn._query_emb_proj_module."
    n_uid = len(cfg.uid_embedding_hash_sizes)
    groups = cfg.query_dot_product_groups - n_uid
    if cfg.query_hidden_dim > 0:
        h = _glu(q, w[pre + "1._w"], w[pre + "1._b"], cfg.query_nonlinearity)
        proj = F.linear(h, w[pre + "2.weight"], w[pre + "2.bias"])
    else:
        proj = F.linear(q, w[pre + "1.weight"], w[pre + "1.bias"])
    split = proj.reshape(q.size(0), groups, cfg.dot_product_dimension)
    if n_uid > 0:
        assert user_ids is not None, "user_ids is required when uid_embedding_hash_sizes is set"
        user_ids = _t(user_ids)
        uid_embs = []
        for i, hash_size in enumerate(cfg.uid_embedding_hash_sizes):
            table = w[f"_query_embeddings_fn._uid_embeddings_{i}.weight"]
            uid_embs.append(F.embedding((user_ids % hash_size) + 1, table).unsqueeze(1))
        split = torch.cat([split] + uid_embs, dim=1)
    if cfg.dot_product_l2_norm:
        split = _l2norm(split, cfg.eps)
    return split


def item_component_embeddings(cfg: MoLConfig, w: Dict[str, object], x: torch.Tensor) -> torch.Tensor:
    """Step 2 -> (..., P_X, d).  item_embeddings_fns.py:149-183."""
    w = _weights(w)
    x = _t(x)
    pre = "_item_embeddings_fn._item_emb_proj_module."
    if cfg.item_hidden_dim > 0:
        h = _glu(x, w[pre + "1._w"], w[pre + "1._b"], cfg.item_nonlinearity)
        proj = F.linear(h, w[pre + "2.weight"], w[pre + "2.bias"])
    else:
        proj = F.linear(x, w[pre + "1.weight"], w[pre + "1.bias"])
    split = proj.reshape(x.size()[:-1] + (cfg.item_dot_product_groups, cfg.dot_product_dimension))
    if cfg.dot_product_l2_norm:
        split = _l2norm(split, cfg.eps)
    return split


def item_gate(cfg: MoLConfig, w: Dict[str, object], x: torch.Tensor) -> torch.Tensor:
    """Step 3 -> (..., L).  Sequential(Dropout, Linear, SiLU, Linear(no bias)); an absent part adds nothing
    (similarity_fn.py:187-197, combination "none")."""
    if not cfg.gating_item_fn:
        return torch.zeros(_t(x).shape[:-1] + (cfg.num_logits,), dtype=torch.float32)
    w = _weights(w)
    pre = "_gating_fn._item_only_partial_module."
    h = F.silu(F.linear(_t(x), w[pre + "1.weight"], w[pre + "1.bias"]))
    return F.linear(h, w[pre + "3.weight"])


def query_gate(cfg: MoLConfig, w: Dict[str, object], q: torch.Tensor) -> torch.Tensor:
    """Step 4 -> (B, L) from the RAW query embedding.  Sequential(Linear, SiLU, Linear(no bias)); absent -> zeros."""
    if not cfg.gating_query_fn:
        return torch.zeros((_t(q).shape[0], cfg.num_logits), dtype=torch.float32)
    w = _weights(w)
    pre = "_gating_fn._query_only_partial_module."
    h = F.silu(F.linear(_t(q), w[pre + "0.weight"], w[pre + "0.bias"]))
    return F.linear(h, w[pre + "2.weight"])


def pair_gate(cfg: MoLConfig, w: Dict[str, object], cl: torch.Tensor) -> torch.Tensor:
    """Step 6 -> (B, X, L).  Sequential(Dropout, Linear(L,H), SiLU, Linear(H,L)) or a single Linear."""
    w = _weights(w)
    pre = "_gating_fn._qi_partial_module."
    if cfg.gating_qi_hidden_dim > 0:
        h = F.silu(F.linear(cl, w[pre + "1.weight"], w[pre + "1.bias"]))
        return F.linear(h, w[pre + "3.weight"], w[pre + "3.bias"])
    return F.linear(cl, w[pre + "1.weight"], w[pre + "1.bias"])


def combine(cfg: MoLConfig, gq: torch.Tensor, gi: torch.Tensor, gqi: torch.Tensor) -> torch.Tensor:
    """Step 7.  similarity_fn.py:175-199 (only glu_silu and none are reachable)."""
    if cfg.gating_combination_type == "glu_silu":
        g = gq * gi + gqi
        return g * torch.sigmoid(g)
    if cfg.gating_combination_type == "none":   # absent parts are zeros here (query_gate / item_gate)
        return gq + gi + gqi
    raise ValueError(f"Unknown combination_type {cfg.gating_combination_type}")


def mixture(cfg: MoLConfig, gw: torch.Tensor, cl: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Step 8 -> (pi, logits).  similarity_fn.py:42-46: the renormalisation runs in eval too."""
    pi = F.softmax(gw, dim=-1)
    if cfg.softmax_dropout_rate > 0.0:
        # F.dropout(training=False) is the identity
        pi = pi / torch.clamp(pi.sum(-1, keepdim=True), min=1e-6)  # combiner eps is hard-coded 1e-6
    return pi, (pi * cl).sum(-1)


def mol_stages(
    cfg: MoLConfig,
    w: Dict[str, object],
    q: torch.Tensor,
    items: torch.Tensor,
    user_ids: Optional[torch.Tensor] = None,
) -> Dict[str, torch.Tensor]:
    """Every intermediate of MoLSimilarity.forward (similarity_fn.py:341-413) for one un-chunked call.

    `items` is (1, X, D) (shared corpus) or (B, X, D) (per-row candidates, the `B'==B` branch).
    """
    w = _weights(w)
    q = _t(q).float()
    items = _t(items).float()
    B = q.size(0)
    Bp, X = items.shape[0], items.shape[1]
    L = cfg.num_logits
    eq = query_component_embeddings(cfg, w, q, user_ids)
    ex = item_component_embeddings(cfg, w, items)
    if Bp == 1:
        cl = torch.einsum("bnd,xmd->bxnm", eq, ex.squeeze(0)).reshape(B, X, L)
    else:
        cl = torch.einsum("bnd,bxmd->bxnm", eq, ex).reshape(B, X, L)
    cl = cl / cfg.temperature
    gq = query_gate(cfg, w, q).unsqueeze(1)
    gi = item_gate(cfg, w, items)
    gqi = pair_gate(cfg, w, cl)
    gw = combine(cfg, gq, gi, gqi)
    pi, logits = mixture(cfg, gw, cl)
    return {"Eq": eq, "Ex": ex, "gq": gq.squeeze(1), "gi": gi, "cl": cl, "gqi": gqi, "w": gw, "pi": pi, "logits": logits}


def mol_logits(
    cfg: MoLConfig,
    w: Dict[str, object],
    q: torch.Tensor,
    items: torch.Tensor,
    user_ids: Optional[torch.Tensor] = None,
    chunk: int = 8192,
) -> torch.Tensor:
    """(B, X) logits, chunked over X so the (B, X, L)/(B, X, H) intermediates stay bounded.

    Per-item arithmetic is identical to the un-chunked call: every op on the path is row-wise in X.
    """
    w = _weights(w)
    q = _t(q).float()
    items = _t(items).float()
    X = items.shape[1]
    out = torch.empty((q.size(0), X), dtype=torch.float32)
    for s in range(0, X, chunk):
        out[:, s : s + chunk] = mol_stages(cfg, w, q, items[:, s : s + chunk], user_ids)["logits"]
    return out


def brute_force_topk(
    cfg: MoLConfig,
    w: Dict[str, object],
    q: torch.Tensor,
    items: torch.Tensor,
    item_ids: torch.Tensor,
    k: int,
    user_ids: Optional[torch.Tensor] = None,
    chunk: int = 8192,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """MoLBruteForceTopK.forward (mol_top_k.py:99-130) -> (scores, ids, all_logits)."""
    logits = mol_logits(cfg, w, q, items, user_ids, chunk)
    s, idx = torch.topk(logits, dim=1, k=k, sorted=True, largest=True)
    return s, _t(item_ids).reshape(-1)[idx], logits


def select_topk_deterministic(scores: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact top-k under the build's tie rule (score desc, then position asc) — what the HIP
    selection kernels implement.  Equal to torch.topk wherever scores are distinct."""
    scores = _t(scores)
    order = torch.argsort(scores, dim=1, descending=True, stable=True)[:, :k]
    return torch.gather(scores, 1, order), order


def filter_seen_ids(
    top_ids: torch.Tensor, top_scores: torch.Tensor, invalid_ids: Optional[torch.Tensor], k: int
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-wise seen-id filter of CandidateIndex.get_top_k_outputs (indexing/candidate_index.py:154-178)."""
    top_ids, top_scores = _t(top_ids), _t(top_scores)
    if invalid_ids is None:
        return top_ids, top_scores
    invalid_ids = _t(invalid_ids)
    is_seen = (top_ids.unsqueeze(2) == invalid_ids.unsqueeze(1)).max(2)[0]
    valid = ~is_seen
    valid = torch.logical_and(valid, torch.cumsum(valid.int(), dim=1) <= k)
    inval = ~valid
    gap = k - valid.int().sum(1, keepdim=True)
    valid = torch.logical_or(valid, torch.logical_and(inval, torch.cumsum(inval.int(), dim=1) <= gap))
    offs = torch.nonzero(valid, as_tuple=True)[1].view(-1, k)
    return torch.gather(top_ids, 1, offs), torch.gather(top_scores, 1, offs)


def k_prime(k: int, invalid_ids: Optional[torch.Tensor], num_objects: int, truncate_k_prime_to: Optional[int]) -> int:
    """indexing/candidate_index.py:144-151."""
    width = 0 if invalid_ids is None else int(invalid_ids.shape[1])
    kp = min(k + width, num_objects)
    if truncate_k_prime_to is not None:
        kp = min(kp, truncate_k_prime_to)
    return kp


def avg_topk_coarse_scores(cfg: MoLConfig, w, q, items, user_ids=None, table_dtype=torch.bfloat16):
    """Pass 1 of MoLAvgTopK.forward (mol_top_k.py:321-325, 350-354): (B, d) x (d, N) in the table dtype."""
    eq = query_component_embeddings(cfg, w, _t(q).float(), user_ids)
    ex = item_component_embeddings(cfg, w, _t(items).float().squeeze(0)).to(table_dtype)
    table_t = (ex.sum(1) / cfg.item_dot_product_groups).transpose(0, 1)
    return torch.mm(eq.sum(1).to(table_dtype), table_t)


def avg_topk(cfg: MoLConfig, w, q, items, item_ids, k: int, avg_top_k: int, user_ids=None, coarse_idx=None):
    """MoLAvgTopK.forward (mol_top_k.py:328-396).  `coarse_idx` lets a caller supply pass-1's
    candidate set (the bf16 coarse scores tie heavily, so their top-k order is implementation-defined)."""
    if k > avg_top_k:
        raise ValueError(f"avg_top_k ({avg_top_k}) must be larger than k ({k})")
    items = _t(items).float()
    if coarse_idx is None:
        coarse = avg_topk_coarse_scores(cfg, w, q, items, user_ids)
        _, coarse_idx = torch.topk(coarse, k=avg_top_k, dim=1, sorted=False)
    coarse_idx = _t(coarse_idx)
    B = coarse_idx.size(0)
    cand = items.squeeze(0)[coarse_idx].view(B, avg_top_k, -1)
    cand_scores = mol_stages(cfg, w, q, cand, user_ids)["logits"]
    s, idx = torch.topk(cand_scores, k=min(k, avg_top_k), dim=1, largest=True, sorted=True)
    return s, _t(item_ids).reshape(-1)[torch.gather(coarse_idx, 1, idx)], coarse_idx


def union_rerank(cfg: MoLConfig, w, q, items, item_ids, sorted_all_indices, user_ids=None):
    """Second half of MoLNaiveTopK / MoLCombTopK.forward (rails/indexing/mol_top_k.py:259-293, :517-551): MoL on the
    sorted candidate union, duplicates masked with -32767.0, top-k over ALL candidates."""
    idx = _t(sorted_all_indices)
    B, k = idx.shape
    items = _t(items).float().squeeze(0)
    cand = items[idx.view(-1)].reshape(B, k, -1)
    scores = mol_stages(cfg, w, q, cand, user_ids)["logits"]
    valid = torch.cat([torch.ones_like(idx[:, 0:1], dtype=torch.bool), idx[:, 1:] != idx[:, :-1]], dim=1)
    scores = torch.where(valid, scores, -32767.0)
    s, ti = torch.topk(scores, k=k, dim=1, largest=True, sorted=True)
    return s, _t(item_ids).reshape(-1)[torch.gather(idx, 1, ti)]


def component_candidate_scores(cfg: MoLConfig, w, q, items, user_ids=None) -> torch.Tensor:
    """First half (mol_top_k.py:242-251): bf16 mm of every query group against every item group's bf16 components.
    -> (B, P_Q, P_X, N) bf16."""
    eq = query_component_embeddings(cfg, w, _t(q).float(), user_ids).to(torch.bfloat16)
    ex = item_component_embeddings(cfg, w, _t(items).float().squeeze(0)).to(torch.bfloat16)  # (N, P_X, d)
    N, PX, d = ex.shape
    table_t = ex.permute(1, 0, 2).reshape(-1, d).transpose(0, 1)  # (d, P_X * N)
    outs = [torch.mm(eq[:, i, :], table_t).view(eq.size(0), PX, N) for i in range(eq.size(1))]
    return torch.stack(outs, dim=1)


def dot_product_similarity(q: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
    """DotProductSimilarity.forward (rails/similarities/dot_product_similarity_fn.py:33-68), all three branches."""
    q, items = _t(q), _t(items)
    B_I, X, D = items.size()
    if B_I == 1:
        return torch.mm(q, items.squeeze(0).t())
    if q.size(0) != B_I:
        return torch.bmm(q.view(B_I, -1, D), items.permute(0, 2, 1)).view(-1, X)
    return torch.bmm(items, q.unsqueeze(2)).squeeze(2)


def mips_brute_force_topk(q, items, item_ids, k: int):
    """MIPSBruteForceTopK.forward (rails/indexing/mips_top_k.py:56-81)."""
    logits = torch.mm(_t(q), _t(items).permute(2, 1, 0).squeeze(2))
    s, idx = torch.topk(logits, dim=1, k=k, sorted=True, largest=True)
    return s, _t(item_ids).reshape(-1)[idx]


def eval_ranks(top_k_ids: torch.Tensor, target_ids: torch.Tensor, max_k: int) -> torch.Tensor:
    """data/eval.py:194-201: 1-based rank of the target inside the returned ids, MAX_K+1 if absent."""
    top_k_ids, target_ids = _t(top_k_ids), _t(target_ids)
    k = top_k_ids.size(1)
    _, idx = torch.max(torch.cat([top_k_ids, target_ids], dim=1) == target_ids, dim=1)
    return torch.where(idx == k, max_k + 1, idx + 1)


def eval_metrics(top_k_ids: torch.Tensor, target_ids: torch.Tensor, max_k: int) -> Dict[str, torch.Tensor]:
    """data/eval.py:203-243 (hr@k, ndcg@k, mrr per example)."""
    r = eval_ranks(top_k_ids, target_ids, max_k)
    out: Dict[str, torch.Tensor] = {"rank": r}
    for kk in (1, 5, 10, 50, 100, 200):
        out[f"ndcg@{kk}"] = torch.where(r <= kk, 1.0 / torch.log2(r + 1), torch.zeros(1, dtype=torch.float32))
    for kk in (1, 5, 10, 50, 100, 200, 500, 1000):
        out[f"hr@{kk}"] = r <= kk
    out["mrr"] = 1.0 / r
    return out


# ---------------------------------------------------------------------------------------------
# Seeded synthetic inputs shared by tests and bench.  Pure integer hashing +
# exact float ops, so every platform (and the HIP generator kernel) produces identical bits.
# ---------------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def hash_item_table(seed: int, first_item: int, n_items: int, dim: int, sigma: float = 0.02) -> np.ndarray:
    """Counter-based item-embedding generator: row i depends only on (seed, i), so any shard or
    CPU sub-range is reproducible without materialising the table.  Each value is an Irwin-Hall(4)
    sum of 16-bit lanes of one splitmix64 word, centred and scaled to std `sigma` (|x| <= 3.47 sigma);
    it stands in for the reference's truncated normal (modeling/initialization.py:18-26)."""
    i = (np.arange(first_item, first_item + n_items, dtype=np.uint64)[:, None] * np.uint64(dim)
         + np.arange(dim, dtype=np.uint64)[None, :])
    with np.errstate(over="ignore"):
        h = _splitmix64(i ^ (np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) & _M64))
    s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))).astype(np.int64)
    centred = (s - 2 * 65535).astype(np.float32)  # exact: |.| < 2^18
    scale = np.float32(sigma * math.sqrt(3.0) / 65536.0)
    return centred * scale


def hash_item_rows(seed: int, items: np.ndarray, dim: int, sigma: float = 0.02) -> np.ndarray:
    """Rows `items` (any order, any subset) of the same table: what lets a test re-create on the CPU, by id, the handful of rows of a
    100 M-item shard it wants to check."""
    items = np.asarray(items, dtype=np.uint64)
    i = items[:, None] * np.uint64(dim) + np.arange(dim, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        h = _splitmix64(i ^ (np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) & _M64))
    s = ((h & np.uint64(0xFFFF)) + ((h >> np.uint64(16)) & np.uint64(0xFFFF))
         + ((h >> np.uint64(32)) & np.uint64(0xFFFF)) + (h >> np.uint64(48))).astype(np.int64)
    return (s - 2 * 65535).astype(np.float32) * np.float32(sigma * math.sqrt(3.0) / 65536.0)


def synthetic_weights(cfg: MoLConfig, seed: int = 0, uid_rows: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Random-init MoL weights with the reference's initialisers (modeling/similarity_utils.py:34-38,
    rails/similarities/layers.py:29-34, torch.nn.Embedding default)."""
    g = torch.Generator().manual_seed(seed)

    def xavier(out_f, in_f):
        a = math.sqrt(6.0 / (in_f + out_f))
        return (torch.rand((out_f, in_f), generator=g) * 2 - 1) * a

    def kaiming_linear(out_f, in_f):
        a = 1.0 / math.sqrt(in_f)
        return (torch.rand((out_f, in_f), generator=g) * 2 - 1) * a, (torch.rand((out_f,), generator=g) * 2 - 1) * a

    D, Di, d = cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension
    L, n_uid = cfg.num_logits, len(cfg.uid_embedding_hash_sizes)
    w: Dict[str, torch.Tensor] = {}
    if cfg.gating_query_fn:
        p = "_gating_fn._query_only_partial_module."
        w[p + "0.weight"], w[p + "0.bias"] = xavier(cfg.gating_query_hidden_dim, D), torch.zeros(cfg.gating_query_hidden_dim)
        w[p + "2.weight"] = xavier(L, cfg.gating_query_hidden_dim)
    if cfg.gating_item_fn:
        p = "_gating_fn._item_only_partial_module."
        w[p + "1.weight"], w[p + "1.bias"] = xavier(cfg.gating_item_hidden_dim, Di), torch.zeros(cfg.gating_item_hidden_dim)
        w[p + "3.weight"] = xavier(L, cfg.gating_item_hidden_dim)
    p = "_gating_fn._qi_partial_module."
    if cfg.gating_qi_hidden_dim > 0:
        w[p + "1.weight"], w[p + "1.bias"] = xavier(cfg.gating_qi_hidden_dim, L), torch.zeros(cfg.gating_qi_hidden_dim)
        w[p + "3.weight"], w[p + "3.bias"] = xavier(L, cfg.gating_qi_hidden_dim), torch.zeros(L)
    else:   # one Linear(L, L) (modeling/similarity_utils.py:199-206)
        w[p + "1.weight"], w[p + "1.bias"] = xavier(L, L), torch.zeros(L)
    p = "_query_embeddings_fn._query_emb_proj_module."
    if cfg.query_hidden_dim > 0:
        w[p + "1._w"] = torch.randn((D, 2 * cfg.query_hidden_dim), generator=g) * 0.02
        w[p + "1._b"] = torch.zeros((1, 2 * cfg.query_hidden_dim))
        w[p + "2.weight"], w[p + "2.bias"] = kaiming_linear(d * (cfg.query_dot_product_groups - n_uid), cfg.query_hidden_dim)
    else:   # plain Linear, xavier + zero bias (modeling/similarity_utils.py:108-116)
        w[p + "1.weight"], w[p + "1.bias"] = xavier(d * (cfg.query_dot_product_groups - n_uid), D), torch.zeros(d * (cfg.query_dot_product_groups - n_uid))
    for i, hs in enumerate(cfg.uid_embedding_hash_sizes):
        rows = hs + 1 if uid_rows is None else uid_rows
        t = torch.randn((rows, d), generator=g)
        t[0] = 0.0  # padding_idx=0
        w[f"_query_embeddings_fn._uid_embeddings_{i}.weight"] = t
    p = "_item_embeddings_fn._item_emb_proj_module."
    if cfg.item_hidden_dim > 0:   # GLU (N(0, 0.02^2), zero bias) then Linear (xavier, zero bias) (similarity_utils.py:127-143)
        w[p + "1._w"] = torch.randn((Di, 2 * cfg.item_hidden_dim), generator=g) * 0.02
        w[p + "1._b"] = torch.zeros((1, 2 * cfg.item_hidden_dim))
        w[p + "2.weight"], w[p + "2.bias"] = xavier(d * cfg.item_dot_product_groups, cfg.item_hidden_dim), torch.zeros(d * cfg.item_dot_product_groups)
    else:
        w[p + "1.weight"], w[p + "1.bias"] = xavier(d * cfg.item_dot_product_groups, Di), torch.zeros(d * cfg.item_dot_product_groups)
    return w


def synthetic_queries(cfg: MoLConfig, batch: int, seed: int = 2) -> torch.Tensor:
    """LayerNorm'd Gaussian queries: what LayerNormEmbeddingPostprocessor emits
    (modeling/sequential/output_postprocessors.py:76-85)."""
    g = torch.Generator().manual_seed(seed)
    return F.layer_norm(torch.randn((batch, cfg.query_embedding_dim), generator=g), (cfg.query_embedding_dim,))


def int8_prefilter_bound(q_bf16: torch.Tensor, table_bf16: torch.Tensor):
    """CPU restatement of the int8 pre-filter of rails_amd's fused coarse top-K' (rails_amd/csrc/mol_coarse.hip: prefilter_*_kernel,
    quantise_query, the preamble of coarse_scan_i8_kernel) -- no counterpart in the reference; test infrastructure for the claim
    that the integer test cannot lose an item the bf16 scan keeps.  q (B, d), table (N, d): bf16 values.
    -> (I (B, N) int64 integer dot products, eps (B,), s (scalar), s_q (B,)) with  |q . x - s s_q I| <= eps  for every pair."""
    q = q_bf16.float()
    x = table_bf16.float()
    d = x.shape[1]
    f32 = torch.float32
    mx = x.abs().max()
    inv = (torch.tensor(127.0, dtype=f32) / mx) if mx > 0 else torch.tensor(1.0)
    s = (mx / 127.0) if mx > 0 else torch.tensor(1.0)
    xi = torch.clamp(torch.round(x * inv), -127, 127).to(torch.int64)            # rintf = round half to even, as torch.round
    x1max = x.abs().sum(1).max()
    qmx = q.abs().amax(1)
    qinv = torch.where(qmx > 0, 127.0 / qmx, torch.ones_like(qmx))
    s_q = torch.where(qmx > 0, qmx / 127.0, torch.ones_like(qmx))
    qi = torch.clamp(torch.round(q * qinv[:, None]), -127, 127).to(torch.int64)
    l1 = q.abs().sum(1)
    eps = 1.002 * (0.5 * s * l1 + 0.5 * s_q * x1max + 0.75 * s * s_q * d)
    return qi @ xi.T, eps, s, s_q
