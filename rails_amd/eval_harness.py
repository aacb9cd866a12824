"""Eval harness entry points of the reference's data/eval.py, bound to the HIP top-k modules.

  EvalState, get_eval_state            reference data/eval.py:41-73
  eval_metrics_v2_from_tensors         reference data/eval.py:76-268
  _avg, add_to_summary_writer          reference data/eval.py:271-292

The retrieval call (`CandidateIndex.get_top_k_outputs`) runs in HIP; what is left here is the reference's protocol:
k / k' / truncate rules (data/eval.py:128-130), the timing protocol (10 % of mini-batches, 3 warm-ups + 20 timed calls,
data/eval.py:139-170 -- with an explicit device sync, which the reference gets implicitly from `nonzero`), and the rank /
HR / NDCG / MRR bookkeeping on the returned ids (integer compares on (B, k) tensors).  The encoder (`model.encode`,
`model.get_item_embeddings`) is whatever the caller passes: it is upstream of the path and out of scope.
"""
from __future__ import annotations

import random
import time
from dataclasses import dataclass
from typing import Callable, Dict, List, NamedTuple, Optional, Set

import torch
import torch.distributed as dist

from .topk_modules import CandidateIndex, TopKModule


class SequentialFeatures(NamedTuple):
    """Reference modeling/sequential/features.py:23-32."""

    past_lengths: torch.Tensor
    past_ids: torch.Tensor
    past_embeddings: Optional[torch.Tensor]
    past_payloads: Dict[str, torch.Tensor]


@dataclass
class EvalState:
    all_item_ids: Set[int]
    candidate_index: CandidateIndex
    top_k_module: TopKModule


@torch.inference_mode()
def get_eval_state(
    model,
    all_item_ids: List[int],
    negatives_sampler,
    top_k_module_fn: Callable[[torch.Tensor, torch.Tensor], TopKModule],
    device: torch.device,
    float_dtype: Optional[torch.dtype] = None,
) -> EvalState:
    eval_negatives_ids = torch.as_tensor(all_item_ids).to(device).unsqueeze(0)  # [1, X]
    emb = model.get_item_embeddings(eval_negatives_ids)
    if negatives_sampler is not None:
        emb = negatives_sampler.normalize_embeddings(emb)
    if float_dtype is not None:
        emb = emb.to(float_dtype)
    return EvalState(
        all_item_ids=set(all_item_ids),
        candidate_index=CandidateIndex(ids=eval_negatives_ids, embeddings=emb),
        top_k_module=top_k_module_fn(emb, eval_negatives_ids),
    )


def _sync(t: torch.Tensor) -> None:
    if t.is_cuda:
        torch.cuda.synchronize(t.device)


@torch.inference_mode()
def eval_metrics_v2_from_tensors(
    eval_state: EvalState,
    model,
    seq_features: SequentialFeatures,
    target_ids: torch.Tensor,  # [B, 1]
    min_positive_rating: int = 4,
    target_ratings: Optional[torch.Tensor] = None,  # [B, 1]
    epoch: Optional[str] = None,
    include_full_matrices: bool = False,
    filter_invalid_ids: bool = True,
    user_max_batch_size: Optional[int] = None,
    dtype: Optional[torch.dtype] = None,
    include_eval_time: bool = False,
    include_eval_top_k_ids: bool = False,
) -> Dict[str, torch.Tensor]:
    if include_full_matrices:
        raise NotImplementedError("include_full_matrices is not supported (the (B, N) logit matrix is internal to the top-k modules)")
    device = target_ids.device
    q = model.encode(
        past_lengths=seq_features.past_lengths,
        past_ids=seq_features.past_ids,
        past_embeddings=model.get_item_embeddings(seq_features.past_ids),
        past_payloads=seq_features.past_payloads,
    )
    if dtype is not None:
        q = q.to(dtype)

    MAX_K = 120 if include_eval_time else 2500
    truncate_k_prime_to = 200 if include_eval_time else None
    k = min(MAX_K, eval_state.candidate_index.ids.size(1))
    user_max_batch_size = user_max_batch_size or q.size(0)
    num_batches = (q.size(0) + user_max_batch_size - 1) // user_max_batch_size
    ids_all, prs_all, eval_time_all = [], [], []

    n_rows = q.size(0)

    def call(mb: int):
        sl = slice(mb * user_max_batch_size, (mb + 1) * user_max_batch_size)
        # per-row payloads (user_ids, ...) travel with their rows; the reference passes them unsliced, which only works
        # when there is a single mini-batch (data/eval.py:143-152)
        payloads = {key: (v[sl] if torch.is_tensor(v) and v.dim() >= 1 and v.size(0) == n_rows else v)
                    for key, v in seq_features.past_payloads.items()}
        return eval_state.candidate_index.get_top_k_outputs(
            query_embeddings=q[sl, ...],
            top_k_module=eval_state.top_k_module,
            k=k,
            aux_payloads=payloads,
            invalid_ids=seq_features.past_ids[sl, :] if filter_invalid_ids else None,
            return_embeddings=False,
            truncate_k_prime_to=truncate_k_prime_to,
        )

    for mb in range(num_batches):
        if include_eval_time and random.random() < 0.1:  # time 10 % of the mini-batches: 3 warm-ups, 20 timed calls
            for _ in range(3):
                call(mb)
            _sync(q)
            start = time.time()
            for _ in range(20):
                call(mb)
            _sync(q)
            eval_time_all.append((time.time() - start) / 20)
        top_ids, top_prs, _ = call(mb)
        ids_all.append(top_ids)
        prs_all.append(top_prs)
    eval_top_k_ids = ids_all[0] if num_batches == 1 else torch.cat(ids_all, dim=0)

    assert eval_top_k_ids.size(1) == k
    _, rank_idx = torch.max(torch.cat([eval_top_k_ids, target_ids], dim=1) == target_ids, dim=1)
    eval_ranks = torch.where(rank_idx == k, MAX_K + 1, rank_idx + 1)
    zero = torch.zeros(1, dtype=torch.float32, device=device)
    output: Dict[str, torch.Tensor] = {}
    for kk in (1, 5, 10, 50, 100, 200):
        output[f"ndcg@{kk}"] = torch.where(eval_ranks <= kk, 1.0 / torch.log2(eval_ranks + 1), zero)
    for kk in (1, 5, 10, 50, 100, 200, 500, 1000):
        output[f"hr@{kk}"] = eval_ranks <= kk
    output["mrr"] = 1.0 / eval_ranks
    if include_eval_time:
        output["eval_time"] = eval_time_all
    if include_eval_top_k_ids:
        output["eval_top_k_ids"] = eval_top_k_ids
    if target_ratings is not None:
        tr = target_ratings.squeeze(1)
        output["ndcg@10_>=4"] = torch.where(eval_ranks[tr >= 4] <= 10, 1.0 / torch.log2(eval_ranks[tr >= 4] + 1), zero)
        pos = tr >= min_positive_rating
        output[f"hr@10_>={min_positive_rating}"] = eval_ranks[pos] <= 10
        output[f"hr@50_>={min_positive_rating}"] = eval_ranks[pos] <= 50
        output[f"mrr_>={min_positive_rating}"] = 1.0 / eval_ranks[pos]
    return output


def _avg(x: torch.Tensor, world_size: int) -> float:
    s = torch.tensor([x.sum(), x.numel()], dtype=torch.float32, device=x.device)
    if world_size > 1:
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
    return s[0] / s[1]


def add_to_summary_writer(writer, batch_id: int, metrics: Dict[str, torch.Tensor], prefix: str, world_size: int) -> None:
    for key, values in metrics.items():
        avg_value = _avg(values, world_size)
        if writer is not None:
            writer.add_scalar(f"{prefix}/{key}", avg_value, batch_id)


def remap_legacy_checkpoint_keys(state_dict: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """The key shim of eval_from_checkpoint.py:366-376: checkpoints written before the item projection moved into
    `_item_embeddings_fn` carry `_ndp_module._item_proj_module.*`; map them to the current names."""
    out = {}
    for k, v in state_dict.items():
        k = k.replace("_ndp_module._item_proj_module.", "_ndp_module._item_embeddings_fn._item_emb_proj_module.")
        out[k] = v
    return out


def extract_mol_state_dict(model_state_dict: Dict[str, torch.Tensor], prefix: str = "module._ndp_module.") -> Dict[str, torch.Tensor]:
    """Pull the MoL module's tensors out of a full reference checkpoint (`checkpoint["model_state_dict"]`, keys under
    `module._ndp_module.` -- eval_from_checkpoint.py:369-372) after applying the legacy-key shim; the result loads into
    rails_amd.MoLSimilarity with strict=True."""
    sd = remap_legacy_checkpoint_keys(model_state_dict)
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
