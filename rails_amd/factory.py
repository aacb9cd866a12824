"""`create_mol_interaction_module`: wires the sub-module topology of a MoL similarity exactly as the
reference's gin factory does (modeling/similarity_utils.py:41-245), so parameter names -- and therefore
state_dict keys -- are identical.  gin is not required: plain keyword arguments with the same names.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .mol_module import (GeGLU, MoLSimilarity, RecoMoLItemEmbeddingsFn, RecoMoLQueryEmbeddingsFn,
                         SoftmaxDropoutCombiner, SwiGLU)


def init_mlp_xavier_weights_zero_bias(m) -> None:
    if isinstance(m, torch.nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight)
        if getattr(m, "bias", None) is not None:
            m.bias.data.fill_(0.0)


def _glu(kind: str, in_features: int, out_features: int) -> torch.nn.Module:
    return GeGLU(in_features, out_features) if kind == "geglu" else SwiGLU(in_features, out_features)


def _proj(kind: str, hidden: int, dropout: float, init_hidden: bool):
    """Dropout -> [GLU ->] Linear; module indices (0, 1, 2) are part of the state_dict contract."""

    def make(input_dim: int, output_dim: int) -> torch.nn.Module:
        if hidden > 0:
            seq = torch.nn.Sequential(torch.nn.Dropout(p=dropout), _glu(kind, input_dim, hidden), torch.nn.Linear(hidden, output_dim))
            return seq.apply(init_mlp_xavier_weights_zero_bias) if init_hidden else seq
        return torch.nn.Sequential(torch.nn.Dropout(p=dropout), torch.nn.Linear(input_dim, output_dim)).apply(init_mlp_xavier_weights_zero_bias)

    return make


def create_mol_interaction_module(
    query_embedding_dim: int,
    item_embedding_dim: int,
    dot_product_dimension: int,
    query_dot_product_groups: int,
    item_dot_product_groups: int,
    temperature: float,
    query_dropout_rate: float,
    query_hidden_dim: int,
    item_dropout_rate: float,
    item_hidden_dim: int,
    gating_query_hidden_dim: int,
    gating_qi_hidden_dim: int,
    gating_item_hidden_dim: int,
    softmax_dropout_rate: float,
    bf16_training: bool,
    gating_query_fn: bool = True,
    gating_item_fn: bool = True,
    dot_product_l2_norm: bool = True,
    query_nonlinearity: str = "geglu",
    item_nonlinearity: str = "geglu",
    uid_dropout_rate: float = 0.5,
    uid_embedding_hash_sizes: Optional[List[int]] = None,
    uid_embedding_level_dropout: bool = False,
    gating_combination_type: str = "glu_silu",
    gating_item_dropout_rate: float = 0.0,
    gating_qi_dropout_rate: float = 0.0,
    eps: float = 1e-6,
) -> Tuple[MoLSimilarity, str]:
    def gate_mlp(hidden: int, dropout: Optional[float], out_bias: bool):
        def make(input_dim: int, output_dim: int) -> torch.nn.Module:
            layers = [] if dropout is None else [torch.nn.Dropout(p=dropout)]
            if hidden > 0:
                layers += [torch.nn.Linear(input_dim, hidden), torch.nn.SiLU(), torch.nn.Linear(hidden, output_dim, bias=out_bias)]
            else:
                layers += [torch.nn.Linear(input_dim, output_dim)]
            return torch.nn.Sequential(*layers).apply(init_mlp_xavier_weights_zero_bias)

        return make

    mol_module = MoLSimilarity(
        query_embedding_dim=query_embedding_dim,
        item_embedding_dim=item_embedding_dim,
        dot_product_dimension=dot_product_dimension,
        query_dot_product_groups=query_dot_product_groups,
        item_dot_product_groups=item_dot_product_groups,
        temperature=temperature,
        dot_product_l2_norm=dot_product_l2_norm,
        query_embeddings_fn=RecoMoLQueryEmbeddingsFn(
            query_embedding_dim=query_embedding_dim,
            query_dot_product_groups=query_dot_product_groups,
            dot_product_dimension=dot_product_dimension,
            dot_product_l2_norm=dot_product_l2_norm,
            # the reference leaves the GLU branch of the QUERY projection at torch's default init
            proj_fn=_proj(query_nonlinearity, query_hidden_dim, query_dropout_rate, init_hidden=False),
            uid_embedding_hash_sizes=uid_embedding_hash_sizes or [],
            uid_dropout_rate=uid_dropout_rate,
            uid_embedding_level_dropout=uid_embedding_level_dropout,
            eps=eps,
        ),
        item_embeddings_fn=RecoMoLItemEmbeddingsFn(
            item_embedding_dim=item_embedding_dim,
            item_dot_product_groups=item_dot_product_groups,
            dot_product_dimension=dot_product_dimension,
            dot_product_l2_norm=dot_product_l2_norm,
            proj_fn=_proj(item_nonlinearity, item_hidden_dim, item_dropout_rate, init_hidden=True),
            eps=eps,
        ),
        item_proj_fn=None,
        gating_query_only_partial_fn=gate_mlp(gating_query_hidden_dim, None, out_bias=False) if gating_query_fn else None,
        gating_item_only_partial_fn=gate_mlp(gating_item_hidden_dim, gating_item_dropout_rate, out_bias=False) if gating_item_fn else None,
        gating_qi_partial_fn=gate_mlp(gating_qi_hidden_dim, gating_qi_dropout_rate, out_bias=True),
        gating_combination_type=gating_combination_type,
        gating_normalization_fn=lambda _: SoftmaxDropoutCombiner(dropout_rate=softmax_dropout_rate, eps=1e-6),
        eps=eps,
        autocast_bf16=bf16_training,
    )
    debug = (
        f"MoL-{query_dot_product_groups}x{item_dot_product_groups}x{dot_product_dimension}-t{temperature}-d{softmax_dropout_rate}"
        + ("-l2" if dot_product_l2_norm else "")
        + (f"-q{query_hidden_dim}d{query_dropout_rate}{query_nonlinearity}" if query_hidden_dim > 0 else f"-cd{query_dropout_rate}")
        + (f"-{item_hidden_dim}d{item_dropout_rate}{item_nonlinearity}" if item_hidden_dim > 0 else f"-id{item_dropout_rate}")
        + (f"-gq{gating_query_hidden_dim}" if gating_query_fn else "")
        + (f"-gi{gating_item_hidden_dim}d{gating_item_dropout_rate}" if gating_item_fn else "")
        + f"-gqi{gating_qi_hidden_dim}d{gating_qi_dropout_rate}-x-{gating_combination_type}"
    )
    if uid_embedding_hash_sizes is not None:
        debug += f"-uids{'-'.join(str(x) for x in uid_embedding_hash_sizes)}"
        if uid_dropout_rate > 0.0:
            debug += f"d{uid_dropout_rate}"
        if uid_embedding_level_dropout:
            debug += "-el"
    return mol_module, debug
