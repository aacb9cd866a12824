"""Host-side mirror of the reference's MoL similarity modules (rails/similarities/**), eval only.

The classes keep the reference's constructor signatures, attribute names and therefore
`state_dict()` keys, so a checkpoint of the reference loads unchanged.  They
hold parameters; the arithmetic of the path runs in the HIP kernels behind `rails_amd.engine`.
Training-time branches (dropout, mi_loss, uid l2 aux loss) are out of scope and raise.
"""
from __future__ import annotations

import abc
from typing import Callable, Dict, List, Optional, Tuple

import torch

from .engine import MolEngine, MolIndex, MolShapeSpec


def _eval_only(module: torch.nn.Module) -> None:
    if module.training:
        raise NotImplementedError(
            f"{type(module).__name__}: rails_amd implements the eval-mode path only; call .eval() first "
            "(training branches of the reference -- dropout, mi_loss, uid aux loss -- are out of scope)"
        )


class SimilarityModule(torch.nn.Module):
    """Type contract of reference rails/similarities/module.py:21-42."""

    @abc.abstractmethod
    def forward(self, query_embeddings: torch.Tensor, item_embeddings: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """(B, D), (1/B, X, D') -> ((B, X) similarities, aux losses)."""


class DotProductSimilarity(SimilarityModule):
    """Reference rails/similarities/dot_product_similarity_fn.py:24-68, on the HIP dot-product kernels."""

    def __init__(self) -> None:
        super().__init__()

    def debug_str(self) -> str:
        return "dp"

    def forward(self, query_embeddings: torch.Tensor, item_embeddings: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        from . import engine as E

        B_I, X, D = item_embeddings.size()
        if B_I == 1:  # (B, D) x (1, X, D) -> (B, X)
            out = E.MipsIndex(item_embeddings[0]).score(query_embeddings)
        else:         # (B * r, D) x (B, X, D) -> (B * r, X); r = 1 is the per-row-candidates case
            if query_embeddings.size(0) % B_I != 0:
                raise RuntimeError(f"shape '[{B_I}, -1, {D}]' is invalid for input of size {query_embeddings.numel()}")
            out = E.dot_rowwise(query_embeddings, item_embeddings)
        return out.to(query_embeddings.dtype), {}


class _GLU(torch.nn.Module):
    """Parameter holder for the query projection's gated unit (reference rails/similarities/layers.py:19-74):
    `_w` (in, 2*out) ~ N(0, 0.02^2), `_b` (1, 2*out) = 0.  Evaluated inside the fused query-prologue kernel."""

    kind = ""

    def __init__(self, in_features: int, out_features: int) -> None:
        super().__init__()
        self._in_features = in_features
        self._out_features = out_features
        self._w = torch.nn.Parameter(torch.empty((in_features, out_features * 2)).normal_(mean=0, std=0.02))
        self._b = torch.nn.Parameter(torch.zeros((1, out_features * 2)))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """The layer on its own (reference layers.py:36-43 / :68-74): rails_glu_f32 = the fp32 MFMA tile GEMM of csrc/hstu.hip for
        x W + b, then the gate pass.  Inside MoLSimilarity the unit is fused into the query prologue / index build instead."""
        from . import _lib
        from .engine import _on_device, _ptr, _require_device, _stream

        _require_device(x, f"{type(self).__name__} input")
        bs = x.size()[:-1]
        x2 = x.reshape(-1, self._in_features).float().contiguous()
        rows, F = x2.size(0), self._out_features
        w = self._w.detach().float().contiguous()
        b = self._b.detach().float().contiguous()
        scratch = torch.empty((rows, 2 * F), dtype=torch.float32, device=x.device)
        out = torch.empty((rows, F), dtype=torch.float32, device=x.device)
        kind = _lib.RAILS_GEGLU if self.kind == "geglu" else _lib.RAILS_SWIGLU
        with _on_device(x.device):
            _lib.check(_lib.load().rails_glu_f32(_ptr(x2), self._in_features, _ptr(w), _ptr(b), rows, self._in_features, F, kind,
                                                 _ptr(scratch), _ptr(out), _stream()), "rails_glu_f32")
        return out.to(x.dtype).reshape(bs + (F,))


class GeGLU(_GLU):
    kind = "geglu"


class SwiGLU(_GLU):
    kind = "swiglu"


class MoLEmbeddingsFn(torch.nn.Module):
    """Base of the query/item component-embedding generators (reference mol/embeddings_fn.py:56-78)."""


class RecoMoLQueryEmbeddingsFn(MoLEmbeddingsFn):
    """Parameters of reference rails/similarities/mol/query_embeddings_fns.py:129-254."""

    def __init__(
        self,
        query_embedding_dim: int,
        query_dot_product_groups: int,
        dot_product_dimension: int,
        dot_product_l2_norm: bool,
        proj_fn: Callable[[int, int], torch.nn.Module],
        eps: float,
        uid_embedding_hash_sizes: List[int],
        uid_dropout_rate: float,
        uid_embedding_level_dropout: bool = False,
    ) -> None:
        super().__init__()
        self._uid_embedding_hash_sizes: List[int] = list(uid_embedding_hash_sizes)
        self._query_emb_based_dot_product_groups: int = query_dot_product_groups - len(self._uid_embedding_hash_sizes)
        self._query_emb_proj_module: torch.nn.Module = proj_fn(
            query_embedding_dim, dot_product_dimension * self._query_emb_based_dot_product_groups
        )
        self._dot_product_dimension: int = dot_product_dimension
        self._dot_product_l2_norm: bool = dot_product_l2_norm
        for i, hash_size in enumerate(self._uid_embedding_hash_sizes):
            setattr(self, f"_uid_embeddings_{i}", torch.nn.Embedding(hash_size + 1, dot_product_dimension, padding_idx=0))
        self._uid_dropout_rate: float = uid_dropout_rate
        self._uid_embedding_level_dropout: bool = uid_embedding_level_dropout
        self._eps: float = eps

    def forward(self, input_embeddings: torch.Tensor, **kwargs):
        """(B, D) -> ((B, P_Q, d) component embeddings, {}) (reference query_embeddings_fns.py:175-254, eval mode): the query prologue
        kernel of the MoLSimilarity this module belongs to."""
        owner = _owner_of(self)
        return owner.get_query_component_embeddings(input_embeddings, **kwargs)


class RecoMoLItemEmbeddingsFn(MoLEmbeddingsFn):
    """Parameters of reference rails/similarities/mol/item_embeddings_fns.py:122-183."""

    def __init__(
        self,
        item_embedding_dim: int,
        item_dot_product_groups: int,
        dot_product_dimension: int,
        dot_product_l2_norm: bool,
        proj_fn: Callable[[int, int], torch.nn.Module],
        eps: float,
    ) -> None:
        super().__init__()
        self._item_emb_based_dot_product_groups: int = item_dot_product_groups
        self._item_emb_proj_module: torch.nn.Module = proj_fn(item_embedding_dim, dot_product_dimension * item_dot_product_groups)
        self._dot_product_dimension: int = dot_product_dimension
        self._dot_product_l2_norm: bool = dot_product_l2_norm
        self._eps: float = eps

    def forward(self, input_embeddings: torch.Tensor, **kwargs):
        """(..., D') -> ((..., P_X, d) component embeddings, {}) (reference item_embeddings_fns.py:149-183, eval mode): the index-build
        kernel of the MoLSimilarity this module belongs to."""
        owner = _owner_of(self)
        return owner.get_item_component_embeddings(input_embeddings, **kwargs)


class SoftmaxDropoutCombiner(torch.nn.Module):
    """Hyper-parameters of reference similarity_fn.py:66-96.  In eval the combiner is softmax followed by
    the renormalisation pi / clamp(sum pi, eps) whenever dropout_rate > 0 (similarity_fn.py:42-46); the
    fused kernel applies it always, which is exact for dropout_rate > 0 and a <=1-ulp rescale otherwise."""

    def __init__(self, dropout_rate: float, eps: float) -> None:
        super().__init__()
        self._dropout_rate: float = dropout_rate
        self._eps: float = eps

    def forward(self, gating_weights: torch.Tensor, x: torch.Tensor):
        """(..., L) gating weights, (..., L) logits -> ((...,) combined logits, {}): softmax, the eval-time renormalisation when
        dropout_rate > 0, weighted sum (reference similarity_fn.py:31-46, :66-96, eval mode) -- rails_mol_gate_combine.  Inside
        MoLSimilarity.forward the same arithmetic is fused into the scoring kernels."""
        if self.training:
            raise NotImplementedError("rails_amd is eval-only (no dropout / mi_loss)")
        from . import engine as E

        lead = gating_weights.shape[:-1]
        L = gating_weights.shape[-1]
        out, _ = E.gate_combine(x.reshape(-1, L), gating_weights.reshape(-1, L), None, None, 1, False, False, self._dropout_rate > 0.0, self._eps)
        return out.reshape(lead).to(x.dtype), {}


class MoLGatingFn(torch.nn.Module):
    """Parameters of reference similarity_fn.py:99-201 (pi_p(q, x))."""

    def __init__(
        self,
        num_logits: int,
        query_embedding_dim: int,
        item_embedding_dim: int,
        query_only_partial_fn: Optional[Callable[[int, int], torch.nn.Module]],
        item_only_partial_fn: Optional[Callable[[int, int], torch.nn.Module]],
        qi_partial_fn: Optional[Callable[[int, int], torch.nn.Module]],
        combination_type: str,
        normalization_fn: Callable[[int], torch.nn.Module],
    ) -> None:
        super().__init__()
        self._query_only_partial_module = query_only_partial_fn(query_embedding_dim, num_logits) if query_only_partial_fn else None
        self._item_only_partial_module = item_only_partial_fn(item_embedding_dim, num_logits) if item_only_partial_fn else None
        self._qi_partial_module = qi_partial_fn(num_logits, num_logits) if qi_partial_fn is not None else None
        if self._query_only_partial_module is None and self._item_only_partial_module is None and self._qi_partial_module is None:
            raise ValueError(
                "At least one of query_only_partial_fn, item_only_partial_fn, and qi_partial_fn must not be None."
            )
        self._num_logits: int = num_logits
        self._combination_type: str = combination_type
        self._normalization_fn: torch.nn.Module = normalization_fn(num_logits)

    def forward(self, logits: torch.Tensor, query_embeddings: torch.Tensor, item_embeddings: torch.Tensor):
        """logits (B, X, L) [already / temperature], query_embeddings (B, D), item_embeddings (1 or B, X, D') -> ((B, X), {}):
        reference similarity_fn.py:148-201 in eval mode, for callers that use the gate on its own (MoLSimilarity.forward never
        materialises these tensors: the same arithmetic is fused into the scoring kernels).  The three partial modules' Linear
        layers run on rails_gemm_f32 (fp32 MFMA), the combination + normalisation on rails_mol_gate_combine."""
        if self.training:
            raise NotImplementedError("rails_amd is eval-only")
        if self._combination_type not in ("glu_silu", "none"):
            raise NotImplementedError(f"combination_type {self._combination_type!r}")
        from . import engine as E

        B, X, L = logits.shape

        def mlp(seq, x2d):
            lin = _find(seq, torch.nn.Linear)
            if len(lin) == 1:
                return E.linear(x2d, lin[0].weight, lin[0].bias)
            if len(lin) != 2 or not _find(seq, torch.nn.SiLU):
                raise NotImplementedError("gate partial modules are Linear or Linear-SiLU-Linear (modeling/similarity_utils.py:147-207)")
            return E.linear(E.linear(x2d, lin[0].weight, lin[0].bias, silu=True), lin[1].weight, lin[1].bias)

        gq = mlp(self._query_only_partial_module, query_embeddings.reshape(B, -1)) if self._query_only_partial_module is not None else None
        per_row = item_embeddings.shape[0] != 1 or B == 1
        gi = mlp(self._item_only_partial_module, item_embeddings.reshape(-1, item_embeddings.shape[-1])) if self._item_only_partial_module is not None else None
        gqi = mlp(self._qi_partial_module, logits.reshape(B * X, L)) if self._qi_partial_module is not None else None
        norm = self._normalization_fn
        if not isinstance(norm, SoftmaxDropoutCombiner):
            raise NotImplementedError("normalization_fn must be a SoftmaxDropoutCombiner")
        out, _ = E.gate_combine(logits.reshape(B * X, L), gqi, gq, gi, X, per_row, self._combination_type == "glu_silu", norm._dropout_rate > 0.0, norm._eps)
        return out.reshape(B, X).to(logits.dtype), {}


def _owner_of(fn: torch.nn.Module):
    """The MoLSimilarity an embeddings-fn module was built into (set by MoLSimilarity.__init__; a weak reference, so the parent can
    still be collected).  The embeddings fns only hold parameters; their kernels belong to the parent's engine."""
    ref = getattr(fn, "_rails_owner", None)
    owner = ref() if ref is not None else None
    if owner is None:
        raise NotImplementedError(f"{type(fn).__name__}.forward needs the MoLSimilarity it was built into (its HIP engine holds the packed weights)")
    return owner


def _find(seq: torch.nn.Module, kind) -> List[torch.nn.Module]:
    return [m for m in seq.children() if isinstance(m, kind)] if seq is not None else []


def _version(t: torch.Tensor) -> int:
    # tensors created under torch.inference_mode() carry no version counter (and cannot be modified in place outside it)
    return 0 if t.is_inference() else t._version


class MoLSimilarity(SimilarityModule):
    """Drop-in for reference rails/similarities/mol/similarity_fn.py:204-413 (eval mode)."""

    def __init__(
        self,
        query_embedding_dim: int,
        item_embedding_dim: int,
        dot_product_dimension: int,
        query_dot_product_groups: int,
        item_dot_product_groups: int,
        temperature: float,
        dot_product_l2_norm: bool,
        query_embeddings_fn: MoLEmbeddingsFn,
        item_embeddings_fn: Optional[MoLEmbeddingsFn],
        item_proj_fn: Optional[Callable[[int, int], torch.nn.Module]],
        gating_query_only_partial_fn: Optional[Callable[[int, int], torch.nn.Module]],
        gating_item_only_partial_fn: Optional[Callable[[int, int], torch.nn.Module]],
        gating_qi_partial_fn: Optional[Callable[[int], torch.nn.Module]],
        gating_combination_type: str,
        gating_normalization_fn: Callable[[int], torch.nn.Module],
        eps: float,
        apply_query_embeddings_fn: bool = True,
        apply_item_embeddings_fn: bool = True,
        autocast_bf16: bool = False,
    ) -> None:
        super().__init__()
        self._gating_fn: MoLGatingFn = MoLGatingFn(
            num_logits=query_dot_product_groups * item_dot_product_groups,
            query_embedding_dim=query_embedding_dim,
            item_embedding_dim=item_embedding_dim,
            query_only_partial_fn=gating_query_only_partial_fn,
            item_only_partial_fn=gating_item_only_partial_fn,
            qi_partial_fn=gating_qi_partial_fn,
            combination_type=gating_combination_type,
            normalization_fn=gating_normalization_fn,
        )
        self._query_embeddings_fn: MoLEmbeddingsFn = query_embeddings_fn
        self._item_embeddings_fn: Optional[MoLEmbeddingsFn] = item_embeddings_fn
        import weakref

        for fn in (query_embeddings_fn, item_embeddings_fn):
            if fn is not None:
                object.__setattr__(fn, "_rails_owner", weakref.ref(self))   # not a submodule / parameter: plain attribute
        self._item_proj_module: Optional[torch.nn.Module] = None
        if item_embeddings_fn is None:
            raise NotImplementedError("the legacy item_proj_fn path (similarity_fn.py:252-259) is not supported")
        self._apply_query_embeddings_fn: bool = apply_query_embeddings_fn
        self._apply_item_embeddings_fn: bool = apply_item_embeddings_fn
        self._dot_product_l2_norm: bool = dot_product_l2_norm
        self._query_embedding_dim: int = query_embedding_dim
        self._item_embedding_dim: int = item_embedding_dim
        self._query_dot_product_groups: int = query_dot_product_groups
        self._item_dot_product_groups: int = item_dot_product_groups
        self._dot_product_dimension: int = dot_product_dimension
        self._temperature: float = temperature
        self._eps: float = eps
        # the reference autocasts to bf16 on CUDA when set; rails_amd always computes in fp32 (closer to
        # the fp32 oracle than the reference's own bf16 run), so the flag is accepted and ignored
        self._autocast_bf16: bool = autocast_bf16
        self._engine: Optional[MolEngine] = None
        self._engine_key = None
        self._param_list = None
        self._extra_engines: Dict[str, tuple] = {}
        # None -> RAILS_PRECISION or "fp32" (exact fp32 MFMA, the parity path); "f16x3" -> opt-in split-f16 gate MLP
        self.precision: Optional[str] = None

    # ---- binding the parameters to the HIP engine -----------------------------------------------
    def shape_spec(self) -> MolShapeSpec:
        """The module's topology as the C ABI's shape (every variant create_mol_interaction_module can wire,
        modeling/similarity_utils.py:41-245, except a pair gate without hidden layer and the broken glu_silu_ln)."""
        g = self._gating_fn
        if g._combination_type not in ("glu_silu", "none"):
            if g._combination_type == "glu_silu_ln":   # the reference's own branch raises a TypeError (normalized_shapes=)
                raise NotImplementedError("gating_combination_type 'glu_silu_ln' has no HIP kernel (it is unreachable in the reference too)")
            raise ValueError(f"Unknown combination_type {g._combination_type}")  # similarity_fn.py:198-199
        qf, itf = self._query_embeddings_fn, self._item_embeddings_fn
        if not isinstance(qf, RecoMoLQueryEmbeddingsFn) or not isinstance(itf, RecoMoLItemEmbeddingsFn):
            raise NotImplementedError("only RecoMoLQueryEmbeddingsFn / RecoMoLItemEmbeddingsFn are supported (LMMoL* is out of scope)")
        q_glus, i_glus = _find(qf._query_emb_proj_module, _GLU), _find(itf._item_emb_proj_module, _GLU)
        if g._qi_partial_module is None:
            raise NotImplementedError("the fused kernel needs the pair gate part")
        has_q, has_i = g._query_only_partial_module is not None, g._item_only_partial_module is not None
        if g._combination_type == "glu_silu" and not (has_q and has_i):
            # the reference fails here too (None * tensor, similarity_fn.py:176-178)
            raise TypeError("gating_combination_type 'glu_silu' needs the query-only and the item-only gate part")
        qi_linears = _find(g._qi_partial_module, torch.nn.Linear)
        if len(qi_linears) not in (1, 2):
            raise NotImplementedError("the pair gate is Linear-SiLU-Linear or one Linear (modeling/similarity_utils.py:186-207)")
        return MolShapeSpec(
            query_embedding_dim=self._query_embedding_dim,
            item_embedding_dim=self._item_embedding_dim,
            dot_product_dimension=self._dot_product_dimension,
            query_dot_product_groups=self._query_dot_product_groups,
            item_dot_product_groups=self._item_dot_product_groups,
            query_hidden_dim=q_glus[0]._out_features if q_glus else -1,
            gating_query_hidden_dim=_find(g._query_only_partial_module, torch.nn.Linear)[0].out_features if has_q else -1,
            gating_item_hidden_dim=_find(g._item_only_partial_module, torch.nn.Linear)[0].out_features if has_i else -1,
            gating_qi_hidden_dim=qi_linears[0].out_features if len(qi_linears) == 2 else -1,   # -1: no hidden layer (one Linear(L, L))
            query_nonlinearity=q_glus[0].kind if q_glus else "geglu",
            uid_embedding_hash_sizes=tuple(qf._uid_embedding_hash_sizes),
            dot_product_l2_norm=bool(self._dot_product_l2_norm),
            temperature=float(self._temperature),
            eps=float(self._eps),
            item_hidden_dim=i_glus[0]._out_features if i_glus else -1,
            item_nonlinearity=i_glus[0].kind if i_glus else "geglu",
            gating_combination_type=g._combination_type,
            gating_query_fn=has_q,
            gating_item_fn=has_i,
        )

    def engine(self, precision: Optional[str] = None, _params_as_checked: bool = False) -> MolEngine:
        """HIP engine bound to the CURRENT parameter values (rebuilt when any parameter changed).  `precision` asks for an engine of
        another precision than the module's own (the proved exact top-k keeps a split-f16 engine next to the fp32 one); each
        precision has its own cached engine."""
        _eval_only(self)
        if not self._apply_query_embeddings_fn or not self._apply_item_embeddings_fn:
            raise NotImplementedError("apply_query_embeddings_fn / apply_item_embeddings_fn = False is not supported")
        # The check runs on every call of the hot path, so it walks a cached list of the parameter objects rather than
        # state_dict() (70 us per call for 27 tensors).  In-place edits (optimizer steps, load_state_dict's copy_) bump
        # `_version`; `.to()/.float()` swap `.data` (new data_ptr) or go through _apply, which drops the cached list.
        plist = self._param_list
        if plist is None:
            plist = self._param_list = [v for _, v in self.state_dict(keep_vars=True).items()]
        if precision is not None and precision != self.precision:
            hit = self._extra_engines.get(precision)
            if _params_as_checked and hit is not None and self._engine_key is not None and hit[0][1:] == self._engine_key[1:]:
                # the caller has just asked for the module's own engine (same thread, nothing in between): the parameters are those of
                # _engine_key, and this precision's engine was built from the same ones -- no second walk over the 27 tensors
                return hit[1]
            key = (precision,) + tuple((v.data_ptr(), _version(v)) for v in plist)
            if hit is None or hit[0] != key:
                params = dict(self.state_dict(keep_vars=True))
                self._param_list = list(params.values())
                key = (precision,) + tuple((v.data_ptr(), _version(v)) for v in self._param_list)
                hit = self._extra_engines[precision] = (key, MolEngine(self.shape_spec(), params, precision=precision))
            return hit[1]
        key = (self.precision,) + tuple((v.data_ptr(), _version(v)) for v in plist)
        if self._engine is None or key != self._engine_key:
            params = dict(self.state_dict(keep_vars=True))
            self._param_list = list(params.values())
            key = (self.precision,) + tuple((v.data_ptr(), _version(v)) for v in self._param_list)
            self._engine = MolEngine(self.shape_spec(), params, precision=self.precision)
            self._engine_key = key
        return self._engine

    def _apply(self, fn, *args, **kwargs):
        self._param_list = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        # drop the engine outright: load_state_dict copies in place, and under torch.inference_mode() an in-place copy does NOT bump
        # the parameters' version counters -- the (data_ptr, version) key above would keep serving the old weights
        self._param_list = None
        self._engine = None
        self._engine_key = None
        self._extra_engines = {}
        return super().load_state_dict(*args, **kwargs)

    # ---- reference API --------------------------------------------------------------------------
    def get_query_component_embeddings(self, input_embeddings: torch.Tensor, decoupled_inference: bool = False, **kwargs):
        """(B, D) -> ((B, P_Q, d), {}).  Reference similarity_fn.py:270-292."""
        if decoupled_inference and not self._apply_query_embeddings_fn:
            return input_embeddings, {}
        _, eq, _ = self.engine().query_pack(input_embeddings, kwargs.get("user_ids"), want_plain=True)
        return eq.to(input_embeddings.dtype), {}

    def get_item_component_embeddings(self, input_embeddings: torch.Tensor, decoupled_inference: bool = False, **kwargs):
        """(..., D') -> ((..., P_X, d), {}).  Reference similarity_fn.py:294-339."""
        if decoupled_inference and not self._apply_item_embeddings_fn:
            return input_embeddings, {}
        eng = self.engine()
        lead = input_embeddings.shape[:-1]
        ex, _ = eng.unpack_index(eng.build_index(input_embeddings.reshape(-1, input_embeddings.shape[-1])), want_gi=False)
        return ex.reshape(lead + ex.shape[1:]).to(input_embeddings.dtype), {}

    def forward(self, query_embeddings: torch.Tensor, item_embeddings: torch.Tensor, **kwargs) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
        """(B, D), (1/B, X, D') -> ((B, X), {}).  Reference similarity_fn.py:341-413."""
        eng = self.engine()
        B = query_embeddings.size(0)
        Bp, X = item_embeddings.shape[0], item_embeddings.shape[1]
        qpack, _, _ = eng.query_pack(query_embeddings, kwargs.get("user_ids"))
        if Bp == 1:
            logits = eng.score_dense(qpack, B, eng.build_index(item_embeddings[0]))
        else:
            if Bp != B:
                raise RuntimeError(f"item_embeddings.shape[0] must be 1 or B={B}, got {Bp}")
            Xp = (X + 31) // 32 * 32
            items = item_embeddings
            if Xp != X:
                items = torch.nn.functional.pad(items, (0, 0, 0, Xp - X))
            cand = eng.build_index(items.reshape(B * Xp, items.shape[-1]))
            logits = eng.score_candidates(qpack, B, cand, Xp)[:, :X]
        return logits.to(query_embeddings.dtype), {}
