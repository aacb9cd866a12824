"""Thin Python driver over the C ABI: owns the packed device buffers, passes raw device pointers.

PyTorch is used for device memory, the current stream and (elsewhere) torch.distributed only; every
floating-point operation of the path runs in the HIP kernels behind librails_amd.so.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import os
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib

TILE_ITEMS = 32
PRECISIONS = ("fp32", "f16x3", "f16x3-exact", "f16-exact", "f16x1")
# "f16x3-exact" / "f16-exact": the f16x3 / one-product f16 kernels pick candidates, an fp32 companion engine (`MolEngine.exact`)
# re-scores them, so the brute-force top-k is the fp32 path's result bit for bit (topk_modules.MoLBruteForceTopK).  Every other
# use of such an engine behaves as "f16x3": the packs are the same, only the dense pass of "f16-exact" runs the one-product kernels.
# "f16x1": the one-product kernels as they are (logits ~1e-2 off: measurement and tools only, NOT a parity mode).
_C_PRECISION = {"fp32": _lib.RAILS_PRECISION_FP32, "f16x3": _lib.RAILS_PRECISION_F16X3, "f16x1": _lib.RAILS_PRECISION_F16X1}


def default_precision() -> str:
    """"fp32" (exact fp32 MFMA, the parity path) unless RAILS_PRECISION selects an opt-in mode ("f16x3", "f16x3-exact")."""
    p = os.environ.get("RAILS_PRECISION", "fp32")
    if p not in PRECISIONS:
        raise ValueError(f"RAILS_PRECISION must be one of {PRECISIONS}, got {p!r}")
    return p


@dataclasses.dataclass(frozen=True)
class MolShapeSpec:
    """Hyper-parameters of a MoL module (names as in create_mol_interaction_module,
    reference modeling/similarity_utils.py:42-70)."""

    query_embedding_dim: int
    item_embedding_dim: int
    dot_product_dimension: int
    query_dot_product_groups: int
    item_dot_product_groups: int
    query_hidden_dim: int
    gating_query_hidden_dim: int
    gating_item_hidden_dim: int
    gating_qi_hidden_dim: int
    query_nonlinearity: str = "geglu"
    uid_embedding_hash_sizes: Tuple[int, ...] = ()
    dot_product_l2_norm: bool = True
    temperature: float = 0.05
    eps: float = 1e-6
    item_hidden_dim: int = -1                    # > 0: GLU hidden layer in the item projection
    item_nonlinearity: str = "geglu"
    gating_combination_type: str = "glu_silu"    # "glu_silu" | "none"
    gating_query_fn: bool = True                 # False: no query-only gate part (only with "none")
    gating_item_fn: bool = True                  # False: no item-only gate part (only with "none")

    @property
    def num_logits(self) -> int:
        return self.query_dot_product_groups * self.item_dot_product_groups

    def to_c(self, precision: str = "fp32") -> _lib.MolShape:
        if self.query_nonlinearity not in ("geglu", "swiglu") or self.item_nonlinearity not in ("geglu", "swiglu"):
            raise ValueError(f"Unknown nonlinearity {self.query_nonlinearity} / {self.item_nonlinearity}")
        if self.gating_combination_type not in ("glu_silu", "none"):
            raise ValueError(f"Unknown combination_type {self.gating_combination_type}")
        return _lib.MolShape(
            self.query_embedding_dim, self.item_embedding_dim, self.dot_product_dimension,
            self.query_dot_product_groups, self.item_dot_product_groups, self.query_hidden_dim,
            self.gating_query_hidden_dim, self.gating_item_hidden_dim, self.gating_qi_hidden_dim,
            _lib.RAILS_GEGLU if self.query_nonlinearity == "geglu" else _lib.RAILS_SWIGLU,
            len(self.uid_embedding_hash_sizes), 1 if self.dot_product_l2_norm else 0,
            float(self.temperature), float(self.eps),
            _C_PRECISION[precision],
            int(self.item_hidden_dim), _lib.RAILS_GEGLU if self.item_nonlinearity == "geglu" else _lib.RAILS_SWIGLU,
            _lib.RAILS_COMBINE_NONE if self.gating_combination_type == "none" else _lib.RAILS_COMBINE_GLU_SILU,
            1 if self.gating_query_fn else 0, 1 if self.gating_item_fn else 0,
        )

    def weight_fields(self) -> Dict[str, str]:
        """state_dict key -> field of rails_mol_weights for THIS topology (the module indices inside the
        reference's Sequentials are part of the keys, modeling/similarity_utils.py:88-207)."""
        f: Dict[str, str] = {}
        q = "_query_embeddings_fn._query_emb_proj_module."
        if self.query_hidden_dim > 0:
            f.update({q + "1._w": "q_glu_w", q + "1._b": "q_glu_b", q + "2.weight": "q_proj_w", q + "2.bias": "q_proj_b"})
        else:
            f.update({q + "1.weight": "q_proj_w", q + "1.bias": "q_proj_b"})
        i = "_item_embeddings_fn._item_emb_proj_module."
        if self.item_hidden_dim > 0:
            f.update({i + "1._w": "i_glu_w", i + "1._b": "i_glu_b", i + "2.weight": "i_proj_w", i + "2.bias": "i_proj_b"})
        else:
            f.update({i + "1.weight": "i_proj_w", i + "1.bias": "i_proj_b"})
        if self.gating_query_fn:
            g = "_gating_fn._query_only_partial_module."
            f.update({g + "0.weight": "gq_w1", g + "0.bias": "gq_b1", g + "2.weight": "gq_w2"})
        if self.gating_item_fn:
            g = "_gating_fn._item_only_partial_module."
            f.update({g + "1.weight": "gi_w1", g + "1.bias": "gi_b1", g + "3.weight": "gi_w2"})
        g = "_gating_fn._qi_partial_module."
        if self.gating_qi_hidden_dim > 0:
            f.update({g + "1.weight": "gqi_w1", g + "1.bias": "gqi_b1", g + "3.weight": "gqi_w2", g + "3.bias": "gqi_b2"})
        else:   # Sequential(Dropout, Linear(L, L)): modeling/similarity_utils.py:199-206
            f.update({g + "1.weight": "gqi_w1", g + "1.bias": "gqi_b1"})
        return f


# state_dict key -> field of rails_mol_weights for the shipped topology (MolShapeSpec.weight_fields() covers the variants)
WEIGHT_FIELDS = {
    "_query_embeddings_fn._query_emb_proj_module.1._w": "q_glu_w",
    "_query_embeddings_fn._query_emb_proj_module.1._b": "q_glu_b",
    "_query_embeddings_fn._query_emb_proj_module.2.weight": "q_proj_w",
    "_query_embeddings_fn._query_emb_proj_module.2.bias": "q_proj_b",
    "_item_embeddings_fn._item_emb_proj_module.1.weight": "i_proj_w",
    "_item_embeddings_fn._item_emb_proj_module.1.bias": "i_proj_b",
    "_gating_fn._query_only_partial_module.0.weight": "gq_w1",
    "_gating_fn._query_only_partial_module.0.bias": "gq_b1",
    "_gating_fn._query_only_partial_module.2.weight": "gq_w2",
    "_gating_fn._item_only_partial_module.1.weight": "gi_w1",
    "_gating_fn._item_only_partial_module.1.bias": "gi_b1",
    "_gating_fn._item_only_partial_module.3.weight": "gi_w2",
    "_gating_fn._qi_partial_module.1.weight": "gqi_w1",
    "_gating_fn._qi_partial_module.1.bias": "gqi_b1",
    "_gating_fn._qi_partial_module.3.weight": "gqi_w2",
    "_gating_fn._qi_partial_module.3.bias": "gqi_b2",
}


_cur_dev = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device    # the binding itself: torch.cuda.current_device() adds a lazy-init check (0.7 us, six per step)


def _stream() -> C.c_void_p:
    # raw handle of torch's current stream on the current device (torch.cuda.current_stream() costs ~9 us per call)
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(_cur_dev()))


class _on_device:
    """`with _on_device(dev)`: make `dev` the current HIP device for the enclosed launches.  Unlike torch.cuda.device it
    costs nothing when `dev` already is current (the one-process-per-GPU case), which matters for sub-millisecond steps."""

    __slots__ = ("idx", "prev")

    def __init__(self, dev: torch.device):
        self.idx = dev.index if dev.index is not None else _cur_dev()
        self.prev = -1

    def __enter__(self):
        cur = _cur_dev()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)

    def __exit__(self, *exc):
        if self.prev >= 0:
            torch.cuda.set_device(self.prev)
        return False


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


_EMPTY_INV: Dict[torch.device, torch.Tensor] = {}


def _inv_ptr(invalid_ids: Optional[torch.Tensor]) -> C.c_void_p:
    """Pointer of a (rows, width) seen-id tensor for the entry points that filter.  A zero-WIDTH tensor (a workload without history:
    BASELINE configs 4 and 5) has a null data_ptr, which the C ABI reads as "no filter requested": hand it one never-read word instead,
    so that `width == 0` keeps meaning "filter with nothing to remove" (k of the k' winners, same outputs contract)."""
    if invalid_ids is None or invalid_ids.numel() > 0:
        return _ptr(invalid_ids)
    dev = invalid_ids.device
    if dev not in _EMPTY_INV:
        _EMPTY_INV[dev] = torch.zeros(1, dtype=torch.int64, device=dev)
    return _ptr(_EMPTY_INV[dev])


def _require_device(t: torch.Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what} must live on the GPU: rails_amd runs its path in HIP kernels only and has no CPU fallback "
            f"(got a tensor on {t.device})"
        )


def _f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 + contiguous view/copy of a device tensor (plumbing: dtype cast, no math)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class MolIndex:
    """Tile-packed item index: Ex (l2-normalised component embeddings) + gi (item gate) per item."""

    def __init__(self, buf: torch.Tensor, n_items: int):
        self.buf = buf
        self.n_items = n_items

    def items(self, lo: int, hi: int) -> "MolIndex":
        """The sub-index of items [lo, hi) as a view (lo must be a tile boundary: tiles are stored back to back)."""
        if lo % TILE_ITEMS != 0 or not 0 <= lo <= hi <= self.n_items:
            raise ValueError(f"sub-index [{lo}, {hi}) must start on a {TILE_ITEMS}-item tile boundary inside [0, {self.n_items}]")
        tiles = (self.n_items + TILE_ITEMS - 1) // TILE_ITEMS
        tile_floats = self.buf.numel() // max(tiles, 1)
        t0, t1 = lo // TILE_ITEMS, (hi + TILE_ITEMS - 1) // TILE_ITEMS
        return MolIndex(self.buf[t0 * tile_floats : t1 * tile_floats], hi - lo)


class MolEngine:
    """One MoL module's weights bound to the HIP kernels."""

    def __init__(self, spec: MolShapeSpec, weights: Dict[str, torch.Tensor], precision: Optional[str] = None):
        self.lib = _lib.load()
        self.spec = spec
        precision = precision or default_precision()
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {PRECISIONS}, got {precision!r}")
        self.precision = {"f16x3-exact": "f16x3", "f16-exact": "f16x3", "f16x1": "f16x3"}.get(precision, precision)   # format of the packs
        self.exact: Optional["MolEngine"] = MolEngine(spec, weights, "fp32") if precision.endswith("-exact") else None
        self.shape = spec.to_c(self.precision)
        # the shape the DENSE pass is launched with: the one-product kernels for "f16-exact" / "f16x1" (same packs, lo halves ignored)
        self.dense_precision = "f16x1" if precision in ("f16-exact", "f16x1") else self.precision
        self.dense_shape = spec.to_c(self.dense_precision)
        self._fp32_shape = spec.to_c("fp32")     # for the derived bf16 tables, which are cut from an fp32-format index
        if not self.lib.rails_mol_shape_supported(C.byref(self.shape)):
            raise NotImplementedError(_lib.last_error())
        self._keep = []  # fp32 contiguous device tensors the weight struct points into
        w = _lib.MolWeights()
        for key, field in spec.weight_fields().items():
            if key not in weights:
                raise KeyError(f"MoL weight '{key}' is missing")
            t = weights[key]
            _require_device(t, f"weight {key}")
            t = _f32c(t.detach())
            self._keep.append(t)
            setattr(w, field, t.data_ptr())
        for i, hs in enumerate(spec.uid_embedding_hash_sizes):
            key = f"_query_embeddings_fn._uid_embeddings_{i}.weight"
            t = _f32c(weights[key].detach())
            _require_device(t, f"weight {key}")
            if t.shape[0] != hs + 1:
                raise ValueError(f"{key} has {t.shape[0]} rows, expected hash_size + 1 = {hs + 1}")
            self._keep.append(t)
            w.uid_table[i] = t.data_ptr()
            w.uid_hash_size[i] = int(hs)
        self.weights = w
        self.device = self._keep[0].device
        if self.precision == "f16x3":
            self._check_f16_range(weights)
        n = self.lib.rails_mol_gate_pack_floats(C.byref(self.shape))
        self.gate_pack = torch.empty(n, dtype=torch.float32, device=self.device)
        with _on_device(self.device):
            _lib.check(
                self.lib.rails_mol_pack_gate_weights(C.byref(self.shape), C.byref(self.weights), _ptr(self.gate_pack), _stream()),
                "rails_mol_pack_gate_weights",
            )

    def _check_f16_range(self, weights: Dict[str, torch.Tensor]) -> None:
        """precision='f16x3' keeps cl, hid and the gate weights as f16 hi + lo.  Small values are safe (f16 subnormals are
        kept by the MFMA: absolute resolution 2^-25); large ones are not (f16 max 65504), so bound the hidden layer for
        |cl| <= 1/temperature (unit-norm components) and refuse weights that could overflow."""
        w1 = weights["_gating_fn._qi_partial_module.1.weight"].detach().float()
        b1 = weights["_gating_fn._qi_partial_module.1.bias"].detach().float()
        w2 = weights["_gating_fn._qi_partial_module.3.weight"].detach().float()
        cl_max = 1.0 / self.spec.temperature * 1.001
        log2e = 1.4426950408889634
        t_max = log2e * (cl_max * float(w1.abs().sum(1).max()) + float(b1.abs().max()))
        limit = 60000.0
        if cl_max >= limit or t_max >= limit or log2e * float(w1.abs().max()) >= limit or float(w2.abs().max()) >= limit:
            raise NotImplementedError("precision='f16x3': pair-gate weights / temperature put an operand outside the f16 range; use fp32")

    # ---- item side ----------------------------------------------------------------------------
    def build_index(self, items: torch.Tensor) -> MolIndex:
        """items: (N, D_i) on the GPU -> tile-packed index (reference: item-side work of
        rails/similarities/mol/similarity_fn.py:378-387 + :170-171, done once)."""
        _require_device(items, "item_embeddings")
        if items.dim() != 2 or items.shape[1] != self.spec.item_embedding_dim:
            raise ValueError(f"item_embeddings must be (N, {self.spec.item_embedding_dim}), got {tuple(items.shape)}")
        items = _f32c(items)
        n = items.shape[0]
        floats = self.lib.rails_mol_index_floats(C.byref(self.shape), n)
        buf = torch.empty(floats, dtype=torch.float32, device=items.device)
        with _on_device(items.device):
            _lib.check(
                self.lib.rails_mol_index_build(C.byref(self.shape), C.byref(self.weights), _ptr(items), n, _ptr(buf), _stream()),
                "rails_mol_index_build",
            )
        return MolIndex(buf, n)

    def unpack_index(self, index: MolIndex, want_ex: bool = True, want_gi: bool = True):
        s = self.spec
        ex = torch.empty((index.n_items, s.item_dot_product_groups, s.dot_product_dimension), dtype=torch.float32, device=index.buf.device) if want_ex else None
        gi = torch.empty((index.n_items, s.num_logits), dtype=torch.float32, device=index.buf.device) if want_gi else None
        with _on_device(index.buf.device):
            _lib.check(
                self.lib.rails_mol_index_unpack(C.byref(self.shape), _ptr(index.buf), index.n_items, _ptr(ex), _ptr(gi), _stream()),
                "rails_mol_index_unpack",
            )
        return ex, gi

    def build_index_rows(self, index: MolIndex) -> torch.Tensor:
        """Row-major copy of an exact-fp32 index (include/rails_amd.h rails_mol_index_rows_build): what score_indexed_rows reads candidates from."""
        floats = self.lib.rails_mol_index_rows_floats(C.byref(self.shape), index.n_items)
        if floats == 0:
            raise NotImplementedError("the row-major index copy exists for exact-fp32 indexes only")
        rows = torch.empty(floats, dtype=torch.float32, device=index.buf.device)
        with _on_device(index.buf.device):
            _lib.check(self.lib.rails_mol_index_rows_build(C.byref(self.shape), _ptr(index.buf), index.n_items, _ptr(rows), _stream()), "rails_mol_index_rows_build")
        return rows

    def score_indexed_rows(self, qpack: torch.Tensor, batch: int, rows: torch.Tensor, n_items: int, positions: torch.Tensor,
                           counts: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """score_indexed with the candidates read from the row-major copy: whole cache lines per candidate, same bits.
        counts: per-row candidate counts (int32 on the device; candidates_select's): only the first counts[b] logits of row b are written."""
        if positions.dtype != torch.int64 or positions.device != rows.device or not positions.is_contiguous():
            positions = positions.to(device=rows.device, dtype=torch.int64).contiguous()
        n_cand = positions.shape[1]
        if out is None:
            out = torch.empty((batch, n_cand), dtype=torch.float32, device=rows.device)
        with _on_device(rows.device):
            _lib.check(
                self.lib.rails_mol_score_indexed_rows(C.byref(self.shape), _ptr(self.gate_pack), _ptr(qpack), batch, _ptr(rows), n_items, _ptr(positions), n_cand,
                                                      _ptr(out), out.stride(0), _ptr(counts), _stream()),
                "rails_mol_score_indexed_rows",
            )
        return out

    def gather_index(self, index: MolIndex, cand_idx: torch.Tensor) -> Tuple[MolIndex, int]:
        """cand_idx: (rows, K) int64 positions -> per-row tile-packed index, K padded to a multiple of 32."""
        _require_device(cand_idx, "candidate indices")
        rows, K = cand_idx.shape
        Kp = (K + TILE_ITEMS - 1) // TILE_ITEMS * TILE_ITEMS
        idx = cand_idx.to(torch.int64)
        if Kp != K:
            idx = torch.nn.functional.pad(idx, (0, Kp - K), value=-1)
        idx = idx.contiguous()
        floats = self.lib.rails_mol_index_floats(C.byref(self.shape), rows * Kp)
        out = torch.empty(floats, dtype=torch.float32, device=idx.device)
        with _on_device(idx.device):
            _lib.check(
                self.lib.rails_mol_index_gather(C.byref(self.shape), _ptr(index.buf), index.n_items, _ptr(idx), rows, Kp, _ptr(out), _stream()),
                "rails_mol_index_gather",
            )
        return MolIndex(out, rows * Kp), Kp

    # ---- query side ---------------------------------------------------------------------------
    def query_pack(self, q: torch.Tensor, user_ids: Optional[torch.Tensor] = None, want_plain: bool = False,
                   out: Optional[torch.Tensor] = None):
        _require_device(q, "query_embeddings")
        if q.dim() != 2 or q.shape[1] != self.spec.query_embedding_dim:
            raise ValueError(f"query_embeddings must be (B, {self.spec.query_embedding_dim}), got {tuple(q.shape)}")
        q = _f32c(q)
        B = q.shape[0]
        uid = None
        if len(self.spec.uid_embedding_hash_sizes) > 0:
            if user_ids is None:
                raise KeyError("user_ids")  # the reference does kwargs["user_ids"] (query_embeddings_fns.py:206)
            uid = user_ids.to(device=q.device, dtype=torch.int64).contiguous()
            if uid.numel() != B:
                # the reference fails in torch.cat on the same mismatch (query_embeddings_fns.py:206-216)
                raise RuntimeError(f"Sizes of tensors must match: user_ids has {tuple(uid.shape)} for a batch of {B} queries")
        n = self.lib.rails_mol_query_pack_floats(C.byref(self.shape), B)
        pack = out if out is not None and out.numel() == n and out.device == q.device else torch.empty(n, dtype=torch.float32, device=q.device)
        s = self.spec
        eq = torch.empty((B, s.query_dot_product_groups, s.dot_product_dimension), dtype=torch.float32, device=q.device) if want_plain else None
        gq = torch.empty((B, s.num_logits), dtype=torch.float32, device=q.device) if want_plain else None
        with _on_device(q.device):
            _lib.check(
                self.lib.rails_mol_query_prologue(C.byref(self.shape), C.byref(self.weights), _ptr(q), _ptr(uid), B, _ptr(pack), _ptr(eq), _ptr(gq), _stream()),
                "rails_mol_query_prologue",
            )
        return pack, eq, gq

    def query_pack_both(self, q: torch.Tensor, user_ids: Optional[torch.Tensor], out: torch.Tensor, out_other: torch.Tensor):
        """One prologue, two packs: `out` in this engine's format, `out_other` in the other one (for the fp32 companion of a
        verified fast mode).  Both must hold rails_mol_query_pack_floats floats."""
        _require_device(q, "query_embeddings")
        if q.dim() != 2 or q.shape[1] != self.spec.query_embedding_dim:
            raise ValueError(f"query_embeddings must be (B, {self.spec.query_embedding_dim}), got {tuple(q.shape)}")
        q = _f32c(q)
        B = q.shape[0]
        uid = None
        if len(self.spec.uid_embedding_hash_sizes) > 0:
            if user_ids is None:
                raise KeyError("user_ids")
            uid = user_ids.to(device=q.device, dtype=torch.int64).contiguous()
            if uid.numel() != B:
                raise RuntimeError(f"Sizes of tensors must match: user_ids has {tuple(uid.shape)} for a batch of {B} queries")
        n = self.lib.rails_mol_query_pack_floats(C.byref(self.shape), B)
        if out.numel() != n or out_other.numel() != n:
            raise ValueError(f"query packs must hold {n} floats")
        with _on_device(q.device):
            _lib.check(self.lib.rails_mol_query_prologue_both(C.byref(self.shape), C.byref(self.weights), _ptr(q), _ptr(uid), B, _ptr(out),
                                                              _ptr(out_other), _stream()), "rails_mol_query_prologue_both")
        return out, out_other

    # ---- scoring ------------------------------------------------------------------------------
    def score_dense(self, qpack: torch.Tensor, batch: int, index: MolIndex, out: Optional[torch.Tensor] = None,
                    run_if: Optional[torch.Tensor] = None) -> torch.Tensor:
        """run_if: launch predicate (an int32 device scalar; see _pred): the launch is a no-op unless it is non-zero on the device."""
        if out is None:
            out = torch.empty((batch, index.n_items), dtype=torch.float32, device=index.buf.device)
        with _on_device(index.buf.device):
            _lib.check(
                self.lib.rails_mol_score_dense(C.byref(self.dense_shape), _ptr(self.gate_pack), _ptr(qpack), batch, _ptr(index.buf), index.n_items, _ptr(out), out.stride(0), _pred(run_if), _stream()),
                "rails_mol_score_dense",
            )
        return out

    def score_dense_upper_supported(self) -> bool:
        """True iff this engine's dense precision has the upper-bound first pass (include/rails_amd.h rails_mol_score_dense_upper)."""
        memo = self.__dict__
        if "_upper_ok" not in memo:
            memo["_upper_ok"] = bool(self.lib.rails_mol_score_dense_upper_supported(C.byref(self.dense_shape)))
        return memo["_upper_ok"]

    def score_dense_upper(self, qpack: torch.Tensor, batch: int, index: MolIndex, poly: Tuple[float, float, float], out: Optional[torch.Tensor] = None,
                          run_if: Optional[torch.Tensor] = None) -> torch.Tensor:
        """f16x3 logit + (ub2 c + ub1) c + ub0 per pair, c = the pair's largest |cross logit|: an upper bound of the fp32 logit when `poly` is
        f16x3_bound.upper_bound_poly's (ub2, ub1, ub0)."""
        if out is None:
            out = torch.empty((batch, index.n_items), dtype=torch.float32, device=index.buf.device)
        with _on_device(index.buf.device):
            _lib.check(
                self.lib.rails_mol_score_dense_upper(C.byref(self.dense_shape), _ptr(self.gate_pack), _ptr(qpack), batch, _ptr(index.buf), index.n_items,
                                                     float(poly[0]), float(poly[1]), float(poly[2]), _ptr(out), out.stride(0), _pred(run_if), _stream()),
                "rails_mol_score_dense_upper",
            )
        return out

    def score_indexed_supported(self, batch: int, n_cand: int) -> bool:
        key = (int(batch), int(n_cand))
        memo = self.__dict__.setdefault("_indexed_ok", {})     # a dry run of the launch per call otherwise: host time of every rerank
        if key not in memo:
            memo[key] = bool(self.lib.rails_mol_score_indexed_supported(C.byref(self.shape), key[0], key[1]))
        return memo[key]

    def score_indexed(self, qpack: torch.Tensor, batch: int, index: MolIndex, positions: torch.Tensor) -> torch.Tensor:
        """(B, n_cand) logits of per-row candidates given as positions of `index` (all inside the index):
        gather_index + score_candidates without the gathered copy (include/rails_amd.h rails_mol_score_indexed)."""
        positions = positions.to(device=index.buf.device, dtype=torch.int64).contiguous()
        n_cand = positions.shape[1]
        out = torch.empty((batch, n_cand), dtype=torch.float32, device=index.buf.device)
        with _on_device(index.buf.device):
            _lib.check(
                self.lib.rails_mol_score_indexed(C.byref(self.shape), _ptr(self.gate_pack), _ptr(qpack), batch, _ptr(index.buf), index.n_items, _ptr(positions), n_cand,
                                                 _ptr(out), out.stride(0), _stream()),
                "rails_mol_score_indexed",
            )
        return out

    def score_candidates(self, qpack: torch.Tensor, batch: int, cand_index: MolIndex, n_cand_padded: int) -> torch.Tensor:
        out = torch.empty((batch, n_cand_padded), dtype=torch.float32, device=cand_index.buf.device)
        with _on_device(cand_index.buf.device):
            _lib.check(
                self.lib.rails_mol_score_candidates(C.byref(self.shape), _ptr(self.gate_pack), _ptr(qpack), batch, _ptr(cand_index.buf), n_cand_padded, _ptr(out), out.stride(0), _stream()),
                "rails_mol_score_candidates",
            )
        return out


    # ---- coarse pass of the two-pass approximate top-k ------------------------------------------------
    def _derived_table(self, fn_name: str, row_elems: int, index: MolIndex, items: Optional[torch.Tensor]) -> torch.Tensor:
        """A bf16 table cut from the fp32 Ex of the index.  In f16x3 precision the index only holds Ex to 22 bits, so the
        table is cut from temporary fp32-format index chunks rebuilt from `items` (same values as the fp32 engine's)."""
        fn = getattr(self.lib, fn_name)
        n, dev = index.n_items, index.buf.device
        table = torch.empty(n * row_elems, dtype=torch.bfloat16, device=dev)
        grouped = fn_name == "rails_mol_component_build"      # item-group-major table: a chunk of items writes a slice of every group
        with _on_device(dev):
            if self.precision == "fp32":
                if grouped:
                    _lib.check(fn(C.byref(self.shape), _ptr(index.buf), n, _ptr(table), n, 0, _stream()), fn_name)
                else:
                    _lib.check(fn(C.byref(self.shape), _ptr(index.buf), n, _ptr(table), _stream()), fn_name)
            else:
                if items is None:
                    raise ValueError(f"{fn_name}: precision='f16x3' needs the raw item embeddings to cut the bf16 table from")
                items = _f32c(items)
                chunk = 1 << 20   # a multiple of the 32-item tile
                tmp = torch.empty(self.lib.rails_mol_index_floats(C.byref(self._fp32_shape), min(chunk, n)), dtype=torch.float32, device=dev)
                for lo in range(0, n, chunk):
                    m = min(chunk, n - lo)
                    _lib.check(self.lib.rails_mol_index_build(C.byref(self._fp32_shape), C.byref(self.weights), _ptr(items[lo : lo + m]), m, _ptr(tmp), _stream()),
                               "rails_mol_index_build")
                    if grouped:
                        _lib.check(fn(C.byref(self._fp32_shape), _ptr(tmp), m, _ptr(table), n, lo, _stream()), fn_name)
                    else:
                        _lib.check(fn(C.byref(self._fp32_shape), _ptr(tmp), m, C.c_void_p(table.data_ptr() + 2 * lo * row_elems), _stream()), fn_name)
        return table

    def build_coarse_table(self, index: MolIndex, items: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(N, d) bf16 table of P_X-averaged component embeddings (reference mol_top_k.py:321-325)."""
        d = self.spec.dot_product_dimension
        return self._derived_table("rails_mol_coarse_build", d, index, items).view(index.n_items, d)

    def coarse_scores(self, eq: torch.Tensor, table: torch.Tensor, average_queries: bool, out: Optional[torch.Tensor] = None,
                      run_if: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eq (B, P_Q, d) fp32 -> (B, N) fp32 holding bf16-rounded dot products (reference mol_top_k.py:351-354)."""
        B, n = eq.shape[0], table.shape[0]
        eq = _f32c(eq)
        if out is None:
            out = torch.empty((B, n), dtype=torch.float32, device=table.device)
        with _on_device(table.device):
            _lib.check(
                self.lib.rails_mol_coarse_score(C.byref(self.shape), _ptr(eq), B, 1 if average_queries else 0, _ptr(table), n, _ptr(out), out.stride(0), _pred(run_if), _stream()),
                "rails_mol_coarse_score",
            )
        return out

    def build_coarse_prefilter(self, table: torch.Tensor) -> Optional[torch.Tensor]:
        """The int8 copy of a coarse table that lets coarse_topk's streaming pass read d instead of 2d bytes per item
        (include/rails_amd.h rails_mol_coarse_prefilter_build); None for shapes without one."""
        n = table.shape[0]
        nbytes = self.lib.rails_mol_coarse_prefilter_bytes(C.byref(self.shape), n)
        if nbytes == 0:
            return None
        pre = torch.empty(nbytes, dtype=torch.uint8, device=table.device)
        with _on_device(table.device):
            _lib.check(self.lib.rails_mol_coarse_prefilter_build(C.byref(self.shape), _ptr(table), n, _ptr(pre), _stream()), "rails_mol_coarse_prefilter_build")
        return pre

    def coarse_topk(self, eq: torch.Tensor, table: torch.Tensor, average_queries: bool, k_prime: int, with_flag: bool = False,
                    prefilter: Optional[torch.Tensor] = None, flag: Optional[torch.Tensor] = None):
        """Fused coarse scoring + exact top-K' (no (B, N) score matrix).  -> (scores (B, K'), positions (B, K'), counts (B,)
        int32) or None when the sizes are unsupported.  The result is exact iff K' <= counts[b] <= capacity for every b
        (see include/rails_amd.h); the caller checks and falls back to coarse_scores + topk otherwise.  with_flag: a fourth
        element, a device int32 that is 1 iff some count is out of range (written by the call's own launches) -- or `flag`, the caller's own
        int32 word (device or PINNED HOST memory: the kernels store through the device-visible address, no copy is needed to read it).
        prefilter: build_coarse_prefilter(table) -- same outputs, the streaming pass reads the int8 copy."""
        B, n = eq.shape[0], table.shape[0]
        memo = self.__dict__.setdefault("_coarse_ws_bytes", {})
        if (B, n, k_prime) not in memo:
            memo[(B, n, k_prime)] = self.lib.rails_mol_coarse_topk_workspace_bytes(C.byref(self.shape), B, n, k_prime)
        ws_bytes = memo[(B, n, k_prime)]
        if ws_bytes == 0:
            return None
        eq = _f32c(eq)
        dev = table.device
        if prefilter is not None:
            # the C side indexes the int8 copy by this call's n and d, and the exactness argument assumes copy and scale belong to THIS
            # table: a buffer of another size or device (a stale copy of a rebuilt table) is refused
            want = memo.get(("prefilter", n))
            if want is None:
                want = memo[("prefilter", n)] = self.lib.rails_mol_coarse_prefilter_bytes(C.byref(self.shape), n)
            if prefilter.device != dev or prefilter.dtype != torch.uint8 or prefilter.numel() != want or not prefilter.is_contiguous():
                raise ValueError(f"coarse_topk: the int8 pre-filter does not belong to this table ({prefilter.numel()} bytes on {prefilter.device}, "
                                 f"expected {want} on {dev}): rebuild it with build_coarse_prefilter(table)")
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out_s = torch.empty((B, k_prime), dtype=torch.float32, device=dev)
        out_p = torch.empty((B, k_prime), dtype=torch.int64, device=dev)
        counts = torch.empty((B + 1,), dtype=torch.int32, device=dev)   # [B]: the out-of-range flag
        if flag is None:
            flag = counts[B:]
        elif flag.dtype != torch.int32 or flag.numel() != 1 or not (flag.is_cuda or flag.is_pinned()):
            raise ValueError("coarse_topk: flag must be one int32 on the device or in pinned host memory")
        else:
            with_flag = True
        with _on_device(dev):
            _lib.check(
                self.lib.rails_mol_coarse_topk(C.byref(self.shape), _ptr(eq), B, 1 if average_queries else 0, _ptr(table), n, k_prime,
                                               _ptr(ws), ws_bytes, _ptr(out_s), _ptr(out_p), _ptr(counts), _ptr(flag) if with_flag else None, _ptr(prefilter), _stream()),
                "rails_mol_coarse_topk",
            )
        return (out_s, out_p, counts[:B], flag) if with_flag else (out_s, out_p, counts[:B])

    @staticmethod
    def coarse_topk_capacity(k_prime: int, n_items: Optional[int] = None, batch: int = 32) -> int:
        """candidates per query the fused coarse top-K' can hold (include/rails_amd.h rails_mol_coarse_topk_capacity); without n_items: the
        figure of shard-sized corpora"""
        if n_items is not None:
            return int(_lib.load().rails_mol_coarse_topk_capacity(int(batch), int(n_items), int(k_prime)))
        cap = min(24576, max(4096, 8 * k_prime))
        return (cap + 63) // 64 * 64

    # ---- per-component candidates (MoLNaiveTopK / MoLCombTopK) -----------------------------------------
    def build_component_table(self, index: MolIndex, items: Optional[torch.Tensor] = None) -> torch.Tensor:
        """(P_X, N, d) bf16 component embeddings (reference mol_top_k.py:61-73), item-group-major: the scans stream one group's rows back to back."""
        px, d = self.spec.item_dot_product_groups, self.spec.dot_product_dimension
        return self._derived_table("rails_mol_component_build", px * d, index, items).view(px, index.n_items, d)

    def component_scores(self, eq: torch.Tensor, table: torch.Tensor, out: Optional[torch.Tensor] = None,
                         run_if: Optional[torch.Tensor] = None) -> torch.Tensor:
        """eq (B, P_Q, d) -> (B * P_Q * P_X, N) fp32 holding bf16 values, row (b * P_Q + i) * P_X + m."""
        B, n = eq.shape[0], table.shape[1]
        eq = _f32c(eq)
        rows = B * self.spec.query_dot_product_groups * self.spec.item_dot_product_groups
        if out is None:
            out = torch.empty((rows, n), dtype=torch.float32, device=table.device)
        with _on_device(table.device):
            _lib.check(
                self.lib.rails_mol_component_score(C.byref(self.shape), _ptr(eq), B, _ptr(table), n, _ptr(out), out.stride(0), _pred(run_if), _stream()),
                "rails_mol_component_score",
            )
        return out

    def component_topk(self, eq: torch.Tensor, table: torch.Tensor, k_group: int, flag: Optional[torch.Tensor] = None):
        """Fused component scoring + exact top-k_group per (b, i, m) row (no (rows, N) score matrix).
        -> (scores (rows, k_group), positions (rows, k_group), counts (rows,) int32) or None when unsupported; exact iff
        k_group <= counts <= component_topk_capacity for every row -- `flag` (an int32 device scalar, zeroed by the call) is raised otherwise."""
        B, n = eq.shape[0], table.shape[1]
        ws_bytes = self.lib.rails_mol_component_topk_workspace_bytes(C.byref(self.shape), B, n, k_group)
        if ws_bytes == 0:
            return None
        eq = _f32c(eq)
        dev = table.device
        rows = B * self.spec.query_dot_product_groups * self.spec.item_dot_product_groups
        memo = self.__dict__.setdefault("_comp_ws", {})          # the workspace is recycled: 30-70 MB of candidate lists per call otherwise
        ws = memo.get(ws_bytes)
        if ws is None or ws.device != dev:
            memo.clear()
            ws = memo[ws_bytes] = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        out_s = torch.empty((rows, k_group), dtype=torch.float32, device=dev)
        out_p = torch.empty((rows, k_group), dtype=torch.int64, device=dev)
        counts = torch.empty((rows,), dtype=torch.int32, device=dev)
        with _on_device(dev):
            _lib.check(
                self.lib.rails_mol_component_topk(C.byref(self.shape), _ptr(eq), B, _ptr(table), n, k_group, _ptr(ws), ws_bytes,
                                                  _ptr(out_s), _ptr(out_p), _ptr(counts), _ptr(flag), _stream()),
                "rails_mol_component_topk",
            )
        return out_s, out_p, counts

    def component_topk_capacity(self, batch: int, n: int, k_group: int) -> int:
        return int(self.lib.rails_mol_component_topk_capacity(C.byref(self.shape), int(batch), int(n), int(k_group)))


def sort_rows(idx: torch.Tensor) -> torch.Tensor:
    """Ascending sort of every row of an int64 (rows, n) tensor, n <= 16384 (torch.sort(dim=1) values)."""
    lib = _lib.load()
    _require_device(idx, "indices")
    idx = idx.to(torch.int64).contiguous()
    out = torch.empty_like(idx)
    with _on_device(idx.device):
        _lib.check(lib.rails_sort_rows_i64(_ptr(idx), idx.shape[0], idx.shape[1], _ptr(out), _stream()), "rails_sort_rows_i64")
    return out


def mask_sorted_duplicates(sorted_idx: torch.Tensor, scores: torch.Tensor, fill: float) -> None:
    """In place: scores[r, j] = fill wherever sorted_idx[r, j] == sorted_idx[r, j - 1]."""
    lib = _lib.load()
    rows, n = sorted_idx.shape
    assert scores.dtype == torch.float32 and scores.stride(1) == 1
    with _on_device(scores.device):
        _lib.check(
            lib.rails_mask_sorted_duplicates(_ptr(sorted_idx), _ptr(scores), scores.stride(0), rows, n, C.c_float(fill), _stream()),
            "rails_mask_sorted_duplicates",
        )


# ---- dot-product (MIPS) scoring ----------------------------------------------------------------
class MipsIndex:
    """Tile-packed fp32 copy of an (N, D) item table for the MFMA dot-product scan."""

    def __init__(self, items: torch.Tensor):
        lib = _lib.load()
        _require_device(items, "item_embeddings")
        items = _f32c(items)
        self.n_items, self.dim = items.shape
        self.buf = torch.empty(lib.rails_mips_index_floats(self.dim, self.n_items), dtype=torch.float32, device=items.device)
        with _on_device(items.device):
            _lib.check(lib.rails_mips_index_build(_ptr(items), self.n_items, self.dim, _ptr(self.buf), _stream()), "rails_mips_index_build")

    def score(self, q: torch.Tensor) -> torch.Tensor:
        """(B, D) -> (B, N) fp32 dot products (reference rails/indexing/mips_top_k.py:72)."""
        lib = _lib.load()
        _require_device(q, "query_embeddings")
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({tuple(q.shape)} and {self.dim}x{self.n_items})")
        q = _f32c(q)
        B = q.shape[0]
        ws = torch.empty(lib.rails_mips_query_ws_floats(self.dim, B), dtype=torch.float32, device=q.device)
        out = torch.empty((B, self.n_items), dtype=torch.float32, device=q.device)
        with _on_device(q.device):
            _lib.check(lib.rails_mips_score(_ptr(q), B, self.dim, _ptr(self.buf), self.n_items, _ptr(ws), _ptr(out), out.stride(0), _stream()), "rails_mips_score")
        return out


def dot_rowwise(q: torch.Tensor, items: torch.Tensor) -> torch.Tensor:
    """q (Bq, D), items (B_I, X, D) with Bq a multiple of B_I -> (Bq, X): <q[bq], items[bq // r][x]>."""
    lib = _lib.load()
    _require_device(q, "query_embeddings")
    q, items = _f32c(q), _f32c(items)
    Bq, D = q.shape
    BI, X, _ = items.shape
    out = torch.empty((Bq, X), dtype=torch.float32, device=q.device)
    with _on_device(q.device):
        _lib.check(lib.rails_dot_rowwise(_ptr(q), _ptr(items), Bq, X, D, Bq // BI, _ptr(out), _stream()), "rails_dot_rowwise")
    return out


# ---- shape-independent kernels ----------------------------------------------------------------
def _pred(flag: Optional[torch.Tensor]):
    """Launch predicate argument (include/rails_amd.h `run_if`): None, or an int32 device scalar read by the kernels when they start."""
    if flag is None:
        return None
    if flag.dtype != torch.int32 or not flag.is_cuda or flag.numel() < 1:
        raise ValueError("the launch predicate is an int32 device tensor")
    return _ptr(flag)


def topk(scores: torch.Tensor, k: int, ids: Optional[torch.Tensor] = None, sorted: bool = True,
         workspace: Optional[torch.Tensor] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
         run_if: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact top-k of every row of `scores` (rows, n) fp32 on the GPU; ties by position ascending.
    ids: None -> positions; (n,) or (1, n) -> shared id row; (rows, n) -> per-row ids.
    Replaces torch.topk + id gather (reference rails/indexing/mol_top_k.py:123-130)."""
    lib = _lib.load()
    _require_device(scores, "scores")
    if scores.dim() != 2:
        raise ValueError("scores must be (rows, n)")
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        scores = _f32c(scores)
    rows, n = scores.shape
    if k > n:
        raise RuntimeError(f"selected index k out of range (k={k}, n={n})")  # what torch.topk raises
    stride = 0
    if ids is not None:
        if ids.dtype != torch.int64 or ids.device != scores.device:
            ids = ids.to(device=scores.device, dtype=torch.int64)
        if ids.dim() == 2 and ids.shape[0] == rows and rows > 1:
            ids = ids.contiguous()
            stride = ids.shape[1]
        else:
            ids = ids.reshape(-1).contiguous()
        if ids.shape[-1] < n:
            raise ValueError("ids has fewer entries than scores has columns")
    if out is not None:      # (rows, k) fp32 / int64, contiguous: overwritten (the predicated fallback of the verified modes)
        out_s, out_i = out
        if out_s.shape != (rows, k) or out_i.shape != (rows, k) or out_s.dtype != torch.float32 or out_i.dtype != torch.int64 or not (out_s.is_contiguous() and out_i.is_contiguous()):
            raise ValueError("topk: out must be contiguous (rows, k) fp32 and int64 tensors")
    else:
        out_s = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
        out_i = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    ws_bytes = lib.rails_topk_workspace_bytes(rows, n, k)
    ws = workspace if workspace is not None and workspace.numel() >= ws_bytes and workspace.device == scores.device else torch.empty(ws_bytes, dtype=torch.uint8, device=scores.device)
    with _on_device(scores.device):
        _lib.check(
            lib.rails_topk(_ptr(scores), scores.stride(0), rows, n, k, 1 if sorted else 0, _ptr(ids), stride, _ptr(out_s), _ptr(out_i), _ptr(ws), ws_bytes, _pred(run_if), _stream()),
            "rails_topk",
        )
    return out_s, out_i


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, silu: bool = False) -> torch.Tensor:
    """act(x @ weight.T + bias) for 2-D fp32 x on the GPU through rails_gemm_f32 (fp32 MFMA); `weight` is a torch Linear weight (N, K)."""
    lib = _lib.load()
    _require_device(x, "x")
    x, weight = _f32c(x), _f32c(weight.detach())
    bias = None if bias is None else _f32c(bias.detach())
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _lib.check(lib.rails_gemm_f32(_ptr(x), K, _ptr(weight), 1, _ptr(bias), None, 0, M, N, K, 1 if silu else 0, None, 0, _ptr(out), N, _stream()), "rails_gemm_f32")
    return out


def gate_combine(logits: torch.Tensor, pair_part: Optional[torch.Tensor], query_part: Optional[torch.Tensor], item_part: Optional[torch.Tensor],
                 items_per_query: int, item_part_per_row: bool, glu_silu: bool, renormalise: bool, eps: float,
                 want_probs: bool = False) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
    """rails_mol_gate_combine: logits / pair_part (rows, L), query_part (rows / X, L), item_part (X or rows, L) -> (out (rows,), pi or None)."""
    lib = _lib.load()
    _require_device(logits, "logits")
    logits = _f32c(logits)
    rows, L = logits.shape
    pair_part = None if pair_part is None else _f32c(pair_part)
    query_part = None if query_part is None else _f32c(query_part)
    item_part = None if item_part is None else _f32c(item_part)
    out = torch.empty((rows,), dtype=torch.float32, device=logits.device)
    probs = torch.empty((rows, L), dtype=torch.float32, device=logits.device) if want_probs else None
    with _on_device(logits.device):
        _lib.check(lib.rails_mol_gate_combine(_ptr(logits), L, _ptr(pair_part), L, _ptr(query_part), _ptr(item_part), rows, items_per_query, L,
                                              1 if item_part_per_row else 0, _lib.RAILS_COMBINE_GLU_SILU if glu_silu else _lib.RAILS_COMBINE_NONE,
                                              1 if renormalise else 0, float(eps), _ptr(out), _ptr(probs), _stream()), "rails_mol_gate_combine")
    return out, probs


def topk_candidates(scores: torch.Tensor, k: int, positions: torch.Tensor, ids: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The final_topk of a candidate rerank in one launch (include/rails_amd.h rails_topk_candidates; reference
    rails/indexing/mol_top_k.py:371-382): top-k of every row of `scores` (rows, n_cand); the id of candidate j of row b is
    ids[positions[b, j]] (ids: the module's flat id row), or positions[b, j] itself without ids."""
    lib = _lib.load()
    _require_device(scores, "scores")
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        scores = _f32c(scores)
    rows, n = scores.shape
    if k > n:
        raise RuntimeError(f"selected index k out of range (k={k}, n={n})")
    positions = positions.to(device=scores.device, dtype=torch.int64).contiguous()
    if positions.shape != (rows, n):
        raise ValueError("positions must be (rows, n_cand) like scores")
    if ids is not None:
        ids = ids.to(device=scores.device, dtype=torch.int64).reshape(-1).contiguous()
    out_s = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    out_i = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    with _on_device(scores.device):
        _lib.check(lib.rails_topk_candidates(_ptr(scores), scores.stride(0), rows, n, k, _ptr(positions), _ptr(ids), _ptr(out_s), _ptr(out_i), _stream()),
                   "rails_topk_candidates")
    return out_s, out_i


def topk_candidates_filterable(n_cand: int, k_prime: int, width: int, k: int) -> bool:
    """sizes rails_topk_candidates_filtered takes (include/rails_amd.h)"""
    return 1024 < n_cand <= 8192 and 0 < k <= k_prime <= min(n_cand, 512) and 0 <= width <= 256


def topk_candidates_filtered(scores: torch.Tensor, k_prime: int, positions: torch.Tensor, ids: Optional[torch.Tensor], invalid_ids: torch.Tensor,
                             k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """topk_candidates(scores, k_prime, positions, ids) followed by filter_seen_ids(..., invalid_ids, k) in one launch (include/rails_amd.h
    rails_topk_candidates_filtered; reference indexing/candidate_index.py:149-175 over a candidate rerank's output).  -> (out_ids (rows, k),
    out_scores (rows, k)), ids first like filter_seen_ids."""
    lib = _lib.load()
    _require_device(scores, "scores")
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        scores = _f32c(scores)
    rows, n = scores.shape
    positions = positions.to(device=scores.device, dtype=torch.int64).contiguous()
    if positions.shape != (rows, n):
        raise ValueError("positions must be (rows, n_cand) like scores")
    if ids is not None:
        ids = ids.to(device=scores.device, dtype=torch.int64).reshape(-1).contiguous()
    inv = invalid_ids.to(device=scores.device, dtype=torch.int64).contiguous()
    if inv.dim() != 2 or inv.shape[0] != rows:
        raise ValueError("invalid_ids must be (rows, width)")
    out_i = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    out_s = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    with _on_device(scores.device):
        _lib.check(lib.rails_topk_candidates_filtered(_ptr(scores), scores.stride(0), rows, n, k_prime, _ptr(positions), _ptr(ids), _ptr(inv), inv.shape[1], k,
                                                      _ptr(out_i), _ptr(out_s), _stream()), "rails_topk_candidates_filtered")
    return out_i, out_s


def rerank_topk_filtered(scores: torch.Tensor, k_prime: int, positions: torch.Tensor, ids: Optional[torch.Tensor], invalid_ids: torch.Tensor, k: int,
                         flag: torch.Tensor, workspace: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The tail of a candidate rerank from UNSORTED candidate positions with duplicates (include/rails_amd.h rails_rerank_topk_filtered): first copy
    of every position, ranked by (score desc, position asc), top k_prime, seen-id filter -> (out_ids (rows, k), out_scores (rows, k)).  `flag`
    (int32, zeroed by the caller; device or pinned host memory) is raised when a row has fewer than k_prime distinct positions -- the outputs are
    then undefined and the caller takes the sorted form."""
    lib = _lib.load()
    _require_device(scores, "scores")
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        scores = _f32c(scores)
    rows, n = scores.shape
    positions = positions.to(device=scores.device, dtype=torch.int64).contiguous()
    if positions.shape != (rows, n):
        raise ValueError("positions must be (rows, n_cand) like scores")
    if ids is not None:
        ids = ids.to(device=scores.device, dtype=torch.int64).reshape(-1).contiguous()
    inv = invalid_ids.to(device=scores.device, dtype=torch.int64).contiguous()
    if inv.dim() != 2 or inv.shape[0] != rows:
        raise ValueError("invalid_ids must be (rows, width)")
    if flag.dtype != torch.int32 or not (flag.is_cuda or flag.is_pinned()):
        raise ValueError("flag must be an int32 tensor on the device or in pinned host memory")
    need = int(lib.rails_rerank_workspace_bytes(rows, n))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=scores.device)
    out_i = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
    out_s = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    with _on_device(scores.device):
        _lib.check(lib.rails_rerank_topk_filtered(_ptr(scores), scores.stride(0), rows, n, k_prime, _ptr(positions), _ptr(ids), _ptr(inv), inv.shape[1], k,
                                                  _ptr(workspace), workspace.numel() * workspace.element_size(), _ptr(out_i), _ptr(out_s), _ptr(flag), _stream()),
                   "rails_rerank_topk_filtered")
    return out_i, out_s


def topk_filter_fusable(n: int, k_prime: int, width: int, k: int) -> bool:
    return bool(_lib.load().rails_topk_filter_fusable(int(n), int(k_prime), int(width), int(k)))


def topk_filtered(scores: torch.Tensor, k_prime: int, ids: Optional[torch.Tensor], invalid_ids: torch.Tensor, k: int,
                  workspace: Optional[torch.Tensor] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                  run_if: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """topk(scores, k_prime, ids) followed by filter_seen_ids(..., invalid_ids, k), the filter fused into the final selection launch
    (include/rails_amd.h rails_topk_filtered).  -> (out_ids (rows, k), out_scores (rows, k)), ids first like filter_seen_ids."""
    lib = _lib.load()
    _require_device(scores, "scores")
    if scores.dtype != torch.float32 or scores.stride(1) != 1:
        scores = _f32c(scores)
    rows, n = scores.shape
    stride = 0
    if ids is not None:
        if ids.dtype != torch.int64 or ids.device != scores.device:
            ids = ids.to(device=scores.device, dtype=torch.int64)
        if ids.dim() == 2 and ids.shape[0] == rows and rows > 1:
            ids = ids.contiguous()
            stride = ids.shape[1]
        else:
            ids = ids.reshape(-1).contiguous()
    invalid_ids = invalid_ids.to(device=scores.device, dtype=torch.int64).contiguous()
    if out is not None:      # (out_ids, out_scores): overwritten (the predicated fallback of the fused score + select path)
        out_i, out_s = out
        if out_i.shape != (rows, k) or out_s.shape != (rows, k) or out_i.dtype != torch.int64 or out_s.dtype != torch.float32 or not (out_i.is_contiguous() and out_s.is_contiguous()):
            raise ValueError("topk_filtered: out must be contiguous (rows, k) int64 and fp32 tensors")
    else:
        out_i = torch.empty((rows, k), dtype=torch.int64, device=scores.device)
        out_s = torch.empty((rows, k), dtype=torch.float32, device=scores.device)
    ws_bytes = lib.rails_topk_workspace_bytes(rows, n, k_prime)
    ws = workspace if workspace is not None and workspace.numel() >= ws_bytes and workspace.device == scores.device else torch.empty(ws_bytes, dtype=torch.uint8, device=scores.device)
    with _on_device(scores.device):
        _lib.check(lib.rails_topk_filtered(_ptr(scores), scores.stride(0), rows, n, k_prime, _ptr(ids), stride, _inv_ptr(invalid_ids), invalid_ids.shape[1], k,
                                           _ptr(out_i), _ptr(out_s), _ptr(ws), ws_bytes, _pred(run_if), _stream()), "rails_topk_filtered")
    return out_i, out_s


def hash_item_table(seed: int, first_item: int, n_items: int, dim: int, device, sigma: float = 0.02, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(n_items, dim) fp32 rows of the synthetic counter-hash item table, drawn on the device: the same bits as
    oracle.mol_oracle.hash_item_table(seed, first_item, n_items, dim, sigma) (include/rails_amd.h rails_hash_item_table)."""
    import math

    import numpy as np

    if out is None:
        out = torch.empty((n_items, dim), dtype=torch.float32, device=device)
    scale = float(np.float32(sigma * math.sqrt(3.0) / 65536.0))
    with _on_device(out.device):
        _lib.check(_lib.load().rails_hash_item_table(seed, first_item, n_items, dim, C.c_float(scale), _ptr(out), _stream()), "rails_hash_item_table")
    return out


def range_flag(values: torch.Tensor, lo: int, hi: int, flag: torch.Tensor) -> None:
    """flag |= any(values < lo or values > hi), on the device (rails_range_flag_i32); `flag` is an int32 device scalar the caller zeroed."""
    lib = _lib.load()
    with _on_device(values.device):
        _lib.check(lib.rails_range_flag_i32(_ptr(values), values.numel(), int(lo), int(hi), _ptr(flag), _stream()), "rails_range_flag_i32")


def rescore_verdict(stats: torch.Tensor, state: torch.Tensor, default_eps: float, safety: float,
                    guard: Optional[torch.Tensor] = None, guard_limit: float = 0.0) -> None:
    """Device-side verdict of a speculative call (rails_rescore_verdict): updates `state` (8 fp32 on the device) in stream order.
    guard: fp32 values (contiguous) whose magnitudes must stay <= guard_limit, else the call is flagged for the redo."""
    lib = _lib.load()
    with _on_device(stats.device):
        _lib.check(lib.rails_rescore_verdict(_ptr(stats), stats.shape[0], float(default_eps), float(safety), _ptr(guard),
                                             0 if guard is None else guard.numel(), float(guard_limit), _ptr(state), _stream()), "rails_rescore_verdict")


def margin_stats(kth_scores: torch.Tensor, col: int, m_max: torch.Tensor, err_max: torch.Tensor) -> torch.Tensor:
    """(rows, >= col + 1) fp32 merged scores, (rows,) fp32, (1,) fp32 -> (rows, 2) row_stats [err_max, kth_scores[:, col] - m_max] (rails_margin_stats)."""
    lib = _lib.load()
    _require_device(kth_scores, "scores")
    rows = kth_scores.shape[0]
    stats = torch.empty((rows, 2), dtype=torch.float32, device=kth_scores.device)
    with _on_device(kth_scores.device):
        _lib.check(lib.rails_margin_stats(_ptr(kth_scores), kth_scores.stride(0), int(col), _ptr(m_max), _ptr(err_max), rows, _ptr(stats), _stream()), "rails_margin_stats")
    return stats


def mfma_probe_f16(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """n x (32 x 16) f16, n x (16 x 32) f16, n x (32 x 32) fp32 -> n x (32 x 32) fp32: one v_mfma_f32_32x32x16_f16 each (rails_mfma_probe_f16)."""
    lib = _lib.load()
    _require_device(a, "a")
    a, b, c = a.to(torch.float16).contiguous(), b.to(torch.float16).contiguous(), _f32c(c)
    n = a.shape[0]
    if tuple(a.shape) != (n, 32, 16) or tuple(b.shape) != (n, 16, 32) or tuple(c.shape) != (n, 32, 32):
        raise ValueError("mfma_probe_f16: a (n, 32, 16), b (n, 16, 32), c (n, 32, 32)")
    d = torch.empty_like(c)
    with _on_device(a.device):
        _lib.check(lib.rails_mfma_probe_f16(_ptr(a), _ptr(b), _ptr(c), _ptr(d), n, _stream()), "rails_mfma_probe_f16")
    return d


def mfma_probe_f32(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor) -> torch.Tensor:
    """n x (32 x 2), n x (2 x 32), n x (32 x 32) fp32 -> n x (32 x 32): one v_mfma_f32_32x32x2_f32 each (rails_mfma_probe_f32)."""
    lib = _lib.load()
    _require_device(a, "a")
    a, b, c = _f32c(a), _f32c(b), _f32c(c)
    n = a.shape[0]
    if tuple(a.shape) != (n, 32, 2) or tuple(b.shape) != (n, 2, 32) or tuple(c.shape) != (n, 32, 32):
        raise ValueError("mfma_probe_f32: a (n, 32, 2), b (n, 2, 32), c (n, 32, 32)")
    d = torch.empty_like(c)
    with _on_device(a.device):
        _lib.check(lib.rails_mfma_probe_f32(_ptr(a), _ptr(b), _ptr(c), _ptr(d), n, _stream()), "rails_mfma_probe_f32")
    return d


def scalar_probe(x: torch.Tensor) -> torch.Tensor:
    """(n,) fp32 -> (3, n): v_exp_f32(x), v_rcp_f32(x), x / (1 + 2^x) as the scoring kernels compute it (rails_scalar_probe_f32)."""
    lib = _lib.load()
    _require_device(x, "x")
    x = _f32c(x).reshape(-1)
    out = torch.empty((3, x.numel()), dtype=torch.float32, device=x.device)
    with _on_device(x.device):
        _lib.check(lib.rails_scalar_probe_f32(_ptr(x), x.numel(), _ptr(out), _stream()), "rails_scalar_probe_f32")
    return out


def rescore_select(exact: torch.Tensor, approx: torch.Tensor, positions: torch.Tensor, ids: Optional[torch.Tensor], n_items: int, k: int,
                   margin_eps: float = float("inf"), check_eps: float = float("inf"),
                   approx_dense: Optional[torch.Tensor] = None, one_sided: bool = False) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Verified finish of a speculative brute-force top-k (include/rails_amd.h rails_rescore_select): exact (rows, >= n_cand) fp32,
    approx (rows, n_ranked), positions (rows, n_cand >= n_ranked; the tail are probes looked up in approx_dense (rows, n_items))
    -> (scores (rows, k), ids (rows, k), row_ok (rows,) int32 for the given eps, row_stats (rows, 2) fp32 = [max |exact - approx|,
    k-th exact - min candidate approx]).  one_sided: approx are upper bounds of the exact scores; the error stat is max(0, exact - approx)."""
    lib = _lib.load()
    _require_device(exact, "exact scores")
    rows, n_cand = positions.shape
    approx, positions = _f32c(approx), positions.to(torch.int64).contiguous()
    out_s = torch.empty((rows, k), dtype=torch.float32, device=exact.device)
    out_i = torch.empty((rows, k), dtype=torch.int64, device=exact.device)
    ok = torch.empty((rows,), dtype=torch.int32, device=exact.device)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=exact.device)
    with _on_device(exact.device):
        _lib.check(lib.rails_rescore_select(_ptr(exact), exact.stride(0), _ptr(approx), _ptr(approx_dense), 0 if approx_dense is None else approx_dense.stride(0),
                                            _ptr(positions), _ptr(ids), n_items, rows, approx.shape[1], n_cand, k, margin_eps, check_eps, 1 if one_sided else 0,
                                            _ptr(out_s), _ptr(out_i), _ptr(ok), _ptr(stats), _stream()), "rails_rescore_select")
    return out_s, out_i, ok, stats


def candidates_workspace(rows: int, device) -> torch.Tensor:
    """The zeroed workspace of candidates_select / candidates_finish for up to `rows` rows (include/rails_amd.h: zeroed once, every
    select + finish pair leaves it zeroed).  Its first `rows` int32 are the per-row candidate counts between the two calls."""
    n = _lib.load().rails_candidates_workspace_bytes(int(rows))
    return torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)


def candidates_select(scores: torch.Tensor, cap: int, lo: float, hi: float, workspace: torch.Tensor, out_pos: torch.Tensor, out_approx: torch.Tensor) -> None:
    """Threshold selection of at most `cap` candidates per row of `scores` (rows, n) fp32 (rails_candidates_select): positions into
    out_pos (rows, >= cap) int64, their scores into out_approx (rows, >= cap) fp32, counts into workspace[:rows]."""
    lib = _lib.load()
    _require_device(scores, "scores")
    rows, n = scores.shape
    if out_pos.stride(0) != out_approx.stride(0) or out_pos.stride(0) < cap:
        raise ValueError("candidates_select: out_pos / out_approx must share a row stride >= cap")
    with _on_device(scores.device):
        _lib.check(lib.rails_candidates_select(_ptr(scores), scores.stride(0), rows, n, int(cap), float(lo), float(hi), _ptr(workspace), _ptr(out_pos), _ptr(out_approx),
                                               out_pos.stride(0), _stream()), "rails_candidates_select")


def candidates_finish(exact: torch.Tensor, approx: torch.Tensor, positions: torch.Tensor, cap: int, workspace: torch.Tensor, ids: Optional[torch.Tensor], n_items: int,
                      k: int, default_eps: float, safety: float, one_sided: bool, guard: Optional[torch.Tensor], guard_per_row: int, guard_limit: float,
                      state: Optional[torch.Tensor], state_host: Optional[torch.Tensor] = None, seen: Optional[Tuple[torch.Tensor, int]] = None,
                      msg: Optional[torch.Tensor] = None):
    """rails_candidates_finish: sort the rows' candidates by (fp32 score, position), write the top k, the verdict and (seen = (invalid_ids, k_out)) the
    seen-id filter's output -> (scores (rows, k), ids (rows, k), f_ids or None, f_scores or None); with `msg` (rows, 2k + 2) int64 the item-sharded
    message is written instead and nothing is returned."""
    lib = _lib.load()
    rows = exact.shape[0]
    dev = exact.device
    out_s = out_i = f_i = f_s = inv = None
    width = f_k = 0
    if msg is None:
        out_s = torch.empty((rows, k), dtype=torch.float32, device=dev)
        out_i = torch.empty((rows, k), dtype=torch.int64, device=dev)
        if seen is not None:
            inv, f_k = seen
            if inv.dtype != torch.int64 or inv.device != dev or not inv.is_contiguous():
                inv = inv.to(device=dev, dtype=torch.int64).contiguous()
            width = inv.shape[1]
            f_i = torch.empty((rows, f_k), dtype=torch.int64, device=dev)
            f_s = torch.empty((rows, f_k), dtype=torch.float32, device=dev)
    with _on_device(dev):
        _lib.check(lib.rails_candidates_finish(_ptr(exact), exact.stride(0), _ptr(approx), _ptr(positions), positions.stride(0), int(cap), _ptr(workspace), _ptr(ids), int(n_items),
                                               rows, int(k), float(default_eps), float(safety), 1 if one_sided else 0, _ptr(guard), int(guard_per_row), float(guard_limit),
                                               _ptr(out_s), _ptr(out_i), None if inv is None else _inv_ptr(inv), width, int(f_k), _ptr(f_i), _ptr(f_s), _ptr(state),
                                               _ptr(state_host), _ptr(msg), _stream()), "rails_candidates_finish")
    return out_s, out_i, f_i, f_s


def pack_candidates(scores: torch.Tensor, ids: torch.Tensor, k: int) -> torch.Tensor:
    """(rows, k_local) fp32 scores + int64 ids -> (rows, 2k) int64 message (score bits | ids), padded with (-inf, -1)."""
    lib = _lib.load()
    _require_device(scores, "scores")
    rows, kl = scores.shape
    scores, ids = _f32c(scores), ids.to(torch.int64).contiguous()
    msg = torch.empty((rows, 2 * k), dtype=torch.int64, device=scores.device)
    with _on_device(scores.device):
        _lib.check(lib.rails_pack_candidates(_ptr(scores), _ptr(ids), rows, kl, k, _ptr(msg), _stream()), "rails_pack_candidates")
    return msg


def merge_candidates(gathered: torch.Tensor, n_ranks: int, k: int, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """gathered (n_ranks * rows, 2k) int64 messages in rank order -> exact top-k_out (scores, ids) per row."""
    lib = _lib.load()
    _require_device(gathered, "gathered messages")
    rows = gathered.shape[0] // n_ranks
    gathered = gathered.contiguous()
    out_s = torch.empty((rows, k_out), dtype=torch.float32, device=gathered.device)
    out_i = torch.empty((rows, k_out), dtype=torch.int64, device=gathered.device)
    with _on_device(gathered.device):
        _lib.check(lib.rails_merge_candidates(_ptr(gathered), n_ranks, rows, k, k_out, _ptr(out_s), _ptr(out_i), _stream()), "rails_merge_candidates")
    return out_s, out_i


def merge_filter_fusable(k_prime: int, width: int, k: int) -> bool:
    return 0 < k <= k_prime <= 512 and 0 <= width <= 256


def merge_candidates_filtered(gathered: torch.Tensor, n_ranks: int, k: int, k_prime: int, invalid_ids: torch.Tensor, k_out: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """merge_candidates(gathered, n_ranks, k, k_prime) followed by filter_seen_ids(..., invalid_ids, k_out) in one launch
    (include/rails_amd.h rails_merge_candidates_filtered) -> (out_ids (rows, k_out), out_scores (rows, k_out))."""
    lib = _lib.load()
    _require_device(gathered, "gathered messages")
    rows = gathered.shape[0] // n_ranks
    gathered = gathered.contiguous()
    invalid_ids = invalid_ids.to(device=gathered.device, dtype=torch.int64).contiguous()
    out_s = torch.empty((rows, k_out), dtype=torch.float32, device=gathered.device)
    out_i = torch.empty((rows, k_out), dtype=torch.int64, device=gathered.device)
    with _on_device(gathered.device):
        _lib.check(lib.rails_merge_candidates_filtered(_ptr(gathered), n_ranks, rows, k, k_prime, _inv_ptr(invalid_ids), invalid_ids.shape[1], k_out,
                                                       _ptr(out_i), _ptr(out_s), _stream()), "rails_merge_candidates_filtered")
    return out_i, out_s


def merge_candidates_verdict(gathered: torch.Tensor, n_ranks: int, k: int, k_out: int, default_eps: float, safety: float, guard: Optional[torch.Tensor],
                             guard_per_row: int, guard_limit: float, state: torch.Tensor, state_host: Optional[torch.Tensor], call_ws: torch.Tensor,
                             seen: Optional[Tuple[torch.Tensor, int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """gathered (n_ranks * rows, 2k + 2) int64 messages of candidates_finish(msg=...) in rank order -> merged exact top-k_out + the global verdict of the
    item-sharded proved top-k in one launch (include/rails_amd.h rails_merge_candidates_verdict).  -> (scores, ids) (rows, k_out), or with
    seen = (invalid_ids, k_f) the filtered (ids, scores) (rows, k_f)."""
    lib = _lib.load()
    _require_device(gathered, "gathered messages")
    rows = gathered.shape[0] // n_ranks
    gathered = gathered.contiguous()
    dev = gathered.device
    inv, width, f_k = None, 0, 0
    cols = k_out
    if seen is not None:
        inv, f_k = seen
        inv = inv.to(device=dev, dtype=torch.int64).contiguous()
        width, cols = inv.shape[1], f_k
    out_s = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    out_i = torch.empty((rows, cols), dtype=torch.int64, device=dev)
    with _on_device(dev):
        _lib.check(lib.rails_merge_candidates_verdict(_ptr(gathered), n_ranks, rows, int(k), int(k_out), float(default_eps), float(safety), _ptr(guard), int(guard_per_row),
                                                      float(guard_limit), _ptr(state), _ptr(state_host), _ptr(call_ws), None if inv is None else _inv_ptr(inv), width, int(f_k),
                                                      _ptr(out_i), _ptr(out_s), _stream()), "rails_merge_candidates_verdict")
    if seen is not None:
        return out_i, out_s
    return out_s, out_i


def filter_seen_ids(top_ids: torch.Tensor, top_scores: torch.Tensor, invalid_ids: torch.Tensor, k: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Row-wise seen-id filter (reference indexing/candidate_index.py:154-178) -> (ids (rows,k), scores (rows,k))."""
    lib = _lib.load()
    _require_device(top_ids, "top_k ids")
    rows, kp = top_ids.shape
    top_ids = top_ids.to(torch.int64).contiguous()
    score_dtype = top_scores.dtype
    top_scores = _f32c(top_scores)
    inv = invalid_ids.to(device=top_ids.device, dtype=torch.int64).contiguous()
    out_i = torch.empty((rows, k), dtype=torch.int64, device=top_ids.device)
    out_s = torch.empty((rows, k), dtype=torch.float32, device=top_ids.device)
    with _on_device(top_ids.device):
        _lib.check(
            lib.rails_filter_seen_ids(_ptr(top_ids), _ptr(top_scores), rows, kp, _inv_ptr(inv), inv.shape[1], k, _ptr(out_i), _ptr(out_s), _stream()),
            "rails_filter_seen_ids",
        )
    return out_i, out_s.to(score_dtype)
