"""HSTU query encoder, eval path -- the step upstream of the retrieval path.

Mirror of modeling/sequential/hstu.py:HSTU for inference: same constructor arguments that matter at eval time, same
parameter / buffer names (so `load_state_dict` of a reference checkpoint's `module.` entries works unchanged), same
`get_item_embeddings` / `encode` / `forward` signatures.  No fbgemm: the layers run on the padded (B, N, D) tensor with
rows at positions >= length held at zero.  Every floating-point operation runs in the HIP kernels
of csrc/hstu.hip through the C ABI (rails_hstu_preprocess, rails_hstu_time_buckets, rails_rows_layer_norm, rails_gemm_f32,
rails_hstu_attention, rails_rows_normalize); torch only holds the parameters and moves rows (embedding lookup).

Not supported (raises): training mode, the cache / delta_x_offsets decoding path (hstu.py:163-186), `concat_ua`,
`normalization="softmax_rel_bias"`, `linear_activation` other than "silu" / "none".
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import torch

from . import _lib
from .engine import _on_device, _ptr, _stream

TIMESTAMPS_KEY = "timestamps"


def _bucket_thresholds(num_buckets: int, max_dt: int = 1 << 62) -> torch.Tensor:
    """thresholds[b - 1] = the smallest |dt| whose bucket floor(log(max(|dt|, 1)) / 0.301) (float32, as hstu.py:611-613
    evaluates it) is >= b.  The attention kernel counts thresholds <= |dt|, which reproduces torch's bucketing exactly."""
    def bucket(x: int) -> int:
        return int((torch.log(torch.tensor([x]).abs().clamp(min=1)) / 0.301).long())
    out = []
    for b in range(1, num_buckets + 1):
        if bucket(max_dt) < b:
            out.append(max_dt + 1)
            continue
        lo, hi = 1, max_dt
        while lo < hi:
            mid = (lo + hi) // 2
            if bucket(mid) >= b:
                hi = mid
            else:
                lo = mid + 1
        out.append(lo)
    return torch.tensor(out, dtype=torch.int64)


class LocalEmbeddingModule(torch.nn.Module):
    """Reference modeling/sequential/embedding_modules.py:40-73: `_item_emb.weight` (num_items + 1, D), row 0 = padding."""

    def __init__(self, num_items: int, item_embedding_dim: int) -> None:
        super().__init__()
        self._item_embedding_dim = item_embedding_dim
        self._item_emb = torch.nn.Embedding(num_items + 1, item_embedding_dim, padding_idx=0)
        torch.nn.init.trunc_normal_(self._item_emb.weight, mean=0.0, std=0.02, a=-0.04, b=0.04)

    def debug_str(self) -> str:
        return f"local_emb_d{self._item_embedding_dim}"

    def get_item_embeddings(self, item_ids: torch.Tensor) -> torch.Tensor:
        return self._item_emb(item_ids)

    @property
    def item_embedding_dim(self) -> int:
        return self._item_embedding_dim


_ItemEmbedding = LocalEmbeddingModule


class LearnablePositionalEmbeddingInputFeaturesPreprocessor(torch.nn.Module):
    """Reference modeling/sequential/input_features_preprocessors.py:43-92: `_pos_emb.weight` (max_sequence_len, D).  A parameter
    holder here: x = emb * sqrt(D) + pos_emb, masked by id != 0, is evaluated by rails_hstu_preprocess / the fused encoder."""

    def __init__(self, max_sequence_len: int, embedding_dim: int, dropout_rate: float = 0.0) -> None:
        super().__init__()
        self._embedding_dim = embedding_dim
        self._pos_emb = torch.nn.Embedding(max_sequence_len, embedding_dim)
        self._dropout_rate = dropout_rate
        std = (1.0 / embedding_dim) ** 0.5
        torch.nn.init.trunc_normal_(self._pos_emb.weight, mean=0.0, std=std, a=-2 * std, b=2 * std)

    def debug_str(self) -> str:
        return f"posi_d{self._dropout_rate}"

    def forward(self, *args, **kwargs):
        raise NotImplementedError("evaluated inside rails_amd's HSTU encoder kernels")


_PositionalPreproc = LearnablePositionalEmbeddingInputFeaturesPreprocessor


class _Postproc(torch.nn.Module):
    mode = ""

    def __init__(self, embedding_dim: int, eps: float = 1e-6) -> None:
        super().__init__()
        self._embedding_dim = embedding_dim
        self._eps = eps

    def forward(self, *args, **kwargs):
        raise NotImplementedError("evaluated by rails_rows_normalize / the fused encoder kernel")


class L2NormEmbeddingPostprocessor(_Postproc):
    """Reference modeling/sequential/output_postprocessors.py:37-59."""
    mode = "l2_norm"

    def debug_str(self) -> str:
        return "l2"


class LayerNormEmbeddingPostprocessor(_Postproc):
    """Reference modeling/sequential/output_postprocessors.py:62-85."""
    mode = "layer_norm"

    def debug_str(self) -> str:
        return "ln"


class _RelBias(torch.nn.Module):                # RelativeBucketedTimeAndPositionBasedBias (hstu.py:82-138)
    def __init__(self, max_seq_len: int, num_buckets: int) -> None:
        super().__init__()
        self._ts_w = torch.nn.Parameter(torch.empty(num_buckets + 1).normal_(mean=0, std=0.02))
        self._pos_w = torch.nn.Parameter(torch.empty(2 * max_seq_len - 1).normal_(mean=0, std=0.02))


class _Layer(torch.nn.Module):                  # SequentialTransductionUnitJagged (hstu.py:215-437): `_uvqk`, `_o`, `_rel_attn_bias`
    def __init__(self, dim: int, dv: int, dqk: int, heads: int, max_seq_len: int, num_buckets: int, rel_bias: bool) -> None:
        super().__init__()
        self._uvqk = torch.nn.Parameter(torch.empty((dim, dv * 2 * heads + dqk * heads * 2)).normal_(mean=0, std=0.02))
        self._o = torch.nn.Linear(dv * heads, dim)
        torch.nn.init.xavier_uniform_(self._o.weight)
        self._rel_attn_bias = _RelBias(max_seq_len, num_buckets) if rel_bias else None


class _Stack(torch.nn.Module):                  # HSTUJagged: `_attention_layers`
    def __init__(self, layers) -> None:
        super().__init__()
        self._attention_layers = torch.nn.ModuleList(layers)


class HSTU(torch.nn.Module):
    """encode(past_lengths (B,), past_ids (B, N), past_embeddings (B, N, D), past_payloads {"timestamps": (B, N)}) -> (B, D).
    N must equal max_sequence_len + max_output_len (what the reference's eval feeds, modeling/sequential/features.py:48-58)."""

    STRICT_DEVICE_LENGTHS = False   # True: validate device-resident past_lengths too (one blocking device-to-host read per call)

    def __init__(self, max_sequence_len: int, max_output_len: int, embedding_dim: int, num_blocks: int, num_heads: int, linear_dim: int,
                 attention_dim: int, *args, **kwargs) -> None:
        """Two signatures:
          the reference's (modeling/sequential/hstu.py:544-565) -- ..., normalization, linear_config, linear_activation,
            linear_dropout_rate, attn_dropout_rate, embedding_module, similarity_module, input_features_preproc_module,
            output_postproc_module, enable_relative_attention_bias=True, concat_ua=False, verbose=True -- with rails_amd's
            LocalEmbeddingModule / LearnablePositionalEmbeddingInputFeaturesPreprocessor / {L2Norm,LayerNorm}EmbeddingPostprocessor
            (or any objects with the same attributes), so encoder_utils.py needs only its imports swapped;
          the compact one -- ..., num_items, similarity_module=None, normalization="rel_bias", linear_config="uvqk",
            linear_activation="silu", output_postproc="layer_norm", enable_relative_attention_bias=True, concat_ua=False,
            num_buckets=128, eps=1e-6."""
        super().__init__()
        reference_style = "embedding_module" in kwargs or (len(args) > 0 and isinstance(args[0], str))
        if reference_style:
            names = ["normalization", "linear_config", "linear_activation", "linear_dropout_rate", "attn_dropout_rate", "embedding_module",
                     "similarity_module", "input_features_preproc_module", "output_postproc_module", "enable_relative_attention_bias",
                     "concat_ua", "verbose"]
            a = dict(enable_relative_attention_bias=True, concat_ua=False, verbose=True)
        else:
            names = ["num_items", "similarity_module", "normalization", "linear_config", "linear_activation", "output_postproc",
                     "enable_relative_attention_bias", "concat_ua", "num_buckets", "eps"]
            a = dict(similarity_module=None, normalization="rel_bias", linear_config="uvqk", linear_activation="silu",
                     output_postproc="layer_norm", enable_relative_attention_bias=True, concat_ua=False, num_buckets=128, eps=1e-6)
        if len(args) > len(names):
            raise TypeError(f"HSTU() takes at most {7 + len(names)} positional arguments")
        a.update(dict(zip(names, args)))
        for key, v in kwargs.items():
            if key not in names:
                raise TypeError(f"HSTU() got an unexpected keyword argument '{key}'")
            a[key] = v
        missing = [n for n in names if n not in a]
        if missing:
            raise TypeError(f"HSTU() missing required arguments: {missing}")
        normalization, linear_config, linear_activation = a["normalization"], a["linear_config"], a["linear_activation"]
        similarity_module, concat_ua = a["similarity_module"], a["concat_ua"]
        enable_relative_attention_bias = a["enable_relative_attention_bias"]
        if reference_style:
            emb_mod, pre_mod, post_mod = a["embedding_module"], a["input_features_preproc_module"], a["output_postproc_module"]
            if not hasattr(emb_mod, "_item_emb") or not hasattr(pre_mod, "_pos_emb"):
                raise NotImplementedError("HSTU needs a LocalEmbeddingModule-like embedding_module (`_item_emb`) and a "
                                          "LearnablePositionalEmbeddingInputFeaturesPreprocessor-like preprocessor (`_pos_emb`)")
            output_postproc = getattr(post_mod, "mode", None) or {"l2": "l2_norm", "ln": "layer_norm"}.get(post_mod.debug_str())
            num_buckets, eps = 128, float(getattr(post_mod, "_eps", 1e-6))
        else:
            emb_mod = pre_mod = post_mod = None
            output_postproc, num_buckets, eps = a["output_postproc"], a["num_buckets"], a["eps"]
        if normalization not in ("rel_bias", "hstu_rel_bias") or linear_config != "uvqk" or concat_ua:
            raise NotImplementedError("only normalization='rel_bias', linear_config='uvqk', concat_ua=False are built")
        if linear_activation not in ("silu", "none"):
            raise ValueError(f"Unknown linear_activation {linear_activation}")
        if output_postproc not in ("layer_norm", "l2_norm"):
            raise ValueError(f"Unknown output_postproc {output_postproc}")
        self._ndp_module = similarity_module
        self._embedding_dim = embedding_dim
        self._max_sequence_length = max_sequence_len
        self._seq = max_sequence_len + max_output_len
        self._num_blocks, self._num_heads, self._dqk, self._dv = num_blocks, num_heads, attention_dim, linear_dim
        self._linear_activation = linear_activation
        self._postproc = output_postproc
        self._num_buckets = num_buckets
        self._eps = eps
        self._embedding_module = emb_mod if emb_mod is not None else LocalEmbeddingModule(a["num_items"], embedding_dim)
        self._input_features_preproc = pre_mod if pre_mod is not None else LearnablePositionalEmbeddingInputFeaturesPreprocessor(self._seq, embedding_dim)
        self._output_postproc = post_mod if post_mod is not None else (
            LayerNormEmbeddingPostprocessor(embedding_dim, eps) if output_postproc == "layer_norm" else L2NormEmbeddingPostprocessor(embedding_dim, eps))
        self._hstu = _Stack([_Layer(embedding_dim, linear_dim, attention_dim, num_heads, self._seq, num_buckets, enable_relative_attention_bias)
                             for _ in range(num_blocks)])
        self.use_fused_kernel = True    # short sequences: the whole encoder in one launch (falls back when it does not fit)
        self._fused_ptrs = None
        self.register_buffer("_attn_mask", torch.triu(torch.ones((self._seq, self._seq), dtype=torch.bool), diagonal=1))
        self.register_buffer("_bucket_thresholds", _bucket_thresholds(num_buckets), persistent=False)

    # ---- reference API ------------------------------------------------------------------------------------------
    def get_item_embeddings(self, ids: torch.Tensor) -> torch.Tensor:
        return self._embedding_module._item_emb(ids)            # a row gather

    def forward(self, past_lengths, past_ids, past_embeddings, past_payloads: Dict[str, torch.Tensor], batch_id=None) -> torch.Tensor:
        """(B, N, D) postprocessed sequence embeddings (hstu.py:711-739); rows at positions >= length are zero rows
        normalised, exactly as the reference's zero-padded output."""
        x = self._run_layers(past_lengths, past_ids, past_embeddings, past_payloads, min_len=0)
        B, N, D = x.shape
        return self._normalize(x.view(B * N, D), None).view(B, N, D)

    def encode(self, past_lengths, past_ids, past_embeddings, past_payloads: Dict[str, torch.Tensor], delta_x_offsets=None, cache=None,
               return_cache_states: bool = False) -> torch.Tensor:
        """(B, D): the postprocessed embedding at position past_lengths - 1 (hstu.py:741-803)."""
        if delta_x_offsets is not None or cache is not None or return_cache_states:
            raise NotImplementedError("the cached / incremental decoding path is not built")
        if self.use_fused_kernel:
            out = self._encode_fused(past_lengths, past_ids, past_embeddings, past_payloads)
            if out is not None:
                return out
        x = self._run_layers(past_lengths, past_ids, past_embeddings, past_payloads)
        B, N, D = x.shape
        rows = torch.arange(B, device=x.device, dtype=torch.int64) * N + (self._lengths(past_lengths, x.device, N) - 1)
        return self._normalize(x.view(B * N, D), rows)

    def _encode_fused(self, past_lengths, past_ids, past_embeddings, past_payloads) -> Optional[torch.Tensor]:
        """Single-launch encoder for short sequences (rails_hstu_encode_fused): one workgroup per sequence, everything in LDS.
        None when the geometry does not fit (the per-layer kernels then run)."""
        if self.training:
            raise NotImplementedError("rails_amd.HSTU is eval-only: call .eval()")
        if not past_embeddings.is_cuda:
            raise RuntimeError("rails_amd.HSTU runs on the GPU only (no CPU fallback)")
        lib = _lib.load()
        B, N = past_ids.shape
        D, H, dqk, dv = self._embedding_dim, self._num_heads, self._dqk, self._dv
        layers = list(self._hstu._attention_layers)
        if N != self._seq or past_embeddings.shape != (B, N, D) or self._linear_activation != "silu":
            return None
        if not lib.rails_hstu_fused_supported(N, D, H, dqk, dv, self._num_buckets):
            return None
        dev = past_embeddings.device
        ts = past_payloads.get(TIMESTAMPS_KEY) if past_payloads else None
        has_bias = ts is not None and all(l._rel_attn_bias is not None for l in layers)
        if ts is not None and not has_bias and any(l._rel_attn_bias is not None for l in layers):
            return None        # mixed bias / no-bias layers: leave to the general path
        tensors = []           # keep fp32 contiguous views alive until the launch is enqueued
        rows = []
        for l in layers:
            ptrs = []
            for t in (l._uvqk, l._o.weight, l._o.bias) + ((l._rel_attn_bias._ts_w, l._rel_attn_bias._pos_w) if has_bias else ()):
                t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
                tensors.append(t)
                ptrs.append(t.data_ptr())
            rows.append(ptrs + [0] * (5 - len(ptrs)))
        key = tuple(p for r in rows for p in r)
        if self._fused_ptrs is None or self._fused_ptrs[0] != key:
            self._fused_ptrs = (key, torch.tensor(rows, dtype=torch.int64).to(dev))
        ltab = self._fused_ptrs[1]
        lengths = self._lengths(past_lengths, dev, N)
        ids = past_ids.to(device=dev, dtype=torch.int64).contiguous()
        emb = past_embeddings.detach().to(dtype=torch.float32).contiguous()
        pos = self._input_features_preproc._pos_emb.weight.detach().to(device=dev, dtype=torch.float32).contiguous()
        out = torch.empty((B, D), dtype=torch.float32, device=dev)
        with _on_device(dev):
            st = _stream()
            buckets = None
            if has_bias:
                ts = ts.to(device=dev, dtype=torch.int64).contiguous()
                buckets = torch.empty((B, N, N), dtype=torch.uint8, device=dev)
                _lib.check(lib.rails_hstu_time_buckets(_ptr(ts), B, N, _ptr(self._bucket_thresholds.to(dev)), self._num_buckets, _ptr(buckets), st),
                           "rails_hstu_time_buckets")
            _lib.check(lib.rails_hstu_encode_fused(_ptr(emb), _ptr(ids), _ptr(lengths), _ptr(buckets) if has_bias else None, _ptr(pos), _ptr(ltab),
                                                   len(layers), B, N, D, H, dqk, dv, self._num_buckets, 0 if self._postproc == "layer_norm" else 1,
                                                   C.c_float(self._eps), _ptr(out), st), "rails_hstu_encode_fused")
        return out

    # ---- HIP path ------------------------------------------------------------------------------------------------
    @staticmethod
    def _lengths(past_lengths: torch.Tensor, dev, N: int, min_len: int = 1) -> torch.Tensor:
        """int64 lengths on the device, VALIDATED to lie in [min_len, N]: a length beyond the padded width, or an empty history in
        encode() (which indexes row `length - 1`; the reference's flattened gather at offset -1 fails there too, hstu.py:773-781),
        is an upstream data bug and raises instead of returning a plausible embedding of the wrong row.  forward() accepts 0 (an
        all-padding sequence is all zero rows, as in the reference).  Lengths that arrive on the HOST (the data loader's case) are
        checked there, for free; lengths that are already device tensors are clamped into range on the device instead -- reading a
        flag back would be a blocking device-to-host sync on every encode and would rule out stream capture (set
        HSTU.STRICT_DEVICE_LENGTHS = True to pay that sync and raise as for host lengths)."""
        lengths = past_lengths.to(dtype=torch.int64)
        if not lengths.is_cuda or HSTU.STRICT_DEVICE_LENGTHS:
            if bool(((lengths < min_len) | (lengths > N)).any()):
                raise ValueError(f"past_lengths must lie in [{min_len}, {N}] (got min {int(lengths.min())}, max {int(lengths.max())})")
            return lengths.to(device=dev).contiguous()
        lengths = lengths.to(device=dev)
        # sync-free, but not silent: out-of-range lengths are counted in a sticky device counter (HSTU.length_violations() reads it at a
        # moment of the caller's choosing -- end of an eval pass, a stats call), then clamped
        bad = ((lengths < min_len) | (lengths > N)).sum()
        # The counter is replaced, not updated in place: a tensor created under torch.inference_mode() is an inference tensor for ever, and an
        # in-place update of it from a later no_grad / grad-mode caller raises.  (The sum below is a tensor of whichever mode the CALLER is in;
        # a value made outside inference mode takes part in inference-mode arithmetic without complaint, the other way round does not -- so the
        # running total is re-made outside inference mode.)  During stream capture the count is skipped: its storage would belong to the graph's pool.
        if not torch.cuda.is_current_stream_capturing():
            with torch.inference_mode(False), torch.no_grad():
                prev = HSTU._violations.get(dev)
                HSTU._violations[dev] = (bad.clone() if prev is None else prev + bad.clone())
        return lengths.clamp(min=min_len, max=N).contiguous()

    _violations: dict = {}

    @staticmethod
    def length_violations() -> int:
        """Out-of-range past_lengths seen (and clamped) on the sync-free device path since the process started: an upstream data bug when
        non-zero.  One synchronising read per device."""
        return sum(int(v.item()) for v in HSTU._violations.values())

    def _normalize(self, x2d: torch.Tensor, rows: Optional[torch.Tensor]) -> torch.Tensor:
        lib = _lib.load()
        n = x2d.shape[0] if rows is None else rows.numel()
        out = torch.empty((n, x2d.shape[1]), dtype=torch.float32, device=x2d.device)
        with _on_device(x2d.device):
            _lib.check(lib.rails_rows_normalize(_ptr(x2d), x2d.stride(0), _ptr(rows) if rows is not None else None, n, x2d.shape[1],
                                                0 if self._postproc == "layer_norm" else 1, C.c_float(self._eps), _ptr(out), _stream()),
                       "rails_rows_normalize")
        return out

    def _run_layers(self, past_lengths, past_ids, past_embeddings, past_payloads, min_len: int = 1) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("rails_amd.HSTU is eval-only: call .eval()")
        if not past_embeddings.is_cuda:
            raise RuntimeError("rails_amd.HSTU runs on the GPU only (no CPU fallback)")
        lib = _lib.load()
        dev = past_embeddings.device
        B, N = past_ids.shape
        D, H, dqk, dv = self._embedding_dim, self._num_heads, self._dqk, self._dv
        if N != self._seq or past_embeddings.shape != (B, N, D):
            raise ValueError(f"expected past_ids (B, {self._seq}) and past_embeddings (B, {self._seq}, {D})")
        keep = []   # fp32 copies of non-fp32 parameters must outlive the launches that read them (the caching allocator
                    # would otherwise hand the same block to the next conversion before the kernel has run)

        def f32(t):
            t = t.detach().to(device=dev, dtype=torch.float32).contiguous()
            keep.append(t)
            return t

        lengths = self._lengths(past_lengths, dev, N, min_len)
        ids = past_ids.to(device=dev, dtype=torch.int64).contiguous()
        ts = past_payloads.get(TIMESTAMPS_KEY) if past_payloads else None
        if ts is not None:
            ts = ts.to(device=dev, dtype=torch.int64).contiguous()
        emb = f32(past_embeddings)
        M, W = B * N, 2 * H * (dv + dqk)
        x = torch.empty((B, N, D), dtype=torch.float32, device=dev)
        nx = torch.empty((M, D), dtype=torch.float32, device=dev)
        mm = torch.empty((M, W), dtype=torch.float32, device=dev)
        att = torch.empty((M, H * dv), dtype=torch.float32, device=dev)
        oin = torch.empty((M, H * dv), dtype=torch.float32, device=dev)
        thr = self._bucket_thresholds.to(dev)
        has_bias = ts is not None and any(l._rel_attn_bias is not None for l in self._hstu._attention_layers)
        buckets = torch.empty((B, N, N), dtype=torch.uint8, device=dev) if has_bias else None
        with _on_device(dev):
            st = _stream()
            if has_bias:   # the time buckets depend on neither layer nor head: once per call
                _lib.check(lib.rails_hstu_time_buckets(_ptr(ts), B, N, _ptr(thr), self._num_buckets, _ptr(buckets), st), "rails_hstu_time_buckets")
            _lib.check(lib.rails_hstu_preprocess(_ptr(emb), _ptr(ids), _ptr(lengths), _ptr(f32(self._input_features_preproc._pos_emb.weight)),
                                                 B, N, D, C.c_float(float(D) ** 0.5), _ptr(x), st), "rails_hstu_preprocess")
            for layer in self._hstu._attention_layers:
                x2 = x.view(M, D)
                _lib.check(lib.rails_rows_layer_norm(_ptr(x2), D, M, D, C.c_float(self._eps), None, 0, _ptr(nx), D, st), "rails_rows_layer_norm")
                _lib.check(lib.rails_gemm_f32(_ptr(nx), D, _ptr(f32(layer._uvqk)), 0, None, None, 0, M, W, D,
                                              1 if self._linear_activation == "silu" else 0, _ptr(lengths), N, _ptr(mm), W, st), "rails_gemm_f32")
                rb = layer._rel_attn_bias
                use_bias = ts is not None and rb is not None
                _lib.check(lib.rails_hstu_attention(_ptr(mm), W, B, N, H, dqk, dv, _ptr(lengths), _ptr(buckets) if use_bias else None,
                                                    _ptr(f32(rb._ts_w)) if use_bias else None, _ptr(f32(rb._pos_w)) if use_bias else None,
                                                    self._num_buckets if use_bias else 0, _ptr(att), st),
                           "rails_hstu_attention")
                # o_input = u * LN(attn);  u = the first H*dv columns of mm
                _lib.check(lib.rails_rows_layer_norm(_ptr(att), H * dv, M, H * dv, C.c_float(self._eps), _ptr(mm), W, _ptr(oin), H * dv, st),
                           "rails_rows_layer_norm")
                xn = torch.empty((B, N, D), dtype=torch.float32, device=dev)
                _lib.check(lib.rails_gemm_f32(_ptr(oin), H * dv, _ptr(f32(layer._o.weight)), 1, _ptr(f32(layer._o.bias)), _ptr(x2), D, M, D, H * dv,
                                              0, _ptr(lengths), N, _ptr(xn), D, st), "rails_gemm_f32")
                x = xn
        return x
