#!/usr/bin/env python3
"""Golden vectors for the HSTU query encoder: runs the REFERENCE `HSTU.encode`
(modeling/sequential/hstu.py, imported from /root/reference, build container only) on seeded inputs and writes
tests/golden/hstu_<config>.npz (inputs, every parameter as an array, the reference's padded outputs and current embeddings).

TEST INFRASTRUCTURE ONLY; nothing of the reference is copied.  The container lacks fbgemm_gpu, whose three jagged LAYOUT
ops the encoder calls (hstu.py:189-210, :513-529, :687).  They move rows and compute nothing, so this generator restates
them from fbgemm's documented semantics, as harness-only shims (like the gin shim of oracle/gen_golden.py):
  asynchronous_complete_cumsum(lengths)                 -> [0, cumsum(lengths)]
  dense_to_jagged(dense (B, N, D), [offsets])            -> rows dense[b, :len_b] concatenated over b
  jagged_to_padded_dense(values, [offsets], [N], pad)    -> (B, N, D), row b = its len_b rows then `pad`
Every arithmetic operation of the fixture outputs is the reference's own code.
  python oracle/gen_golden_hstu.py
"""
import os
import sys
import types

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
gin = types.ModuleType("gin")
gin.configurable = lambda f=None, **kw: (f if f is not None else (lambda g: g))
sys.modules["gin"] = gin
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _npz import savez_deterministic  # noqa: E402
import torch  # noqa: E402


def _cumsum(lengths):
    return torch.cat([torch.zeros(1, dtype=lengths.dtype), torch.cumsum(lengths, 0)])


def _dense_to_jagged(dense, offsets_list, total_L=None):
    off = offsets_list[0]
    rows = [dense[b, : int(off[b + 1] - off[b])] for b in range(dense.shape[0])]
    return [torch.cat(rows, 0), offsets_list]


def _jagged_to_padded_dense(values, offsets, max_lengths, padding_value=0.0):
    off, n = offsets[0], max_lengths[0]
    B = off.numel() - 1
    out = values.new_full((B, n) + tuple(values.shape[1:]), padding_value)
    for b in range(B):
        ln = min(int(off[b + 1] - off[b]), n)
        out[b, :ln] = values[int(off[b]) : int(off[b]) + ln]
    return out


class _Fbgemm:
    asynchronous_complete_cumsum = staticmethod(_cumsum)
    dense_to_jagged = staticmethod(_dense_to_jagged)
    jagged_to_padded_dense = staticmethod(_jagged_to_padded_dense)


class _Ops:   # torch.ops with an `fbgemm` namespace; everything else falls through to the real torch.ops
    fbgemm = _Fbgemm()

    def __getattr__(self, name):
        return getattr(_real_ops, name)


_real_ops = torch.ops
torch.ops = _Ops()

from modeling.sequential.embedding_modules import LocalEmbeddingModule  # noqa: E402  (reference)
from modeling.sequential.hstu import HSTU  # noqa: E402  (reference)
from modeling.sequential.input_features_preprocessors import LearnablePositionalEmbeddingInputFeaturesPreprocessor  # noqa: E402
from modeling.sequential.output_postprocessors import L2NormEmbeddingPostprocessor, LayerNormEmbeddingPostprocessor  # noqa: E402
from rails.similarities.dot_product_similarity_fn import DotProductSimilarity  # noqa: E402  (reference)

from oracle import hstu_oracle as HO  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")


def build_reference(cfg: HO.HSTUConfig, seed: int) -> HSTU:
    torch.manual_seed(seed)
    D = cfg.embedding_dim
    post = (L2NormEmbeddingPostprocessor(embedding_dim=D, eps=1e-6) if cfg.postproc == "l2_norm"
            else LayerNormEmbeddingPostprocessor(embedding_dim=D, eps=1e-6))
    model = HSTU(
        max_sequence_len=cfg.max_sequence_len - 1, max_output_len=1, embedding_dim=D, num_blocks=cfg.num_blocks, num_heads=cfg.num_heads,
        linear_dim=cfg.linear_dim, attention_dim=cfg.attention_dim, normalization="rel_bias", linear_config="uvqk", linear_activation="silu",
        linear_dropout_rate=0.2, attn_dropout_rate=0.0,
        embedding_module=LocalEmbeddingModule(num_items=cfg.num_items, item_embedding_dim=D),
        similarity_module=DotProductSimilarity(),
        input_features_preproc_module=LearnablePositionalEmbeddingInputFeaturesPreprocessor(max_sequence_len=cfg.max_sequence_len, embedding_dim=D, dropout_rate=0.2),
        output_postproc_module=post, enable_relative_attention_bias=True, verbose=False)
    model.eval()
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():   # the reference zero-initialises the output bias; seeded values so that a dropped bias cannot pass
        for name, p in model.named_parameters():
            if name.endswith("_o.bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return model


def make_inputs(cfg: HO.HSTUConfig, B: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    N = cfg.max_sequence_len
    lengths = torch.randint(1, N + 1, (B,), generator=g, dtype=torch.int64)
    lengths[0] = N          # a full row
    lengths[1] = 1          # and the shortest possible
    ids = torch.randint(1, cfg.num_items + 1, (B, N), generator=g, dtype=torch.int64)
    ids = ids * (torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1))
    # timestamps: increasing with gaps from seconds to months, constant over the padding (what the datasets hold)
    gaps = (10.0 ** (torch.rand((B, N), generator=g) * 7.0)).long()
    ts = 1_000_000_000 + torch.cumsum(gaps, 1)
    last = ts[torch.arange(B), lengths - 1].unsqueeze(1)
    ts = torch.where(torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1), ts, last)
    return lengths, ids, ts


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, cfg in HO.HSTU_CONFIGS.items():
        model = build_reference(cfg, seed=11)
        B = 6
        lengths, ids, ts = make_inputs(cfg, B, seed=5)
        with torch.inference_mode():
            emb = model.get_item_embeddings(ids)
            seq, _ = model.generate_user_embeddings(past_lengths=lengths, past_ids=ids, past_embeddings=emb, past_payloads={"timestamps": ts})
            cur = model.encode(past_lengths=lengths, past_ids=ids, past_embeddings=emb, past_payloads={"timestamps": ts})
            cur_nots = model.encode(past_lengths=lengths, past_ids=ids, past_embeddings=emb, past_payloads={})
        sd = {k: v.detach().clone() for k, v in model.state_dict().items() if k.startswith(("_embedding_module", "_input_features_preproc", "_hstu", "_attn_mask"))}
        # the restatement must reproduce the reference before it is trusted as the oracle
        w = {k: v for k, v in sd.items()}
        o_cur, layers = HO.encode(cfg, w, lengths, ids, ts, return_layers=True)
        err = float((o_cur - cur).abs().max())
        err2 = float((HO.encode(cfg, w, lengths, ids, None) - cur_nots).abs().max())
        print(f"{name}: reference vs oracle max|diff| = {err:.3e} (with timestamps), {err2:.3e} (without)")
        assert err < 2e-5 and err2 < 2e-5
        arrays = {"in/past_lengths": lengths.numpy(), "in/past_ids": ids.numpy(), "in/timestamps": ts.numpy(),
                  "out/sequence_embeddings": seq.numpy(), "out/current_embeddings": cur.numpy(),
                  "out/current_embeddings_no_timestamps": cur_nots.numpy(),
                  "meta/torch_version": np.array(torch.__version__)}
        for k, v in sd.items():
            arrays["w/" + k] = v.numpy()
        savez_deterministic(os.path.join(OUT, f"hstu_{name}.npz"), **arrays)
        print(f"  wrote tests/golden/hstu_{name}.npz ({os.path.getsize(os.path.join(OUT, f'hstu_{name}.npz')) // 1024} KB)")


if __name__ == "__main__":
    main()
