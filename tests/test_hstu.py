"""HSTU query encoder: oracle and HIP path against the REFERENCE's outputs
(tests/golden/hstu_*.npz, written by oracle/gen_golden_hstu.py from modeling/sequential/hstu.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import hstu_oracle as HO

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
NAMES = sorted(HO.HSTU_CONFIGS)
TOL = 2e-5     # fp32 end to end; the outputs are unit-scale (LayerNorm / L2-normalised)


def load(name):
    d = np.load(os.path.join(GOLDEN, f"hstu_{name}.npz"))
    w = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w/")}
    return d, w


def build(cfg, w, dev):
    from rails_amd.hstu import HSTU

    m = HSTU(max_sequence_len=cfg.max_sequence_len - 1, max_output_len=1, embedding_dim=cfg.embedding_dim, num_blocks=cfg.num_blocks,
             num_heads=cfg.num_heads, linear_dim=cfg.linear_dim, attention_dim=cfg.attention_dim, num_items=cfg.num_items,
             output_postproc=cfg.postproc)
    m.load_state_dict(w, strict=True)      # the reference's own names and shapes
    return m.to(dev).eval()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_reference(name):
    d, w = load(name)
    cfg = HO.HSTU_CONFIGS[name]
    lengths, ids, ts = (torch.from_numpy(d[f"in/{k}"]) for k in ("past_lengths", "past_ids", "timestamps"))
    cur, layers = HO.encode(cfg, w, lengths, ids, ts, return_layers=True)
    assert float((cur - torch.from_numpy(d["out/current_embeddings"])).abs().max()) <= 1e-6
    assert float((HO.encode(cfg, w, lengths, ids, None) - torch.from_numpy(d["out/current_embeddings_no_timestamps"])).abs().max()) <= 1e-6
    # the padded sequence output at the valid positions (the reference's padded rows are postprocessed zeros)
    y = layers[-1]
    y = y / torch.clamp(torch.linalg.norm(y, dim=-1, keepdim=True), min=cfg.eps) if cfg.postproc == "l2_norm" else torch.nn.functional.layer_norm(y, [cfg.embedding_dim], eps=cfg.eps)
    valid = torch.arange(cfg.max_sequence_len).unsqueeze(0) < lengths.unsqueeze(1)
    assert float((y - torch.from_numpy(d["out/sequence_embeddings"]))[valid].abs().max()) <= 1e-6


def test_module_mirrors_the_reference_state_dict_and_bucketing():
    from rails_amd.hstu import HSTU, _bucket_thresholds

    for name in NAMES:
        d, w = load(name)
        cfg = HO.HSTU_CONFIGS[name]
        m = HSTU(cfg.max_sequence_len - 1, 1, cfg.embedding_dim, cfg.num_blocks, cfg.num_heads, cfg.linear_dim, cfg.attention_dim, cfg.num_items,
                 output_postproc=cfg.postproc)
        sd = m.state_dict()
        assert sorted(sd) == sorted(w) and all(tuple(sd[k].shape) == tuple(w[k].shape) for k in w)
    thr = _bucket_thresholds(128)
    assert torch.equal(thr, HO.bucket_thresholds(128))
    g = torch.Generator().manual_seed(0)
    dt = torch.cat([torch.arange(0, 5000), (10.0 ** (torch.rand(20000, generator=g) * 11)).long(), thr, thr - 1, thr + 1])
    by_table = (thr.unsqueeze(0) <= dt.abs().unsqueeze(1)).sum(1)
    assert torch.equal(by_table, HO.bucketize(dt, 128))
    with pytest.raises(NotImplementedError):
        HSTU(10, 1, 16, 1, 1, 8, 8, 20, concat_ua=True)


def test_reference_style_constructor_builds_the_same_module():
    """modeling/sequential/encoder_utils.py constructs HSTU from module objects (hstu.py:544-565); with rails_amd's classes of
    the same names that call works unchanged and yields the same parameters as the compact constructor."""
    from rails_amd.hstu import HSTU
    from rails_amd.modeling.sequential.embedding_modules import LocalEmbeddingModule
    from rails_amd.modeling.sequential.input_features_preprocessors import LearnablePositionalEmbeddingInputFeaturesPreprocessor
    from rails_amd.modeling.sequential.output_postprocessors import L2NormEmbeddingPostprocessor, LayerNormEmbeddingPostprocessor

    for post in (L2NormEmbeddingPostprocessor(embedding_dim=16, eps=1e-6), LayerNormEmbeddingPostprocessor(embedding_dim=16, eps=1e-6)):
        ref_style = HSTU(
            max_sequence_len=10, max_output_len=1, embedding_dim=16, num_blocks=2, num_heads=1, linear_dim=8, attention_dim=8,
            normalization="rel_bias", linear_config="uvqk", linear_activation="silu", linear_dropout_rate=0.2, attn_dropout_rate=0.0,
            embedding_module=LocalEmbeddingModule(num_items=20, item_embedding_dim=16), similarity_module=None,
            input_features_preproc_module=LearnablePositionalEmbeddingInputFeaturesPreprocessor(max_sequence_len=11, embedding_dim=16, dropout_rate=0.2),
            output_postproc_module=post, enable_relative_attention_bias=True, verbose=False)
        compact = HSTU(10, 1, 16, 2, 1, 8, 8, 20, output_postproc=post.mode)
        assert sorted(ref_style.state_dict()) == sorted(compact.state_dict())
        assert all(tuple(ref_style.state_dict()[k].shape) == tuple(v.shape) for k, v in compact.state_dict().items())
        assert ref_style._postproc == post.mode and ref_style._output_postproc is post
    positional = HSTU(10, 1, 16, 2, 1, 8, 8, "rel_bias", "uvqk", "silu", 0.2, 0.0, LocalEmbeddingModule(20, 16), None,
                      LearnablePositionalEmbeddingInputFeaturesPreprocessor(11, 16, 0.2), L2NormEmbeddingPostprocessor(16))
    assert positional._postproc == "l2_norm"
    with pytest.raises(TypeError):
        HSTU(10, 1, 16, 2, 1, 8, 8, 20, no_such_argument=1)


def test_encoder_refuses_cpu_and_training():
    from rails_amd.hstu import HSTU

    m = HSTU(10, 1, 16, 1, 1, 8, 8, 20).eval()
    ids = torch.ones((2, 11), dtype=torch.int64)
    with pytest.raises(RuntimeError, match="GPU only"):
        m.encode(torch.tensor([3, 4]), ids, m.get_item_embeddings(ids), {})
    m.train()
    with pytest.raises(NotImplementedError, match="eval-only"):
        m.encode(torch.tensor([3, 4]), ids, m.get_item_embeddings(ids), {})


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_encoder_matches_the_reference(name):
    dev = torch.device("cuda", 0)
    d, w = load(name)
    cfg = HO.HSTU_CONFIGS[name]
    m = build(cfg, w, dev)
    lengths, ids, ts = (torch.from_numpy(d[f"in/{k}"]).to(dev) for k in ("past_lengths", "past_ids", "timestamps"))
    with torch.inference_mode():
        emb = m.get_item_embeddings(ids)
        cur = m.encode(lengths, ids, emb, {"timestamps": ts})
        cur_nots = m.encode(lengths, ids, emb, {})
        seq = m(lengths, ids, emb, {"timestamps": ts})
    assert float((cur.cpu() - torch.from_numpy(d["out/current_embeddings"])).abs().max()) <= TOL
    assert float((cur_nots.cpu() - torch.from_numpy(d["out/current_embeddings_no_timestamps"])).abs().max()) <= TOL
    valid = (torch.arange(cfg.max_sequence_len).unsqueeze(0) < lengths.cpu().unsqueeze(1))
    assert float((seq.cpu() - torch.from_numpy(d["out/sequence_embeddings"]))[valid].abs().max()) <= TOL
    # determinism
    with torch.inference_mode():
        assert torch.equal(cur, m.encode(lengths, ids, emb, {"timestamps": ts}))
    # the two implementations of encode (single-launch kernel for seq_len <= 64, per-layer kernels) against each other
    from rails_amd import _lib
    fits = bool(_lib.load().rails_hstu_fused_supported(cfg.max_sequence_len, cfg.embedding_dim, cfg.num_heads, cfg.attention_dim, cfg.linear_dim, 128))
    assert fits == (name == "amzn-books")   # ML-1M: D = 50 is not a multiple of 32; the ML-20M fixture: D = 256 does not fit LDS
    with torch.inference_mode():
        m.use_fused_kernel = False
        per_layer = m.encode(lengths, ids, emb, {"timestamps": ts})
        per_layer_nots = m.encode(lengths, ids, emb, {})
    assert float((per_layer.cpu() - torch.from_numpy(d["out/current_embeddings"])).abs().max()) <= TOL
    assert float((per_layer_nots.cpu() - torch.from_numpy(d["out/current_embeddings_no_timestamps"])).abs().max()) <= TOL
    assert float((per_layer - cur).abs().max()) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_bf16_module_equals_fp32_module_with_rounded_weights(name):
    """The reference evaluates with model.to(bfloat16) (eval_from_checkpoint.py:320).  The HIP path upcasts such parameters to
    fp32 copies per call; the copies must stay alive until the launches that read them are enqueued (a freed copy's block
    is handed to the next conversion).  A bf16 module must therefore equal, bit for bit, an fp32 module holding the same
    bf16-rounded values -- on the per-layer path (many conversions per call) and on the fused one."""
    dev = torch.device("cuda", 0)
    d, w = load(name)
    cfg = HO.HSTU_CONFIGS[name]
    w_rounded = {k: (v.to(torch.bfloat16).float() if v.is_floating_point() else v) for k, v in w.items()}
    m16 = build(cfg, w, dev).to(torch.bfloat16)
    m32 = build(cfg, w_rounded, dev)
    lengths, ids, ts = (torch.from_numpy(d[f"in/{k}"]).to(dev) for k in ("past_lengths", "past_ids", "timestamps"))
    with torch.inference_mode():
        emb32 = m32.get_item_embeddings(ids)
        emb16 = m16.get_item_embeddings(ids)
        assert torch.equal(emb16.float(), emb32)
        for fused in (True, False):
            m16.use_fused_kernel = m32.use_fused_kernel = fused
            for _ in range(3):   # repeated calls recycle the allocator's blocks
                a = m16.encode(lengths, ids, emb16, {"timestamps": ts})
                b = m32.encode(lengths, ids, emb32, {"timestamps": ts})
                assert torch.equal(a, b), (name, fused, float((a - b).abs().max()))


@pytest.mark.gpu
def test_encode_rejects_out_of_range_lengths():
    """past_lengths outside [1, N] are an upstream data bug: encode raises instead of returning the embedding of a clamped row
    (the reference fails on both: its gather at offset length - 1 leaves the buffer)."""
    dev = torch.device("cuda", 0)
    d, w = load("amzn-books")
    cfg = HO.HSTU_CONFIGS["amzn-books"]
    m = build(cfg, w, dev)
    lengths, ids = (torch.from_numpy(d[f"in/{k}"]).to(dev) for k in ("past_lengths", "past_ids"))
    with torch.inference_mode():
        emb = m.get_item_embeddings(ids)
        for fused in (True, False):
            m.use_fused_kernel = fused
            for pos, val in ((0, 0), (1, cfg.max_sequence_len + 5), (2, -3)):
                bad = lengths.clone()
                bad[pos] = val
                with pytest.raises(ValueError, match="past_lengths"):
                    m.encode(bad.cpu(), ids, emb, {})          # lengths from the host (the data loader's case): checked there
                # device-resident lengths are clamped on the device (no blocking read in the hot path) unless strict mode is on
                clamped = bad.clamp(min=1, max=ids.shape[1])
                before = type(m).length_violations()
                assert torch.equal(m.encode(bad, ids, emb, {}), m.encode(clamped, ids, emb, {}))
                assert type(m).length_violations() > before               # ... and counted: the clamp is not silent
                type(m).STRICT_DEVICE_LENGTHS = True
                try:
                    with pytest.raises(ValueError, match="past_lengths"):
                        m.encode(bad, ids, emb, {})
                finally:
                    type(m).STRICT_DEVICE_LENGTHS = False
            ok = lengths.clone()
            ok[0], ok[1] = 1, cfg.max_sequence_len
            assert m.encode(ok, ids, emb, {}).shape == (lengths.shape[0], cfg.embedding_dim)
        empty = lengths.clone()
        empty[0] = 0
        assert m(empty, ids, emb, {}).shape[0] == lengths.shape[0]       # forward() accepts an all-padding sequence
    # the sticky counter survives callers of every autograd mode (round-5 advice: a counter created under inference_mode used to be an
    # inference tensor, and the next no_grad caller's in-place update of it raised)
    bad = lengths.clone()
    bad[0] = 0
    before = type(m).length_violations()
    with torch.inference_mode():
        m.encode(bad, ids, emb.clone(), {})
    with torch.no_grad():
        m.encode(bad, ids, emb.clone(), {})
    with torch.inference_mode():
        m.encode(bad, ids, emb.clone(), {})
    assert type(m).length_violations() >= before + 3       # (every length check of a call counts)


@pytest.mark.gpu
def test_hip_encoder_full_width_against_the_oracle():
    """The real ML-1M encoder geometry (8 blocks, 2 heads x 25, N = 211, B = 32): seven key tiles, the dv = 25 padding."""
    dev = torch.device("cuda", 0)
    cfg = HO.HSTUConfig(max_sequence_len=211, embedding_dim=50, num_blocks=8, num_heads=2, attention_dim=25, linear_dim=25, num_items=3883)
    from rails_amd.hstu import HSTU

    torch.manual_seed(3)
    m = HSTU(210, 1, 50, 8, 2, 25, 25, 3883).eval()
    with torch.no_grad():
        for n_, p in m.named_parameters():
            if n_.endswith("_o.bias"):
                p.normal_(0, 0.05)
    w = {k: v.detach().clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(1)
    B, N = 32, 211
    lengths = torch.randint(1, N + 1, (B,), generator=g)
    lengths[0] = N
    ids = torch.randint(1, 3884, (B, N), generator=g) * (torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1))
    ts = 1_000_000_000 + torch.cumsum((10.0 ** (torch.rand((B, N), generator=g) * 6)).long(), 1)
    ref = HO.encode(cfg, w, lengths, ids, ts)
    m = m.to(dev)
    with torch.inference_mode():
        cur = m.encode(lengths.to(dev), ids.to(dev), m.get_item_embeddings(ids.to(dev)), {"timestamps": ts.to(dev)})
    assert float((cur.cpu() - ref).abs().max()) <= 5e-5


@pytest.mark.gpu
def test_sequences_to_metrics_end_to_end():
    """Ranks 3 + 4 of SURVEY.md section 8(f) chained, all in HIP: interaction sequences -> HSTU encoder -> MoL brute-force
    top-k with seen-id filter -> the harness's HR / NDCG / MRR; against the oracle chain (HSTU oracle -> MoL oracle)."""
    import rails_amd
    from oracle import mol_oracle as O
    from rails_amd import eval_harness as H
    from rails_amd.hstu import HSTU

    dev = torch.device("cuda", 0)
    mcfg = O.CONFIGS["amzn-books"]
    n_items, N, B = 3000, 51, 24
    hcfg = HO.HSTUConfig(max_sequence_len=N, embedding_dim=64, num_blocks=2, num_heads=8, attention_dim=8, linear_dim=8, num_items=n_items)
    mol, _ = rails_amd.create_mol_interaction_module(
        mcfg.query_embedding_dim, mcfg.item_embedding_dim, mcfg.dot_product_dimension, mcfg.query_dot_product_groups,
        mcfg.item_dot_product_groups, mcfg.temperature, 0.0, mcfg.query_hidden_dim, 0.1, mcfg.item_hidden_dim,
        mcfg.gating_query_hidden_dim, mcfg.gating_qi_hidden_dim, mcfg.gating_item_hidden_dim, mcfg.softmax_dropout_rate, False,
        query_nonlinearity=mcfg.query_nonlinearity)
    mw = O.synthetic_weights(mcfg, seed=4)
    mol.load_state_dict(mw, strict=True)
    torch.manual_seed(9)
    model = HSTU(N - 1, 1, 64, 2, 8, 8, 8, n_items, similarity_module=mol).eval()
    with torch.no_grad():   # item embeddings at the scale the MoL item tower was initialised for; non-zero output biases
        for n_, p in model.named_parameters():
            if n_.endswith("_o.bias"):
                p.normal_(0, 0.05)
    hw = {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("_ndp_module")}
    g = torch.Generator().manual_seed(2)
    lengths = torch.randint(2, N, (B,), generator=g)
    ids = torch.stack([torch.randperm(n_items, generator=g)[:N] + 1 for _ in range(B)]) * (torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1))
    ts = 1_000_000_000 + torch.cumsum((10.0 ** (torch.rand((B, N), generator=g) * 6)).long(), 1)
    target = torch.randint(1, n_items + 1, (B, 1), generator=g)

    # oracle chain
    q_ref = HO.encode(hcfg, hw, lengths, ids, ts)
    all_ids = torch.arange(1, n_items + 1, dtype=torch.int64)
    X = hw["_embedding_module._item_emb.weight"][all_ids].unsqueeze(0)
    k = 120
    kp = O.k_prime(k, ids, n_items, 200)
    rs, ri, _ = O.brute_force_topk(mcfg, mw, q_ref, X, all_ids.unsqueeze(0), kp)
    ref_ids, _ = O.filter_seen_ids(ri, rs, ids, k)
    ref_metrics = O.eval_metrics(ref_ids, target, 120)

    model = model.to(dev)
    with torch.inference_mode():
        state = H.get_eval_state(model, all_ids.tolist(), None, lambda emb, eids: rails_amd.MoLBruteForceTopK(model._ndp_module, emb, eids), dev)
        feats = H.SequentialFeatures(lengths.to(dev), ids.to(dev), None, {"timestamps": ts.to(dev)})
        random_state = __import__("random").getstate()
        out = H.eval_metrics_v2_from_tensors(state, model, feats, target.to(dev), include_eval_time=True, include_eval_top_k_ids=True)
        __import__("random").setstate(random_state)
    got = out["eval_top_k_ids"].cpu()
    agree = float((got == ref_ids).float().mean())
    assert agree >= 0.98, agree            # near-tie swaps only (fp32 summation order in two chained models)
    for key in ("hr@10", "hr@50", "hr@100"):
        assert float((out[key].cpu() != ref_metrics[key]).float().mean()) <= 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,seq,small", [(6752, 1024, 256, 211, 211), (300, 64, 32, 50, 200), (320, 192, 96, 64, 192), (1000, 260, 64, 50, 200)])
@pytest.mark.parametrize("w_is_nk", [1, 0])
def test_tiled_gemm_equals_the_per_wave_kernel(M, N, K, seq, small, w_is_nk):
    """rails_gemm_f32 takes the LDS-tiled kernel for M >= 256 rows with aligned K; the per-wave kernel for fewer rows.  Same operand
    assignment and order over k in both, so the first rows computed alone (fewer than 256: per-wave) equal the same rows of the big call
    (tiled) bit for bit; both sit within fp32 rounding of a float64 product; padded rows are written as zeros."""
    import ctypes as C

    from rails_amd import _lib
    from rails_amd.engine import _ptr, _stream

    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + N + K + w_is_nk)
    x = torch.randn((M, K), generator=g).to(dev)
    w = (torch.randn((N, K) if w_is_nk else (K, N), generator=g) / K ** 0.5).to(dev)
    bias = torch.randn((N,), generator=g).to(dev)
    res = torch.randn((M, N), generator=g).to(dev)
    lengths = torch.randint(1, seq + 1, (M // seq,), generator=g).to(dev)

    def run(rows):
        out = torch.full((rows, N), float("nan"), device=dev)
        _lib.check(lib.rails_gemm_f32(_ptr(x), K, _ptr(w), w_is_nk, _ptr(bias), _ptr(res), N, rows, N, K, 1, _ptr(lengths), seq, _ptr(out), N, _stream()), "rails_gemm_f32")
        return out

    big, part = run(M), run(small)
    assert torch.equal(big[:small], part)
    ref = x.double() @ (w.double().T if w_is_nk else w.double()) + bias.double()
    ref = ref * torch.sigmoid(ref) + res.double()
    pos = torch.arange(M, device=dev)
    pad = (pos % seq) >= lengths[pos // seq]
    ref[pad] = 0.0
    assert float((big.double() - ref).abs().max()) < 5e-5 and bool((big[pad] == 0).all())
