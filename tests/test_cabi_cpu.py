"""CPU: the C-ABI library loads and exports every symbol include/rails_amd.h declares; argument
validation and the host-side mirror work without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import pytest
import torch

import rails_amd
from rails_amd import _lib
from rails_amd import engine as E
from tests._fixtures import Fixture

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    header = open(os.path.join(ROOT, "include", "rails_amd.h")).read()
    declared = set(re.findall(r"\b(rails_[a-z0-9_]+)\s*\(", header))
    declared -= {"rails_mol_shape", "rails_mol_weights", "rails_hstu_layer"}   # structs
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header(lib):
    assert C.sizeof(_lib.MolShape) == 20 * 4
    assert C.sizeof(_lib.MolWeights) == 8 * (4 + 4 + 4 + 12 + 2)  # 26 pointer-sized fields


def test_size_helpers_and_validation(lib):
    s = E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128).to_c()
    assert lib.rails_mol_shape_supported(C.byref(s)) == 1
    # fp32 real-dataset shapes carry the pair-gate weights twice: 32x32x2 fragment order + the small-unit kernel's 16x16x4 order
    assert lib.rails_mol_gate_pack_floats(C.byref(s)) == 2 * (2 * 128 * 64 + 128 + 64)
    assert lib.rails_mol_index_floats(C.byref(s), 33) == 2 * 32 * (8 * 32 + 64)     # two tiles
    assert lib.rails_mol_query_pack_floats(C.byref(s), 5) == 2 * 32 * 32 + 5 * 64 + 32 * (512 + 128 + 8 * 32 + 64)   # two query groups of 4 + scratch rows
    bad = E.MolShapeSpec(64, 64, 40, 8, 8, 512, 128, 128, 128).to_c()
    assert lib.rails_mol_shape_supported(C.byref(bad)) == 0 and "no fused scoring kernel" in _lib.last_error()
    # shapes beyond the tuned ones and the model variants (modeling/similarity_utils.py:41-245)
    for spec in (E.MolShapeSpec(64, 64, 48, 8, 8, 512, 128, 128, 128), E.MolShapeSpec(64, 64, 32, 16, 4, 512, 128, 128, 64 * 2),
                 E.MolShapeSpec(64, 64, 32, 8, 8, -1, 128, 128, 128), E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128, item_hidden_dim=256),
                 E.MolShapeSpec(64, 64, 32, 8, 8, 512, -1, -1, 128, gating_combination_type="none", gating_query_fn=False, gating_item_fn=False)):
        assert lib.rails_mol_shape_supported(C.byref(spec.to_c())) == 1, _lib.last_error()
    none16 = E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128, gating_combination_type="none").to_c("f16x3")
    assert lib.rails_mol_shape_supported(C.byref(none16)) == 1, _lib.last_error()    # "none" is built in the f16 precisions too (round 3)
    glu_missing = E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128, gating_query_fn=False).to_c()
    assert lib.rails_mol_shape_supported(C.byref(glu_missing)) == 0 and "glu_silu needs" in _lib.last_error()
    assert lib.rails_mol_shape_supported(C.byref(E.MolShapeSpec(64, 64, 64, 16, 16, 512, 128, 128, 128).to_c())) == 1
    # precision f16x3: same buffer sizes (f16 hi + lo in the bytes of the fp32 fragment); needs bounded cross logits
    s16 = E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128).to_c("f16x3")
    assert s16.precision == _lib.RAILS_PRECISION_F16X3 and lib.rails_mol_shape_supported(C.byref(s16)) == 1
    assert lib.rails_mol_index_floats(C.byref(s16), 33) == lib.rails_mol_index_floats(C.byref(s), 33)
    assert lib.rails_mol_gate_pack_floats(C.byref(s16)) == 2 * 128 * 64 + 128 + 64    # the f16 builds have no small-unit kernel: one copy
    no_norm = E.MolShapeSpec(64, 64, 32, 8, 8, 512, 128, 128, 128, dot_product_l2_norm=False).to_c("f16x3")
    assert lib.rails_mol_shape_supported(C.byref(no_norm)) == 0 and "dot_product_l2_norm" in _lib.last_error()
    s.precision = 7
    assert lib.rails_mol_shape_supported(C.byref(s)) == 0 and "precision" in _lib.last_error()
    # k > n is rejected before any launch
    assert lib.rails_topk(1, 10, 1, 10, 11, 1, None, 0, 1, 1, None, 0, None, None) == _lib.RAILS_EINVAL
    with pytest.raises(ValueError):
        _lib.check(_lib.RAILS_EINVAL, "x")
    with pytest.raises(NotImplementedError):
        _lib.check(_lib.RAILS_ENOTSUP, "x")


def test_round6_size_helpers_and_host_rules(lib):
    """ABI 10's plan queries answer without a device, and the host-side rules of round 6 are what DESIGN says they are."""
    import rails_amd

    # rails_mol_coarse_topk_capacity: 4 K' on corpora of up to 4 Mi items (dense sample), min(24 576, max(4 096, 8 K')) beyond; 0 = unsupported sizes
    cap = lambda n, kp, b=32: int(lib.rails_mol_coarse_topk_capacity(b, n, kp))
    assert cap(695_762, 4000) == 16000 and cap(695_762, 1000) == 4096 and cap(695_762, 200) == 4096
    assert cap(125_000_000, 1000) == 8000 and cap(125_000_000, 4000) == 24576 and cap(125_000_000, 200) == 4096
    assert cap(695_762, 5000) == 0 and cap(100, 200) == 0 and cap(695_762, 200, 0) == 0
    assert E.MolEngine.coarse_topk_capacity(1000) == 8000 and E.MolEngine.coarse_topk_capacity(1000, 695_762) == 4096
    # rails_rerank_workspace_bytes: one 64-bit key per candidate
    assert lib.rails_rerank_workspace_bytes(32, 6400) == 32 * 6400 * 8 and lib.rails_rerank_workspace_bytes(0, 6400) == 0
    # sizes the filtered candidate selections take (the module composes the two calls elsewhere)
    assert E.topk_candidates_filterable(6400, 181, 61, 120) and E.topk_candidates_filterable(8192, 512, 256, 512)
    assert not E.topk_candidates_filterable(640, 181, 61, 120) and not E.topk_candidates_filterable(8193, 181, 61, 120)
    assert not E.topk_candidates_filterable(6400, 513, 61, 120) and not E.topk_candidates_filterable(6400, 181, 257, 120) and not E.topk_candidates_filterable(6400, 100, 61, 120)
    # argument validation before any launch
    assert lib.rails_topk_candidates_filtered(1, 10, 1, 10, 11, 1, None, None, 0, 5, 1, 1, None) == _lib.RAILS_EINVAL          # k' > n_cand
    assert lib.rails_rerank_topk_filtered(1, 10, 1, 10, 5, 1, None, None, 0, 6, 1, 80, 1, 1, 1, None) == _lib.RAILS_EINVAL      # k > k'
    # the default exact mode speculates by pair count
    T = rails_amd.MoLBruteForceTopK
    assert T.speculation_pays(1, 695_762) and T.speculation_pays(32, 86_971) and T.speculation_pays(16, 27_278)
    assert not T.speculation_pays(8, 27_278) and not T.speculation_pays(1, 200_000) and not T.speculation_pays(32, 3_883)
    # plain Python state of the top-k modules bypasses nn.Module.__setattr__; tensors and modules do not
    class M(rails_amd.topk_modules.TopKModule):
        def forward(self, *a, **k):
            raise NotImplementedError
    m = M()
    m.counter = 3
    m.lin = torch.nn.Linear(2, 2)
    m.register_buffer("buf", torch.zeros(2))
    m.buf = torch.ones(2)
    m.w = torch.nn.Parameter(torch.zeros(1))
    assert m.__dict__["counter"] == 3 and "lin" in m._modules and "w" in m._parameters and torch.equal(m._buffers["buf"], torch.ones(2))
    assert "lin" not in m.__dict__ and "buf" not in m.__dict__ and set(dict(m.named_parameters())) == {"lin.weight", "lin.bias", "w"}


def test_module_mirror_state_dict_and_loud_failure_on_cpu():
    fx = Fixture("c1_ml1m")
    cfg = fx.cfg
    mol, dbg = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes))
    assert dbg.startswith("MoL-8x4x64-t0.05-d0.2-l2-q512d0.0swiglu-id0.1-gq128-gi128d0.0-gqi128d0.0-x-glu_silu-uids6040")
    assert set(mol.state_dict()) == set(fx.weights)          # the reference's keys, exactly
    mol.load_state_dict(fx.weights, strict=True)
    with pytest.raises(NotImplementedError):                 # training mode is out of scope
        mol(fx.t("q"), fx.t("X"), **fx.kw)
    mol.eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # CPU tensors never reach a fallback
        mol(fx.t("q"), fx.t("X"), **fx.kw)
    with pytest.raises(ValueError, match="Invalid top-k method"):
        rails_amd.get_top_k_module("Nope", None, None, None)


def test_checkpoint_key_shim_and_extraction():
    from rails_amd.eval_harness import extract_mol_state_dict

    fx = Fixture("c3_books")
    ckpt = {"module._ndp_module." + k: v for k, v in fx.weights.items()}
    # a legacy checkpoint stores the item projection under _item_proj_module (eval_from_checkpoint.py:366-376)
    for leaf in ("weight", "bias"):
        new = f"module._ndp_module._item_embeddings_fn._item_emb_proj_module.1.{leaf}"
        ckpt[f"module._ndp_module._item_proj_module.1.{leaf}"] = ckpt.pop(new)
    ckpt["module._hstu._some_encoder_weight"] = torch.zeros(3)
    sd = extract_mol_state_dict(ckpt)
    assert set(sd) == set(fx.weights)
    cfg = fx.cfg
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity)
    mol.load_state_dict(sd, strict=True)
