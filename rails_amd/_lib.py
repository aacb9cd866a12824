"""ctypes binding of librails_amd.so (the C ABI declared in include/rails_amd.h).

There is no fallback: if the shared library is missing or a call fails, this module raises.
Build it with `python -c "import __graft_entry__ as g; g.build()"` or `make -C rails_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAILS_AMD_LIBRARY: load another build of the same library (e.g. the phase-stamp debug build of tools/query_phases.sh)
LIB_PATH = os.environ.get("RAILS_AMD_LIBRARY") or os.path.join(_HERE, "librails_amd.so")

RAILS_ABI_VERSION = 10  # include/rails_amd.h
RAILS_OK = 0
RAILS_EINVAL = -22
RAILS_ENOTSUP = -95
RAILS_ENOMEM = -12
RAILS_ELAUNCH = -5
RAILS_GEGLU = 0
RAILS_SWIGLU = 1
RAILS_MAX_UID_TABLES = 4
RAILS_PRECISION_FP32 = 0
RAILS_PRECISION_F16X3 = 1
RAILS_PRECISION_F16X1 = 2
RAILS_COMBINE_GLU_SILU = 0
RAILS_COMBINE_NONE = 1

_f32p = C.POINTER(C.c_float)


class MolShape(C.Structure):
    _fields_ = [
        ("query_embedding_dim", C.c_int32),
        ("item_embedding_dim", C.c_int32),
        ("dot_product_dimension", C.c_int32),
        ("query_dot_product_groups", C.c_int32),
        ("item_dot_product_groups", C.c_int32),
        ("query_hidden_dim", C.c_int32),
        ("gating_query_hidden_dim", C.c_int32),
        ("gating_item_hidden_dim", C.c_int32),
        ("gating_qi_hidden_dim", C.c_int32),
        ("query_nonlinearity", C.c_int32),
        ("num_uid_tables", C.c_int32),
        ("dot_product_l2_norm", C.c_int32),
        ("temperature", C.c_float),
        ("eps", C.c_float),
        ("precision", C.c_int32),
        ("item_hidden_dim", C.c_int32),
        ("item_nonlinearity", C.c_int32),
        ("gating_combination", C.c_int32),
        ("gating_has_query", C.c_int32),
        ("gating_has_item", C.c_int32),
    ]


class MolWeights(C.Structure):
    _fields_ = [
        ("q_glu_w", C.c_void_p),
        ("q_glu_b", C.c_void_p),
        ("q_proj_w", C.c_void_p),
        ("q_proj_b", C.c_void_p),
        ("uid_table", C.c_void_p * RAILS_MAX_UID_TABLES),
        ("uid_hash_size", C.c_int64 * RAILS_MAX_UID_TABLES),
        ("i_proj_w", C.c_void_p),
        ("i_proj_b", C.c_void_p),
        ("gq_w1", C.c_void_p),
        ("gq_b1", C.c_void_p),
        ("gq_w2", C.c_void_p),
        ("gi_w1", C.c_void_p),
        ("gi_b1", C.c_void_p),
        ("gi_w2", C.c_void_p),
        ("gqi_w1", C.c_void_p),
        ("gqi_b1", C.c_void_p),
        ("gqi_w2", C.c_void_p),
        ("gqi_b2", C.c_void_p),
        ("i_glu_w", C.c_void_p),
        ("i_glu_b", C.c_void_p),
    ]


# name -> (restype, argtypes): one entry per declaration in include/rails_amd.h
_SHAPE_P = C.POINTER(MolShape)
_WEIGHTS_P = C.POINTER(MolWeights)
PROTOTYPES = {
    "rails_last_error": (C.c_char_p, []),
    "rails_device_compute_units": (C.c_int, []),
    "rails_mol_shape_supported": (C.c_int, [_SHAPE_P]),
    "rails_mol_gate_pack_floats": (C.c_size_t, [_SHAPE_P]),
    "rails_mol_pack_gate_weights": (C.c_int, [_SHAPE_P, _WEIGHTS_P, C.c_void_p, C.c_void_p]),
    "rails_mol_index_floats": (C.c_size_t, [_SHAPE_P, C.c_int64]),
    "rails_mol_index_build": (C.c_int, [_SHAPE_P, _WEIGHTS_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_mol_index_unpack": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_mol_index_gather": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_query_pack_floats": (C.c_size_t, [_SHAPE_P, C.c_int32]),
    "rails_mol_query_prologue": (
        C.c_int,
        [_SHAPE_P, _WEIGHTS_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_query_prologue_both": (
        C.c_int,
        [_SHAPE_P, _WEIGHTS_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_score_dense": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_index_rows_floats": (C.c_size_t, [_SHAPE_P, C.c_int64]),
    "rails_mol_index_rows_build": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_mol_score_indexed_rows": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_merge_candidates_verdict": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_float,
                                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_candidates_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "rails_candidates_select": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_candidates_finish": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                          C.c_int32, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_mol_score_dense_upper_supported": (C.c_int, [_SHAPE_P]),
    "rails_mol_score_dense_upper": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_score_candidates": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p],
    ),
    "rails_mips_index_floats": (C.c_size_t, [C.c_int32, C.c_int64]),
    "rails_mips_index_build": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_mips_query_ws_floats": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rails_mips_score": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p],
    ),
    "rails_dot_rowwise": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_coarse_table_bytes": (C.c_size_t, [_SHAPE_P, C.c_int64]),
    "rails_mol_coarse_build": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_mol_coarse_topk_workspace_bytes": (C.c_size_t, [_SHAPE_P, C.c_int32, C.c_int64, C.c_int32]),
    "rails_mol_coarse_topk_capacity": (C.c_int32, [C.c_int32, C.c_int64, C.c_int32]),
    "rails_mol_coarse_topk": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                        C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_mol_coarse_prefilter_bytes": (C.c_size_t, [_SHAPE_P, C.c_int64]),
    "rails_mol_coarse_prefilter_build": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_mol_coarse_score": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "rails_mol_component_table_bytes": (C.c_size_t, [_SHAPE_P, C.c_int64]),
    "rails_mol_component_build": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "rails_mol_component_topk_capacity": (C.c_int32, [_SHAPE_P, C.c_int32, C.c_int64, C.c_int32]),
    "rails_mol_component_topk_workspace_bytes": (C.c_size_t, [_SHAPE_P, C.c_int32, C.c_int64, C.c_int32]),
    "rails_mol_component_topk": (C.c_int, [_SHAPE_P, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_mol_component_score": (
        C.c_int,
        [_SHAPE_P, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p],
    ),
    "rails_hstu_preprocess": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "rails_rows_layer_norm": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_gemm_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_glu_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "rails_hstu_time_buckets": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_hstu_attention": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_hstu_fused_supported": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "rails_hstu_encode_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "rails_rows_normalize": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "rails_sort_rows_i64": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_mask_sorted_duplicates": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_void_p],
    ),
    "rails_topk_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64, C.c_int32]),
    "rails_topk": (
        C.c_int,
        [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p],
    ),
    "rails_topk_candidates": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_topk_candidates_filtered": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                                  C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_rerank_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rails_rerank_topk_filtered": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                              C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_pack_candidates": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_merge_candidates": (
        C.c_int,
        [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rails_rescore_select": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_abi_version": (C.c_int, []),
    "rails_hash_item_table": (C.c_int, [C.c_uint64, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    "rails_mol_gate_combine": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_topk_filter_fusable": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "rails_topk_filtered": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rails_merge_candidates_filtered": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rails_mol_score_indexed_supported": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64]),
    "rails_mol_score_indexed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_range_flag_i32": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_rescore_verdict": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_void_p]),
    "rails_margin_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "rails_mfma_probe_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_mfma_probe_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "rails_scalar_probe_f32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "rails_filter_seen_ids": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
}

_lib: Optional[C.CDLL] = None


class RailsAmdError(RuntimeError):
    """A call into librails_amd.so failed (message from rails_last_error())."""


def load() -> C.CDLL:
    """Load librails_amd.so once.  Fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C rails_amd/csrc`). "
            "rails_amd has no CPU or PyTorch fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    got = lib.rails_abi_version()
    if got != RAILS_ABI_VERSION:
        raise ImportError(f"{LIB_PATH} implements ABI version {got}, this binding was written for {RAILS_ABI_VERSION}: rebuild the library "
                          "(the structs of include/rails_amd.h changed size between versions)")
    _lib = lib
    return lib


def last_error() -> str:
    return (load().rails_last_error() or b"").decode("utf-8", "replace")


def check(code: int, what: str) -> None:
    """Turn a RAILS_E* code into the exception the reference raises in the same situation."""
    if code == RAILS_OK:
        return
    msg = f"{what}: {last_error()} (code {code})"
    if code == RAILS_EINVAL:
        raise ValueError(msg)
    if code == RAILS_ENOTSUP:
        raise NotImplementedError(msg)
    if code == RAILS_ENOMEM:
        raise MemoryError(msg)
    raise RailsAmdError(msg)
