#!/usr/bin/env python3
"""Shader-clock stamps inside one wave's unit of the f16x3 scoring kernel (debug build with -DRAILS_F16_PHASES,
tools/f16_ablation.sh build).  Prints the cycle deltas between stamps for the overlapped (RAILS_F16_OVERLAP=1: x0, then per
query [stage Y, stage X(next) || epilogue]) and the query-by-query stream (per query [X, Y, epilogue])."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RAILS_AMD_LIBRARY", os.path.join(ROOT, "rails_amd", "librails_amd_phases16.so"))
import torch  # noqa: E402

import bench  # noqa: E402
import rails_amd  # noqa: E402
from oracle import mol_oracle as O  # noqa: E402
from rails_amd import _lib  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "amzn-books"
    cfg_key, N, _ = bench.WORKLOADS[workload]
    if len(sys.argv) > 2:
        N = int(sys.argv[2])
    variants = (("0", "1"),) if cfg_key.endswith("16x16x64") else (("2", "1"), ("2", "0"), ("4", "1"), ("4", "0"), ("5", "1"), ("5", "0"))
    cfg = O.CONFIGS[cfg_key]
    dev = torch.device("cuda:0")
    w = O.synthetic_weights(cfg, seed=0)
    mol, _ = rails_amd.create_mol_interaction_module(
        cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
        cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
        cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
        query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
    mol.load_state_dict(w, strict=True)
    mol = mol.to(dev).eval()
    mol.precision = sys.argv[3] if len(sys.argv) > 3 else "f16x3"    # "f16x1": the one-product build (its own phases library)
    X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).to(dev)
    q = O.synthetic_queries(cfg, 32).to(dev)
    uid = torch.arange(32, dtype=torch.int64, device=dev) if cfg.uid_embedding_hash_sizes else None
    lib = _lib.load()
    lib.rails_debug_f16_phases.argtypes = [C.POINTER(C.c_longlong)]
    with torch.inference_mode():
        eng = mol.engine()
        index = eng.build_index(X)
        qpack, _, _ = eng.query_pack(q, uid)
        for variant, overlap in variants:
            os.environ["RAILS_SCORE_VARIANT"], os.environ["RAILS_F16_OVERLAP"] = variant, overlap
            try:
                for _ in range(2):
                    eng.score_dense(qpack, 32, index)
            except Exception as e:   # variant not available for this shape
                print(f"variant {variant}: {e}")
                continue
            torch.cuda.synchronize()
            buf = (C.c_longlong * 32)()
            lib.rails_debug_f16_phases(buf)
            st = list(buf)[:16]
            print(f"{workload} variant {variant} overlap {overlap}: stamp deltas (shader cycles):", [st[i + 1] - st[i] for i in range(len(st) - 1) if st[i + 1] and st[i]])


if __name__ == "__main__":
    main()
