#!/bin/bash
# Shader-clock phase stamps of the 16x16x64 team scoring kernel (mol_score_wsplit.h; workgroup 0, wave 0, second unit):
#   tools/wsplit_phases.sh build   (here)      tools/wsplit_phases.sh run [precision]   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd rails_amd/csrc
  F="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -DRAILS_WS_PHASES"
  /opt/rocm/bin/hipcc $F -c mol_score.hip -o /tmp/mol_score_wsp.o &
  /opt/rocm/bin/hipcc $F -fno-slp-vectorize -DRAILS_WS_PHASES_F16 -c mol_score_f16.hip -o /tmp/mol_score_f16_wsp.o &
  /opt/rocm/bin/hipcc $F -fno-slp-vectorize -DRAILS_WS_PHASES_F16 -c mol_score_f16x1.hip -o /tmp/mol_score_f16x1_wsp.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o mol_score_extra.o mol_score_f16_extra.o mol_score_f16x1_extra.o mol_index.o mol_query.o mol_coarse.o mips.o topk.o hstu.o \
      /tmp/mol_score_wsp.o /tmp/mol_score_f16_wsp.o /tmp/mol_score_f16x1_wsp.o -o ../librails_amd_phases.so
else
  PREC=${2:-fp32} RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_phases.so python - <<'PY'
import ctypes, os, sys, torch
sys.path.insert(0, ".")
import rails_amd
from rails_amd import _lib
from oracle import mol_oracle as O
prec = os.environ["PREC"]
cfg = O.CONFIGS["synthetic-16x16x64"]; dev = torch.device("cuda", 0); B, N = 32, 200000
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True); mol = mol.to(dev).eval()
mol.precision = None if prec == "fp32" else prec
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).to(dev)
q = O.synthetic_queries(cfg, B).to(dev)
lib = _lib.load(); out = (ctypes.c_longlong * 16)()
fn = {"fp32": "rails_debug_ws_phases", "f16x3": "rails_debug_ws_phases_f16x3", "f16x1": "rails_debug_ws_phases_f16x1"}[prec]
with torch.inference_mode():
    eng = mol.engine(); index = eng.build_index(X); qpack, _, _ = eng.query_pack(q, None)
    for i in range(4):
        eng.score_dense(qpack, B, index); torch.cuda.synchronize(); getattr(lib, fn)(out)
        p = list(out)
        print("%s cycles | GEMM1 %d | cl pack+store+B1 %d | GEMM2 %d | silu+store+B2 %d | GEMM3 q0 %d, GEMM3 q1 || gate q0 %d, end q0 %d, prefetch + gate q1 %d | partials %d | unit %d"
              % (prec, p[1]-p[0], p[2]-p[1], p[3]-p[2], p[4]-p[3], p[7]-p[4], p[8]-p[7], p[9]-p[8], p[5]-p[9], p[6]-p[5], p[6]-p[0]))
PY
fi
