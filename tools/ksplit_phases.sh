#!/bin/bash
# Wall-clock phase stamps of the 16x16x64 k-split scoring kernel (workgroup 0, wave 0, first unit):
#   tools/ksplit_phases.sh build   (here)      tools/ksplit_phases.sh run   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd rails_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -DRAILS_SCORE_PHASES -c mol_score.hip -o /tmp/mol_score_phases.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o mol_index.o mol_query.o mol_coarse.o mips.o topk.o hstu.o /tmp/mol_score_phases.o -o ../librails_amd_phases.so
else
  RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_phases.so python - <<'PY'
import ctypes, sys, torch
sys.path.insert(0, ".")
import rails_amd
from rails_amd import _lib
from oracle import mol_oracle as O
cfg = O.CONFIGS["synthetic-16x16x64"]; dev = torch.device("cuda", 0); B, N = 32, 200000
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity)
mol.load_state_dict(O.synthetic_weights(cfg, seed=0), strict=True); mol = mol.to(dev).eval()
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
lib = _lib.load(); out = (ctypes.c_longlong * 16)()
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    for i in range(4):
        tk.all_logits(q); torch.cuda.synchronize(); lib.rails_debug_score_phases(out)
        p = list(out)
        print("ticks of 10 ns | first GEMM1 chunk %d | rest of pass 1 (3 GEMM1 chunks + GEMM2) %d | silu %d | half0: GEMM3+gate %d, sweep (2 GEMM1 chunks + mix) %d | half1: %d, %d | unit total %d"
              % (p[1]-p[0], p[3]-p[1], p[4]-p[3], p[5]-p[4], p[6]-p[5], p[7]-p[6], p[8]-p[7], p[8]-p[0]))
PY
fi
