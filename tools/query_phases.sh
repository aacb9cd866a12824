#!/bin/bash
# Debug build of the library with wall-clock phase stamps in the batched prologue's P1 kernel (workgroup 0):
#   tools/query_phases.sh build            (here: cross-compiles rails_amd/librails_amd_phases.so)
#   tools/query_phases.sh run <workload>   (on the GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd rails_amd/csrc
  objs=""
  for f in capi mol_score mol_index mol_coarse mips topk; do objs="$objs $f.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -DRAILS_QUERY_PHASES -c mol_query.hip -o /tmp/mol_query_phases.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/mol_query_phases.o -o ../librails_amd_phases.so
else
  RAILS_PROLOGUE=2 RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_phases.so python - "$2" <<'PY'
import ctypes, sys, torch
sys.path.insert(0, ".")
import bench, rails_amd
from rails_amd import _lib
from oracle import mol_oracle as O
name = sys.argv[1]
cfg_key, N, width = bench.WORKLOADS[name]
cfg = O.CONFIGS[cfg_key]; dev = torch.device("cuda", 0); B = 32
w = O.synthetic_weights(cfg, seed=0)
mol, _ = rails_amd.create_mol_interaction_module(
    cfg.query_embedding_dim, cfg.item_embedding_dim, cfg.dot_product_dimension, cfg.query_dot_product_groups,
    cfg.item_dot_product_groups, cfg.temperature, 0.0, cfg.query_hidden_dim, 0.1, cfg.item_hidden_dim,
    cfg.gating_query_hidden_dim, cfg.gating_qi_hidden_dim, cfg.gating_item_hidden_dim, cfg.softmax_dropout_rate, False,
    query_nonlinearity=cfg.query_nonlinearity, uid_embedding_hash_sizes=list(cfg.uid_embedding_hash_sizes) or None)
mol.load_state_dict(w, strict=True); mol = mol.to(dev).eval()
N = min(N, 100000)
X = torch.from_numpy(O.hash_item_table(1, 0, N, cfg.item_embedding_dim)).unsqueeze(0).to(dev)
ids = torch.arange(1, N + 1, dtype=torch.int64, device=dev).unsqueeze(0)
q = O.synthetic_queries(cfg, B).to(dev)
kw = {"user_ids": torch.randint(0, 1000, (B,), dtype=torch.int64).to(dev)} if len(cfg.uid_embedding_hash_sizes) else {}
lib = _lib.load()
out = (ctypes.c_longlong * 8)()
with torch.inference_mode():
    tk = rails_amd.MoLBruteForceTopK(mol, X, ids)
    eng = tk._bind()
    def show(tag):
        torch.cuda.synchronize(); lib.rails_debug_query_phases(out)
        print(f"{name} [{tag}]: P1 wg0 ticks of 10 ns: loads+mfma {out[1]-out[0]}  partials+barrier {out[2]-out[1]}  tail {out[3]-out[2]}")
    for i in range(3):
        tk(q, k=200, **kw); show("after a scoring pass")
    for i in range(3):
        eng.query_pack(q, kw.get("user_ids")); eng.query_pack(q, kw.get("user_ids")); show("second of two back-to-back prologues")
    big = torch.empty(600_000_000, dtype=torch.uint8, device=dev)
    for i in range(3):
        big.zero_(); eng.query_pack(q, kw.get("user_ids")); show("after a 600 MB memset")
PY
fi
