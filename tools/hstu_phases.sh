#!/bin/bash
# Wall-clock phase stamps of the single-launch HSTU encoder kernel (sequence 0, block 1):
#   tools/hstu_phases.sh build (here)    tools/hstu_phases.sh run (GPU box)
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  cd rails_amd/csrc
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -DRAILS_HSTU_PHASES -c hstu.hip -o /tmp/hstu_phases.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o mol_score.o mol_index.o mol_query.o mol_coarse.o mips.o topk.o /tmp/hstu_phases.o -o ../librails_amd_phases.so
else
  RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_phases.so python - <<'PY'
import ctypes, sys, torch
sys.path.insert(0, ".")
from rails_amd.hstu import HSTU
from rails_amd import _lib
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = HSTU(60, 1, 64, 16, 8, 8, 8, 5000).eval().to(dev)
B, N = 32, 61
g = torch.Generator().manual_seed(1)
lengths = torch.randint(30, N + 1, (B,), generator=g); lengths[0] = N
ids = (torch.randint(1, 5001, (B, N), generator=g) * (torch.arange(N).unsqueeze(0) < lengths.unsqueeze(1))).to(dev)
ts = (1_000_000_000 + torch.cumsum((10.0 ** (torch.rand((B, N), generator=g) * 6)).long(), 1)).to(dev)
lib = _lib.load(); out = (ctypes.c_longlong * 8)()
with torch.inference_mode():
    emb = m.get_item_embeddings(ids)
    for _ in range(3):
        m.encode(lengths.to(dev), ids, emb, {"timestamps": ts}); torch.cuda.synchronize(); lib.rails_debug_hstu_phases(out)
        p = list(out)
        print("block 1 of sequence 0, ticks of 10 ns: LN1 %d | uvqk GEMM %d | attention %d | LN2*u %d | o GEMM %d | block %d" %
              (p[1]-p[0], p[2]-p[1], p[3]-p[2], p[4]-p[3], p[5]-p[4], p[5]-p[0]))
PY
fi
