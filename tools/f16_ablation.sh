#!/bin/bash
# In-situ price of each instruction class of the f16x3 scoring kernel: builds librails_amd_abl{1,2,3}.so with
# RAILS_F16_ABLATE = 1 (no MFMAs) / 2 (no transcendentals) / 3 (no VALU arithmetic) and times them next to the real library.
# Results of the ablated builds are wrong by construction.  Build step runs anywhere (hipcc cross-compiles); the timing
# step needs the GPU:  tools/f16_ablation.sh build   |   tools/f16_ablation.sh run [score_bench args]
set -eu
cd "$(dirname "$0")/../rails_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize"
if [ "${1:-build}" = build ]; then
  make -s -j8
  for a in 1 2 3; do
    /opt/rocm/bin/hipcc $FLAGS -DRAILS_F16_ABLATE=$a -c mol_score_f16.hip -o /tmp/mol_score_f16_abl$a.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o mol_score.o mol_score_extra.o mol_score_f16_extra.o /tmp/mol_score_f16_abl$a.o mol_index.o mol_query.o mol_coarse.o mips.o topk.o hstu.o -o ../librails_amd_abl$a.so
  done
  /opt/rocm/bin/hipcc $FLAGS -DRAILS_F16_PHASES -c mol_score_f16.hip -o /tmp/mol_score_f16_ph.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC capi.o mol_score.o mol_score_extra.o mol_score_f16_extra.o /tmp/mol_score_f16_ph.o mol_index.o mol_query.o mol_coarse.o mips.o topk.o hstu.o -o ../librails_amd_phases16.so
else
  shift
  cd ../..
  echo "== real"; python tools/score_bench.py --precision f16x3 "$@"
  for a in 1 2 3; do echo "== RAILS_F16_ABLATE=$a"; RAILS_AMD_LIBRARY=$PWD/rails_amd/librails_amd_abl$a.so python tools/score_bench.py --precision f16x3 "$@"; done
fi
