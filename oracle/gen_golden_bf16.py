#!/usr/bin/env python3
"""F8: the reference's own bf16 evaluation -- `model.to(torch.bfloat16)` and a bf16 item table, as
eval_batch.py --eval_dtype=bf16 runs it (eval_from_checkpoint.py:320, :392) -- on the amzn-books shape, CPU.  Writes
tests/golden/bf16_books.npz: inputs, fp32 weights, the bf16 run's logits and top-200, and the same module's fp32 logits.

The HIP path keeps every operand of such a module as the bf16-rounded value but does ALL arithmetic in fp32 (or f16x3), so it
sits next to the reference's fp32 run on bf16-rounded weights (<= 1e-4) and away from the reference's bf16 run by that run's own
rounding noise; the fixture lets the tests state that distance.
TEST INFRASTRUCTURE ONLY (arrays in, arrays out).   python oracle/gen_golden_bf16.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG  # noqa: E402  (reference import + shims)

import json  # noqa: E402

import numpy as np  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _npz import savez_deterministic  # noqa: E402
import torch  # noqa: E402

from oracle.mol_oracle import CONFIGS, hash_item_table, synthetic_queries  # noqa: E402


def main():
    cfg = CONFIGS["amzn-books"]
    mol = GG.build_reference_module(cfg, 77)
    N, B = 4096, 16
    X = torch.from_numpy(hash_item_table(41, 0, N, cfg.item_embedding_dim)).unsqueeze(0)
    q = synthetic_queries(cfg, B, seed=42)
    w32 = {k: v.detach().clone().numpy() for k, v in mol.state_dict().items()}
    with torch.inference_mode():
        m16 = GG.build_reference_module(cfg, 77).to(torch.bfloat16)
        l16, _ = m16(q.bfloat16(), X.bfloat16())                      # the reference's bf16 run
        # the same bf16-rounded operands, fp32 arithmetic (what rails_amd computes)
        m32 = GG.build_reference_module(cfg, 77)
        m32.load_state_dict({k: v.bfloat16().float() for k, v in m32.state_dict().items()})
        l32, _ = m32(q.bfloat16().float(), X.bfloat16().float())
        s16, i16 = torch.topk(l16.float(), 200, dim=1)
    out = {"cfg_json": np.array(json.dumps(cfg.to_dict())), "q": q.numpy(), "X": X.numpy(), "logits_bf16_run": l16.float().numpy(),
           "logits_fp32_run_on_bf16_operands": l32.numpy(), "top200_idx_bf16_run": i16.numpy()}
    out.update({"w/" + k: v for k, v in w32.items()})
    savez_deterministic(os.path.join(GG.OUT, "bf16_books.npz"), **out)
    d = (l16.float() - l32).abs()
    print("bf16 run vs fp32 run on the same bf16 operands: max |d| = %.4f, mean %.5f" % (float(d.max()), float(d.mean())))


if __name__ == "__main__":
    main()
