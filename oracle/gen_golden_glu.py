#!/usr/bin/env python3
"""GeGLU / SwiGLU as stand-alone layers: the reference's own modules (rails/similarities/layers.py:19-74) run on CPU in fp32 on
seeded inputs, including a 3-D input (the reshape in forward) and a ragged row count.  Writes tests/golden/glu.npz: x, _w, _b and
the reference output per case.  TEST INFRASTRUCTURE ONLY (arrays in, arrays out).   python oracle/gen_golden_glu.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as GG  # noqa: E402,F401  (puts /root/reference on sys.path + shims)

import numpy as np  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _npz import savez_deterministic  # noqa: E402
import torch  # noqa: E402

from rails.similarities.layers import GeGLU, SwiGLU  # noqa: E402

CASES = [("geglu_2d", GeGLU, (37, 64), 48), ("swiglu_2d", SwiGLU, (130, 256), 128), ("geglu_3d", GeGLU, (3, 5, 96), 40),
         ("swiglu_1row", SwiGLU, (1, 33), 7)]


def main():
    out = {}
    for seed, (name, cls, shape, f_out) in enumerate(CASES):
        torch.manual_seed(100 + seed)
        m = cls(shape[-1], f_out)
        with torch.no_grad():
            m._w.mul_(5.0)                       # N(0, 0.1^2): pre-activations of order 1, so the gate's curvature is exercised
            m._b.normal_(0.0, 0.5)
        x = torch.randn(shape)
        with torch.inference_mode():
            y = m(x)
        out[f"{name}.x"], out[f"{name}.w"], out[f"{name}.b"], out[f"{name}.y"] = x.numpy(), m._w.detach().numpy(), m._b.detach().numpy(), y.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "glu.npz")
    savez_deterministic(path, **out)
    print("wrote", os.path.normpath(path), {k: v.shape for k, v in out.items() if k.endswith(".y")})


if __name__ == "__main__":
    main()
