"""Scratch (register spill) budget of the built kernels, read from the code-object metadata of rails_amd/csrc/*.o with the LLVM
tools of the ROCm image (tools/kernel_resources.sh): a kernel that starts spilling -- a changed unroll, a new live range -- shows up
here on the CPU, before anyone times it.  tests/golden/kernel_scratch_ceiling.json lists the kernels that spill today and by how
much; every other kernel must use no scratch at all."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _resources():
    if not (os.path.exists(os.path.join(LLVM, "llvm-readelf")) and os.path.exists(os.path.join(LLVM, "clang-offload-bundler"))):
        pytest.skip("LLVM tools of the ROCm image not found")
    if not os.path.exists(os.path.join(ROOT, "rails_amd", "csrc", "mol_score.o")):
        pytest.skip("objects not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = subprocess.run(["bash", os.path.join(ROOT, "tools", "kernel_resources.sh"), "mol_|coarse|row_select|hstu|gemm|component"],
                         capture_output=True, text=True, timeout=600).stdout
    rows = []
    for line in out.splitlines():
        m = re.match(r"vgpr\s+(\d+) agpr\s+(\d+) sgpr\s+(\d+) scratch\s+(\d+) lds\s+(\d+)\s+(.*)$", line.strip())
        if m:
            rows.append({"vgpr": int(m.group(1)), "scratch": int(m.group(4)), "name": m.group(6)})
    assert len(rows) > 100, out[-2000:]
    return rows


def test_no_kernel_spills_beyond_its_recorded_ceiling():
    ceiling = json.load(open(os.path.join(ROOT, "tests", "golden", "kernel_scratch_ceiling.json")))["ceiling"]
    over = [(r["name"], r["scratch"], ceiling.get(r["name"], 0)) for r in _resources() if r["scratch"] > ceiling.get(r["name"], 0)]
    assert not over, over


def test_hot_kernels_of_the_baseline_shapes():
    rows = {r["name"]: r for r in _resources()}
    pick = lambda pat: [r for n, r in rows.items() if re.search(pat, n)]   # noqa: E731
    small = pick(r"mol_score_small_kernel<")
    assert len(small) == 3 and all(r["scratch"] == 0 and r["vgpr"] <= 128 for r in small)          # four waves per SIMD, no spills
    assert all(r["scratch"] == 0 for r in pick(r"mol_score_wsplit_kernel<"))                         # 16x16x64 (config 4), all precisions
    assert all(r["scratch"] == 0 for r in pick(r"mol_score_\w+_kernel<mol::Fp32Unit, 8, 4, (64|128), 128, 8"))   # ML-1M / ML-20M shapes
    head = pick(r"mol_score_staged_kernel<mol::Fp32Unit, 8, 8, 32, 128, 8>")
    assert len(head) == 1 and head[0]["scratch"] <= 20                                               # the headline kernel (amzn-books, fp32)
