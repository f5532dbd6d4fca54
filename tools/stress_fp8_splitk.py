"""Stress: the fp8 weight-streaming kernel's split-K meeting right behind its int8 twin on the same workspace slots (the order of
tests/test_fuzz_gpu.py::test_int8_fp8_linear_shape_sweep at (33, 4096, 4096)), optionally after the MX stream-K kernel has grown and
used the workspace.  Counts launches whose output differs from the first one (the meeting adds its parts in part order: every launch
must give the same bits) and says where.

    python tools/stress_fp8_splitk.py [--iters 3000] [--mx-first 1]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--mx-first", type=int, default=1)
ap.add_argument("--shape", default="33,4096,4096")
args = ap.parse_args()
dev = "cuda"
lib = _lib.lib()
if args.mx_first:
    for variant in (119, 118, 114, 129, 128, 113):
        lib.ao_gemm8_set_variant(variant)
        for sizes, n, k in (([16, 16, 16, 16], 256, 4096), ([3, 0, 0, 1], 64, 14336), ([32, 0, 0, 0, 32, 64, 0, 0], 1024, 4096), ([2] * 64, 48, 384)):
            a = torch.randn(sum(sizes), k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(len(sizes), n, k, device=dev, dtype=torch.bfloat16) * 0.1
            aq, a_s = ops.mxfp8_quantize(a, "rceil")
            wq, ws = ops.mxfp8_quantize(w, "rceil")
            offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=dev)
            for _ in range(3):
                ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs)
    lib.ao_gemm8_set_variant(0)
    torch.cuda.synchronize()
m, n, k = (int(v) for v in args.shape.split(","))
gen = torch.Generator(device=dev).manual_seed(m + 3 * n + k)
x = torch.randn(m, k, device=dev, generator=gen).to(torch.bfloat16)
w = (torch.randn(n, k, device=dev, generator=gen) * 0.05).to(torch.bfloat16)
b = torch.randn(n, device=dev, generator=gen).to(torch.bfloat16)
wq8, ws8 = ops.int8_quantize_rowwise(w)
xq8, xs8 = ops.int8_quantize_rowwise(x)
wqf, wsf = ops.fp8_quantize_rowwise(w)
xqf, xsf = ops.fp8_quantize_rowwise(x)
y8_0 = ops.int8_scaled_mm(xq8, xs8, wq8, ws8, b)
yf_0 = ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), b)
lib.ao_gemm8_set_variant(100)
gf = ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), b)
lib.ao_gemm8_set_variant(0)
print("first launch vs tiled kernel: rel", float((yf_0.float() - gf.float()).norm() / gf.float().norm()))
bad8 = badf = 0
for i in range(args.iters):
    y8 = ops.int8_scaled_mm(xq8, xs8, wq8, ws8, b)
    yf = ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), b)
    if i % 7 == 3:  # the test's order: the tiled kernels in between
        lib.ao_gemm8_set_variant(100)
        ops.int8_scaled_mm(xq8, xs8, wq8, ws8, b)
        ops.fp8_scaled_mm(xqf, wqf.t(), xsf, wsf.t(), b)
        lib.ao_gemm8_set_variant(0)
    if not torch.equal(y8, y8_0):
        bad8 += 1
    if i < 3:
        print(f"iter {i}: this launch vs tiled kernel: rel {float((yf.float() - gf.float()).norm() / gf.float().norm()):.4g}")
    if not torch.equal(yf, yf_0):
        badf += 1
        if badf > 3:
            continue
        d = (yf.float() - yf_0.float()).abs()
        nz = d.nonzero()
        print(f"iter {i}: fp8 differs in {nz.shape[0]} elements, rows {nz[:, 0].unique().tolist()[:6]}, columns {int(nz[:, 1].min())}..{int(nz[:, 1].max())}, "
              f"max |diff| {float(d.max()):.4g}, rel {float(d.norm() / yf_0.float().norm()):.4g}")
print(f"{args.iters} iterations: int8 mismatches {bad8}, fp8 mismatches {badf}")
