#!/usr/bin/env python3
"""Secondary workloads (BASELINE.json configs 3-5) on one MI355X: per-shape time and fraction of
the roofline that bounds them.  Not the driver's bench (bench.py is): a measurement aid whose output
is committed under profiles/.

    python tools/bench_8bit.py [--m 2048] [--iters 20] [--which int8,fp8,mx,quant]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops  # noqa: E402

PEAK_INT8 = 5.0e15   # dense int8 MFMA, 2x bf16 rate (MI355X_MICROARCH.md: >= 3944 TOPS measured)
PEAK_FP8 = 5.0e15    # dense fp8 (MX-scaled K=128 path), spec ~5 PF
PEAK_HBM = 8.0e12

LLAMA8B = [("qkv_proj", 6144, 4096), ("o_proj", 4096, 4096), ("gate_up_proj", 28672, 4096), ("down_proj", 4096, 14336)]
LLAMA70B_TP8 = [("q(col)", 1024, 8192), ("o(row)", 8192, 1024), ("gate_up(col)", 7168, 8192), ("down(row)", 8192, 3584)]
LLAMA70B = [("qkv_proj", 10240, 8192), ("o_proj", 8192, 8192), ("gate_up_proj", 57344, 8192), ("down_proj", 8192, 28672)]


def timeit(fn, iters, warmup=3):
    """Seconds per call.  The calls are captured into one hipGraph and replayed (like bench.py), so that kernels of a
    few microseconds are not hidden behind ~18 us of eager Python + launch overhead per call."""
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):  # on the capture stream: the library's split-K workspace is per stream, allocated on first use
            fn()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (iters * reps)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--which", default="int8,fp8,mx,quant")
    ap.add_argument("--gemm-variant", type=int, default=0, help="ao_gemm8_set_variant (0 product, 1 reg-staged, 2/4/8 tile shapes)")
    args = ap.parse_args()
    which = set(args.which.split(","))
    dev = torch.device("cuda", 0)
    from ao_amd import _lib
    _lib.lib().ao_gemm8_set_variant(args.gemm_variant)
    torch.manual_seed(0)
    out = []

    def rec(**kw):
        out.append(kw)
        print(json.dumps(kw))

    M = args.m
    if "quant" in which:
        for k in (4096, 14336):
            # inputs rotate over enough distinct buffers to exceed the 256 MiB Infinity Cache (a re-used input is read from it)
            nbuf = max(2, -(-(300 << 20) // (M * k * 2)))
            xs = [torch.randn(M, k, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
            turn = [0]

            def next_x():
                turn[0] = (turn[0] + 1) % nbuf
                return xs[turn[0]]

            for name, fn, wb in (("int8_quantize_rowwise", ops.int8_quantize_rowwise, 1), ("fp8_quantize_rowwise", ops.fp8_quantize_rowwise, 1),
                                 ("mxfp8_quantize", ops.mxfp8_quantize, 1 + 1 / 32)):
                t = timeit(lambda: fn(next_x()), args.iters)
                b = M * k * (2 + wb)
                rec(kernel=name, M=M, K=k, us=t * 1e6, GBps=b / t / 1e9, frac_hbm=b / t / PEAK_HBM)
    if "int8" in which:
        for name, n, k in LLAMA8B:
            x = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            wq, ws = ops.int8_quantize_rowwise(w)
            xq, xs = ops.int8_quantize_rowwise(x)
            t = timeit(lambda: ops.int8_scaled_mm(xq, xs, wq, ws), args.iters)
            t2 = timeit(lambda: ops.int8_scaled_mm(*ops.int8_quantize_rowwise(x), wq, ws), args.iters)
            f = 2.0 * M * n * k
            extra = {}
            if ops.dynamic_linear_fits(M, n, k):  # decode sizes: cast fused into the matmul (one launch)
                extra["us_fused_act_quant"] = timeit(lambda: ops.int8_dynamic_linear(x, wq, ws), args.iters) * 1e6
            rec(kernel="int8_scaled_mm", shape=name, M=M, N=n, K=k, us=t * 1e6, TOPs=f / t / 1e12, frac_mfma=f / t / PEAK_INT8,
                us_with_act_quant=t2 * 1e6, **extra)
    if "fp8" in which:
        for m in sorted({1, 16, 64, 128, M}):
            for name, n, k in LLAMA70B_TP8:
                x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
                w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
                wq, ws = ops.fp8_quantize_rowwise(w)
                xq, xs = ops.fp8_quantize_rowwise(x)
                t = timeit(lambda: ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t()), args.iters)
                f = 2.0 * m * n * k
                b = n * k + m * k + m * n * 2
                rec(kernel="fp8_scaled_mm", shape=name, M=m, N=n, K=k, us=t * 1e6, TFLOPs=f / t / 1e12, frac_mfma=f / t / PEAK_FP8,
                    GBps=b / t / 1e9, frac_hbm=b / t / PEAK_HBM)
    if "fp8l" in which:  # fp8 rowwise on the Llama-3-8B shapes at one M (GEMM tile-shape A/B)
        for name, n, k in LLAMA8B:
            x = torch.randn(M, k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            wq, ws = ops.fp8_quantize_rowwise(w)
            xq, xs = ops.fp8_quantize_rowwise(x)
            t = timeit(lambda: ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t()), args.iters)
            f = 2.0 * M * n * k
            rec(kernel="fp8_scaled_mm", shape=name, M=M, N=n, K=k, us=t * 1e6, TFLOPs=f / t / 1e12, frac_mfma=f / t / PEAK_FP8)
    if "mx" in which:
        E, rows = 8, 128
        sizes = [32, 0, 32, 16, 16, 0, 32, 0]
        if args.m != 2048:  # --m R: R rows spread evenly over the 8 experts instead of the default ragged decode batch
            rows = args.m
            sizes = [rows // E] * E
        offs = torch.tensor([sum(sizes[: i + 1]) for i in range(E)], dtype=torch.int32, device=dev)
        for name, n, k in (("w1/w3", 14336, 4096), ("w2", 4096, 14336)):
            a = torch.randn(rows, k, device=dev, dtype=torch.bfloat16)
            w = torch.randn(E, n, k, device=dev, dtype=torch.bfloat16) * 0.02
            wq, ws = ops.mxfp8_quantize(w)
            del w
            aq, a_s = ops.mxfp8_quantize(a)
            t = timeit(lambda: ops.mxfp8_grouped_mm(aq, a_s, wq, ws, offs), args.iters)
            used = sum(1 for s in sizes if s)
            b = used * n * k * (1 + 1 / 32) + rows * k + rows * n * 2
            rec(kernel="mxfp8_grouped_mm", shape=name, E=E, rows=rows, N=n, K=k, us=t * 1e6, GBps=b / t / 1e9, frac_hbm=b / t / PEAK_HBM,
                note="bytes count only experts with tokens")


if __name__ == "__main__":
    main()
