#!/usr/bin/env python3
"""Decode-size timing of the fp8-activation x int4-weight kernel on the Llama-3-8B linears, next to the bf16 x int4
(tinygemm) kernel on the SAME packed weights: one token through 160 linears, hipGraph replay, cold weights (bench.py's model).
    python tools/bench_fp8_int4.py [--batch 1]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from ao_amd import ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    model = bench.Int4Linears(dev, bench.N_LAYERS, bench.LLAMA3_8B_UNMERGED)
    stream = torch.cuda.Stream(device=dev)
    B = args.batch
    xs = {}
    for qdata, sz, n, k, _ in model.weights:
        if k not in xs:
            x = torch.randn(B, k, device=dev, dtype=torch.bfloat16)
            xs[k] = ops.fp8_quantize_rowwise(x)
    def step_fp8():
        for qdata, sz, n, k, _ in model.weights:
            xq, xsc = xs[k]
            ops.fp8_int4_linear(xq, xsc, qdata, sz, bench.GROUP)
    def step_fp8_with_cast():
        for qdata, sz, n, k, _ in model.weights:
            xq, xsc = ops.fp8_quantize_rowwise(xs_bf16[k])
            ops.fp8_int4_linear(xq, xsc, qdata, sz, bench.GROUP)
    xs_bf16 = {k: torch.randn(B, k, device=dev, dtype=torch.bfloat16) for k in xs}
    def step_fp8_fused():  # round 4: cast inside the matmul launch at decode sizes (ops.fp8_int4_act_linear)
        for qdata, sz, n, k, _ in model.weights:
            ops.fp8_int4_act_linear(xs_bf16[k], qdata, sz, bench.GROUP)
    out = {"batch": B, "layout": "five", "bytes_per_step": model.bytes_per_step(B)}
    with torch.cuda.stream(stream):
        t_bf16, _ = bench.run_int4(model, B, args.steps, 3, stream, dev, True)
        out["bf16_x_int4_tokens_per_s"] = B * args.steps / t_bf16
        for name, fn in (("fp8_x_int4_gemm_only", step_fp8), ("fp8_x_int4_with_activation_cast", step_fp8_with_cast),
                         ("fp8_x_int4_act_linear_one_op", step_fp8_fused)):
            run, graphed = bench.capture(fn, stream)
            t = bench.time_steps(run, stream, dev, args.steps, 3) / args.steps
            out[name + "_tokens_per_s"] = B / t
            out[name + "_frac_of_hbm_roofline"] = model.bytes_per_step(B) / t / 1e9 / bench.HBM_PEAK_GBS
    print(json.dumps(out))


if __name__ == "__main__":
    main()
