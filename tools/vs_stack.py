#!/usr/bin/env python3
"""Per shape: this library's 8-bit GEMMs next to PyTorch core's (hipBLASLt) on the same box and inputs -- what torchao-on-ROCm runs today.

    python tools/vs_stack.py [--ms 2048,16384] > profiles/vs_stack_rNN.jsonl

One JSON line per (kind, M, shape): us and T(FL)OP/s of both, ratio ours / core (> 1: ours is faster).  hipGraph replay of 20 calls, median of 3.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops  # noqa: E402
from tools.bench_8bit import LLAMA8B, LLAMA70B_TP8, timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="128,2048,16384")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for m in [int(v) for v in args.ms.split(",")]:
        for fam, shapes in (("llama3-8b", LLAMA8B), ("llama3-70b/tp8", LLAMA70B_TP8)):
            for name, n, k in shapes:
                x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
                w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
                f = 2.0 * m * n * k
                for kind in ("fp8", "int8"):
                    rec = {"kind": kind, "family": fam, "shape": name, "M": m, "N": n, "K": k}
                    try:
                        if kind == "fp8":
                            wq, ws = ops.fp8_quantize_rowwise(w)
                            xq, xs = ops.fp8_quantize_rowwise(x)
                            wt, wst = wq.t(), ws.t()
                            t_ours = timeit(lambda: ops.fp8_scaled_mm(xq, wt, xs, wst), args.iters)
                            xs2, ws2 = xs.reshape(m, 1).contiguous(), ws.reshape(1, n).contiguous()
                            t_core = timeit(lambda: torch._scaled_mm(xq, wt, scale_a=xs2, scale_b=ws2, out_dtype=torch.bfloat16, use_fast_accum=True), args.iters)
                        else:
                            wq, ws = ops.int8_quantize_rowwise(w)
                            xq, xs = ops.int8_quantize_rowwise(x)
                            t_ours = timeit(lambda: ops.int8_scaled_mm(xq, xs, wq, ws), args.iters)
                            wt = wq.t()
                            # the reference's path: _int_mm, then the two scale multiplies (int8/kernels.py:114-144)
                            t_core = timeit(lambda: (torch._int_mm(xq, wt).to(torch.bfloat16) * xs.reshape(m, 1).to(torch.bfloat16)) * ws.reshape(1, n).to(torch.bfloat16), args.iters) if m > 16 else None
                        rec.update(us_ours=t_ours * 1e6, T_ours=f / t_ours / 1e12)
                        if t_core:
                            rec.update(us_core=t_core * 1e6, T_core=f / t_core / 1e12, ours_over_core=t_core / t_ours)
                    except Exception as e:  # noqa: BLE001
                        rec["error"] = repr(e)[:200]
                    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
