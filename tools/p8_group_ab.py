#!/usr/bin/env python3
"""gemm8_p8_kernel: the XCD tile-group height (ao_gemm8_set_tuning(4, v)) on the Llama-3-8B int8 shapes at M = 16384 and the fp8 70B / TP8 shards at
M = 2048 / 8192 (round 5; VERDICT r4 item 3: the grouping was never swept)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops
from tools.midm_sweep import graph_time

lib = _lib.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cases = [("int8", m, n, k) for m in (16384,) for n, k in ((6144, 4096), (4096, 4096), (14336, 4096), (4096, 14336))] + \
        [("fp8", m, n, k) for m in (2048, 8192) for n, k in ((7168, 8192), (8192, 3584), (8192, 1024))]
for kind, m, n, k in cases:
    quant = ops.int8_quantize_rowwise if kind == "int8" else ops.fp8_quantize_rowwise
    ws = [quant(torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(3)]
    xq, xs = quant(torch.randn(m, k, device=dev, dtype=torch.bfloat16))
    for gr in (8, 1, 2, 4, 16, 32):
        lib.ao_gemm8_set_tuning(4, gr)
        try:
            if kind == "int8":
                calls = [lambda wq=wq, wsc=wsc: ops.int8_scaled_mm(xq, xs, wq, wsc) for wq, wsc in ws]
            else:
                calls = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(xq, wq.t(), xs, wsc.t()) for wq, wsc in ws]
            t = graph_time(calls)
            print(json.dumps({"kind": kind, "M": m, "N": n, "K": k, "group_rows": gr, "us": round(t * 1e6, 1), "TOPs": round(2.0 * m * n * k / t / 1e12, 1),
                              "kernel": lib.ao_gemm8_kernel_name(1 if kind == "int8" else 0, m, n, k).decode()}), flush=True)
        finally:
            lib.ao_gemm8_set_tuning(4, 0)
