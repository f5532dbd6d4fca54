#!/usr/bin/env python3
"""gemm8_p8p_kernel (round 6, persistent tiles + register-only epilogue) next to gemm8_p8_kernel (one workgroup per tile) on the Llama-3-8B
int8 shapes at M = 16384 (BASELINE config 3's chunk) and the fp8 shapes at M = 2048 ... 16384: ao_gemm8_set_tuning(6, 1 = never / 2 = wherever
the shape allows), alternating, same process."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops
from tools.midm_sweep import graph_time

lib = _lib.lib()
dev = torch.device("cuda", 0)
torch.manual_seed(0)
cases = [("int8", m, n, k) for m in (16384, 4096) for n, k in ((6144, 4096), (4096, 4096), (14336, 4096), (4096, 14336))] + \
        [("fp8", m, n, k) for m in (16384, 8192, 2048) for n, k in ((7168, 8192), (8192, 3584), (8192, 1024), (6144, 4096), (14336, 4096))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1].split(",")]
for kind, m, n, k in cases:
    quant = ops.int8_quantize_rowwise if kind == "int8" else ops.fp8_quantize_rowwise
    ws = [quant(torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(3)]
    xq, xs = quant(torch.randn(m, k, device=dev, dtype=torch.bfloat16))
    row = {"kind": kind, "M": m, "N": n, "K": k, "tiles": (m // 256) * (n // 256)}
    for rep in range(2):
        for form in (1, 2):
            lib.ao_gemm8_set_variant(32)
            lib.ao_gemm8_set_tuning(6, form)
            try:
                if kind == "int8":
                    calls = [lambda wq=wq, wsc=wsc: ops.int8_scaled_mm(xq, xs, wq, wsc) for wq, wsc in ws]
                else:
                    calls = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(xq, wq.t(), xs, wsc.t()) for wq, wsc in ws]
                t = graph_time(calls)
            finally:
                lib.ao_gemm8_set_tuning(6, 0)
                lib.ao_gemm8_set_variant(0)
            key = ("tile", "persistent")[form - 1] + f"_{rep}"
            row[key + "_us"] = round(t * 1e6, 1)
            row[key + "_TOPs"] = round(2.0 * m * n * k / t / 1e12, 1)
    print(json.dumps(row), flush=True)
