#!/usr/bin/env python3
"""gemm8_p8_kernel with K parts (ao_gemm8_set_tuning key 7) next to the product dispatch and PyTorch core (hipBLASLt), cold weights, on the
narrow shapes at 256 <= M <= 2048.  One JSON line per (kind, shape, M): {"core", "default", "p8_s<S>" ...} in microseconds."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402
from midm_sweep import SHAPES, graph_time  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="256,512,768,1024,2048")
    ap.add_argument("--kinds", default="fp8")
    ap.add_argument("--splits", default="1,2,3,4,6,8,12,16")
    ap.add_argument("--no-half", action="store_true")
    ap.add_argument("--variant", type=int, default=32, help="32: K parts of the 256 x 256 kernel; 33: of the 256 x 128 one")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for kind in args.kinds.split(","):
        quant = ops.fp8_quantize_rowwise if kind == "fp8" else ops.int8_quantize_rowwise
        for fam in ("70b", "8b"):
            for name, n, k in SHAPES[fam]:
                copies = max(2, -(-(300 << 20) // (n * k)))
                ws = [quant(torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02) for _ in range(copies)]
                for m in [int(v) for v in args.ms.split(",")]:
                    xq, xs = quant(torch.randn(m, k, device=dev, dtype=torch.bfloat16))
                    rec = {"kind": kind, "shape": name, "N": n, "K": k, "M": m}
                    if kind == "fp8":
                        xs2 = xs.reshape(m, 1).contiguous()
                        core = [lambda wq=wq, wsc=wsc: torch._scaled_mm(xq, wq.t(), scale_a=xs2, scale_b=wsc.reshape(1, n), out_dtype=torch.bfloat16, use_fast_accum=True) for wq, wsc in ws]
                        calls = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(xq, wq.t(), xs, wsc.t()) for wq, wsc in ws]
                    else:
                        xsb = xs.reshape(m, 1).to(torch.bfloat16)
                        core = [lambda wq=wq, wsc=wsc: (torch._int_mm(xq, wq.t()).to(torch.bfloat16) * xsb) * wsc.reshape(1, n).to(torch.bfloat16) for wq, wsc in ws]
                        calls = [lambda wq=wq, wsc=wsc: ops.int8_scaled_mm(xq, xs, wq, wsc) for wq, wsc in ws]
                    try:
                        rec["core"] = round(graph_time(core) * 1e6, 1)
                    except Exception as e:  # noqa: BLE001
                        rec["core_error"] = repr(e)[:100]
                    rec["default"] = round(graph_time(calls) * 1e6, 1)
                    rec["default_kernel"] = lib.ao_gemm8_kernel_name(0 if kind == "fp8" else 1, m, n, k).decode()
                    tiles = -(-m // 256) * -(-n // (256 if args.variant == 32 else 128))
                    done = set()
                    try:
                        lib.ao_gemm8_set_variant(args.variant)
                        for s in [int(v) for v in args.splits.split(",")]:
                            eff = max(1, min(s, 256 // tiles, k // 128))
                            if eff in done:
                                continue
                            done.add(eff)
                            lib.ao_gemm8_set_tuning(7, eff)
                            rec[f"{'p8' if args.variant == 32 else 'p8h'}_s{eff}"] = round(graph_time(calls) * 1e6, 1)
                        lib.ao_gemm8_set_tuning(7, 0)
                        if not args.no_half and args.variant == 32:
                            lib.ao_gemm8_set_variant(33)  # 256 x 128 tiles
                            rec["p8h"] = round(graph_time(calls) * 1e6, 1)
                    finally:
                        lib.ao_gemm8_set_variant(0)
                        lib.ao_gemm8_set_tuning(7, 0)
                    print(json.dumps(rec), flush=True)
                del ws


if __name__ == "__main__":
    main()
