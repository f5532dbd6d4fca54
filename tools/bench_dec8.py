#!/usr/bin/env python3
"""8-bit decode linears (M <= 16) per shape, COLD: every call of a replay reads a different copy of the weight, enough copies to
exceed the 256 MiB Infinity Cache.  A/B over the kernel forms of ao_gemm8_set_variant (0 product, 299 the round-3 kernels,
290 half-line loads, 201..208 forced ring depth).  Measurement aid; output goes to profiles/.

    python tools/bench_dec8.py [--ms 1,4,16] [--variants 0,299,290,204,208] [--kinds fp8,int8]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402

SHAPES = {
    "fp8": [("qkv70b/8", 1280, 8192), ("o70b/8", 8192, 1024), ("gate_up70b/8", 7168, 8192), ("down70b/8", 8192, 3584)],
    "int8": [("qkv8b", 6144, 4096), ("o8b", 4096, 4096), ("gate8b", 14336, 4096), ("down8b", 4096, 14336)],
}


def graph_time(calls, reps=5):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls[0]()
        side.synchronize()
        with torch.cuda.graph(g, stream=side):
            for c in calls:
                c()
    torch.cuda.current_stream().wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3 / len(calls))
    return best


def _cast(quant, x, wt, wsct):
    xq, xs = quant(x)
    return xq, wt, xs, wsct


def _int8_cast_mm(x, wq, wsc):
    xq, xs = ops.int8_quantize_rowwise(x)
    return ops.int8_scaled_mm(xq, xs, wq, wsc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,4,16")
    ap.add_argument("--variants", default="0,299,290,201,202,204,207,208")
    ap.add_argument("--kinds", default="fp8,int8")
    ap.add_argument("--linear", action="store_true", help="also time the whole linear on a bf16 activation: cast + matmul (two launches) next to the fused kernel")
    ap.add_argument("--all-shapes", action="store_true", help="both shape families for both kinds")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for kind in args.kinds.split(","):
        quant = ops.fp8_quantize_rowwise if kind == "fp8" else ops.int8_quantize_rowwise
        for name, n, k in (SHAPES["fp8"] + SHAPES["int8"] if args.all_shapes else SHAPES[kind]):
            copies = max(2, -(-(300 << 20) // (n * k)))
            ws = []
            for _ in range(copies):
                w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
                ws.append(quant(w))
                del w
            for m in [int(v) for v in args.ms.split(",")]:
                x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
                xq, xs = quant(x)
                for v in [int(v) for v in args.variants.split(",")]:
                    if v < 0:  # -1: PyTorch core's own kernel (hipBLASLt) on the same cold copies -- what torchao-on-ROCm runs today
                        rec = {"kind": kind, "shape": name, "N": n, "K": k, "M": m, "variant": "core", "copies": copies}
                        try:
                            if kind == "fp8":
                                xs2 = xs.reshape(m, 1).contiguous()
                                core = [lambda wq=wq, wsc=wsc: torch._scaled_mm(xq, wq.t(), scale_a=xs2, scale_b=wsc.reshape(1, n), out_dtype=torch.bfloat16, use_fast_accum=True)
                                        for wq, wsc in ws]
                            else:
                                core = [lambda wq=wq: torch._int_mm(xq, wq.t()) for wq, wsc in ws]
                            t = graph_time(core)
                            rec["mm_us"] = round(t * 1e6, 2)
                            rec["mm_TBps"] = round(n * k / t / 1e12, 3)
                        except Exception as e:  # noqa: BLE001
                            rec["error"] = repr(e)[:200]
                        print(json.dumps(rec), flush=True)
                        continue
                    lib.ao_gemm8_set_variant(v)
                    try:
                        if kind == "fp8":
                            two = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(xq, wq.t(), xs, wsc.t()) for wq, wsc in ws]
                            one = [lambda wq=wq, wsc=wsc: ops.fp8_dynamic_linear(x, wq, wsc) for wq, wsc in ws]
                        else:
                            two = [lambda wq=wq, wsc=wsc: ops.int8_scaled_mm(xq, xs, wq, wsc) for wq, wsc in ws]
                            one = [lambda wq=wq, wsc=wsc: ops.int8_dynamic_linear(x, wq, wsc) for wq, wsc in ws]
                        rec = {"kind": kind, "shape": name, "N": n, "K": k, "M": m, "variant": v, "copies": copies}
                        t = graph_time(two)
                        rec["mm_us"] = round(t * 1e6, 2)
                        rec["mm_TBps"] = round(n * k / t / 1e12, 3)
                        if args.linear:
                            if kind == "fp8":
                                lin = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(*_cast(ops.fp8_quantize_rowwise, x, wq.t(), wsc.t())) for wq, wsc in ws]
                            else:
                                lin = [lambda wq=wq, wsc=wsc: _int8_cast_mm(x, wq, wsc) for wq, wsc in ws]
                            rec["cast_mm_us"] = round(graph_time(lin) * 1e6, 2)
                            rec["preferred"] = bool(ops.dynamic_linear_preferred(m, n, k))
                        if ops.dynamic_linear_fits(m, n, k):
                            t = graph_time(one)
                            rec["fused_us"] = round(t * 1e6, 2)
                            rec["fused_TBps"] = round(n * k / t / 1e12, 3)
                    except Exception as e:  # noqa: BLE001
                        rec["error"] = repr(e)[:200]
                    finally:
                        lib.ao_gemm8_set_variant(0)
                    print(json.dumps(rec), flush=True)
            del ws


if __name__ == "__main__":
    main()
