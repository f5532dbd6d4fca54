#!/usr/bin/env python3
"""fp8 / int8 rowwise GEMMs at 128 <= M <= 1024 on the narrow shapes (Llama-3-70B TP=8 shards, Llama-3-8B linears), COLD: every call of
a replay reads another copy of the weight (> 256 MiB of copies per shape: nothing comes from the Infinity Cache).  Per (shape, M): PyTorch
core's kernel (hipBLASLt: what torchao-on-ROCm runs today), this library's dispatch, and the tuning forms of ao_gemm8_set_tuning /
ao_gemm8_set_variant named on the command line.  One JSON line per measurement; output goes to profiles/.

    python tools/midm_sweep.py [--ms 128,256,512,1024] [--kinds fp8] [--families 70b,8b] [--forms default,bm64,bn32+s2,...]

A form is a comma list of key=value tunings joined by '+', or a name from FORMS below.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402

SHAPES = {
    "70b": [("qkv70b/8", 1280, 8192), ("o70b/8", 8192, 1024), ("gate_up70b/8", 7168, 8192), ("down70b/8", 8192, 3584)],
    "8b": [("qkv8b", 6144, 4096), ("o8b", 4096, 4096), ("gate_up8b", 28672, 4096), ("down8b", 4096, 14336)],
    # shapes no rule was fitted on (round 6, generalisation check): Llama-2-13B, Qwen2-7B, Llama-3-70B TP = 4 shards
    "13b": [("qkv13b", 15360, 5120), ("o13b", 5120, 5120), ("gate_up13b", 27648, 5120), ("down13b", 5120, 13824)],
    "qwen7b": [("qkvq7b", 4608, 3584), ("oq7b", 3584, 3584), ("gate_upq7b", 37888, 3584), ("downq7b", 3584, 18944)],
    "70b4": [("qkv70b/4", 2560, 8192), ("o70b/4", 8192, 2048), ("gate_up70b/4", 14336, 8192), ("down70b/4", 8192, 7168)],
}
# name -> (variant, {tuning key: value})
FORMS = {
    "default": (0, {}),
    "bm64": (0, {3: 64}),       # 64-row slabs at any M (M is cut instead of K)
    "bn32": (0, {1: 32}),
    "bn64": (0, {1: 64}),
    "bn128": (0, {1: 128}),
    "rb": (101, {}),            # the weight-streaming kernel forced
    "tile": (100, {}),          # never a weight-streaming kernel
    "p8": (32, {}),
}


def parse_form(name):
    if name in FORMS:
        return FORMS[name]
    variant, tun = 0, {}
    for part in name.split("+"):
        if part in FORMS:
            v, t = FORMS[part]
            variant = v or variant
            tun.update(t)
        elif part.startswith("bm"):  # slab height (64: two slabs share a weight tile through L2 at M = 128)
            tun[3] = int(part[2:])
        elif part.startswith("v"):
            variant = int(part[1:])
        elif part.startswith("s"):  # K parts
            tun[2] = int(part[1:])
        else:
            k, v = part.split("=")
            tun[int(k)] = int(v)
    return variant, tun


def graph_time(calls, reps=4):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="128,256,512,1024")
    ap.add_argument("--kinds", default="fp8")
    ap.add_argument("--families", default="70b,8b")
    ap.add_argument("--forms", default="default,bm64")
    ap.add_argument("--no-core", action="store_true")
    ap.add_argument("--check", action="store_true", help="compare every form's output with the default dispatch's (max abs diff)")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    forms = [(f, parse_form(f)) for f in args.forms.split(",")]
    for kind in args.kinds.split(","):
        quant = ops.fp8_quantize_rowwise if kind == "fp8" else ops.int8_quantize_rowwise
        for fam in args.families.split(","):
            for name, n, k in SHAPES[fam]:
                copies = max(2, -(-(300 << 20) // (n * k)))
                ws = []
                for _ in range(copies):
                    w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
                    ws.append(quant(w))
                    del w
                for m in [int(v) for v in args.ms.split(",")]:
                    x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
                    xq, xs = quant(x)
                    base = {"kind": kind, "shape": name, "N": n, "K": k, "M": m, "copies": copies}
                    t_core = None
                    if not args.no_core:
                        rec = dict(base, form="core")
                        try:
                            if kind == "fp8":
                                xs2 = xs.reshape(m, 1).contiguous()
                                core = [lambda wq=wq, wsc=wsc: torch._scaled_mm(xq, wq.t(), scale_a=xs2, scale_b=wsc.reshape(1, n), out_dtype=torch.bfloat16, use_fast_accum=True)
                                        for wq, wsc in ws]
                            else:  # the reference's path: _int_mm, then the two scale multiplies (int8/kernels.py:114-144)
                                xsb = xs.reshape(m, 1).to(torch.bfloat16)
                                core = [lambda wq=wq, wsc=wsc: (torch._int_mm(xq, wq.t()).to(torch.bfloat16) * xsb) * wsc.reshape(1, n).to(torch.bfloat16) for wq, wsc in ws]
                            t_core = graph_time(core)
                            rec["us"] = round(t_core * 1e6, 2)
                        except Exception as e:  # noqa: BLE001
                            rec["error"] = repr(e)[:200]
                        print(json.dumps(rec), flush=True)
                    y_ref = None
                    for fname, (variant, tun) in forms:
                        lib.ao_gemm8_set_variant(variant)
                        for key in (1, 2, 3, 4, 5, 6, 7, 8):
                            lib.ao_gemm8_set_tuning(key, tun.get(key, 0))
                        rec = dict(base, form=fname, kernel=lib.ao_gemm8_kernel_name(0 if kind == "fp8" else 1, m, n, k).decode() if fname == "default" else None)
                        try:
                            if kind == "fp8":
                                calls = [lambda wq=wq, wsc=wsc: ops.fp8_scaled_mm(xq, wq.t(), xs, wsc.t()) for wq, wsc in ws]
                            else:
                                calls = [lambda wq=wq, wsc=wsc: ops.int8_scaled_mm(xq, xs, wq, wsc) for wq, wsc in ws]
                            if args.check:
                                y = calls[0]().float()
                                torch.cuda.synchronize()
                                if y_ref is None:
                                    y_ref = y
                                else:
                                    rec["max_abs_diff_vs_first_form"] = float((y - y_ref).abs().max())
                            t = graph_time(calls)
                            rec["us"] = round(t * 1e6, 2)
                            rec["TFLOPs"] = round(2.0 * m * n * k / t / 1e12, 1)
                            if t_core:
                                rec["ours_over_core"] = round(t_core / t, 3)
                        except Exception as e:  # noqa: BLE001
                            rec["error"] = repr(e)[:200]
                        finally:
                            lib.ao_gemm8_set_variant(0)
                            for key in (1, 2, 3, 4, 5, 6, 7, 8):
                                lib.ao_gemm8_set_tuning(key, 0)
                        print(json.dumps(rec), flush=True)
                    if m == int(args.ms.split(",")[0]):
                        pass  # (the same-XCD meeting was removed in round 6)
                del ws


if __name__ == "__main__":
    main()
