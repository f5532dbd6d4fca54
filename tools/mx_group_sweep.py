#!/usr/bin/env python3
"""MXFP8 / fp8-rowwise grouped GEMMs over the tokens-per-expert axis (Mixtral-8x7B expert shapes, E = 8, every expert the same count),
COLD weights (two copies of the 8 experts per shape: 0.94 GB per w1 call pair), one hipGraph per point: microseconds per launch and the
weight stream's TB/s -- where the kernel choice changes (decode-size groups <= 48 rows per expert: stream-K; 64- / 128-row slabs beyond),
the curve shows whether the seams are level.  One JSON line per (op, shape, tokens per expert).

    python tools/mx_group_sweep.py [--tokens 1,2,4,8,16,32,48,49,64,96,128,256] > profiles/mx_group_sweep_rNN.jsonl
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops  # noqa: E402
from tools.bench_dec8 import graph_time  # noqa: E402

SHAPES = [("w1", 14336, 4096), ("w2", 4096, 14336)]
E = 8


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="1,2,4,8,16,32,48,49,64,96,128,256")
    ap.add_argument("--slab-rows", default="0", help="comma list of forced slab heights (ao_gemm8_set_tuning key 3: 0 product rule, 64, 128)")
    ap.add_argument("--shapes", default="", help="name:N:K,... instead of the Mixtral-8x7B pair (round 6: other MoE models)")
    ap.add_argument("--experts", type=int, default=8)
    ap.add_argument("--variants", default="0", help="comma list of ao_gemm8_set_variant values (0 product, 113 one workgroup per tile, 129 per-step scales)")
    args = ap.parse_args()
    from ao_amd import _lib
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    global E
    E = args.experts
    shapes = SHAPES if not args.shapes else [(a, int(b), int(c)) for a, b, c in (t.split(":") for t in args.shapes.split(","))]
    for name, n, k in shapes:
        mx, f8 = [], []
        for _ in range(2):
            w = torch.randn(E, n, k, device=dev, dtype=torch.bfloat16) * 0.05
            mx.append(ops.mxfp8_quantize(w, "rceil"))
            q, s = ops.fp8_quantize_rowwise(w.reshape(E * n, k))
            f8.append((q.reshape(E, n, k), s.reshape(E, n)))
            del w
        for t in [int(v) for v in args.tokens.split(",")]:
            m = t * E
            offs = torch.arange(1, E + 1, device=dev, dtype=torch.int32) * t
            a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            aq, asc = ops.mxfp8_quantize(a, "rceil")
            fq, fs = ops.fp8_quantize_rowwise(a)
            forms = {
                "mxfp8_grouped_mm": [lambda w=w: ops.mxfp8_grouped_mm(aq, asc, w[0], w[1], offs) for w in mx],
                "mxfp8_cast_then_mm": [lambda w=w: ops.mxfp8_grouped_mm(*ops.mxfp8_quantize(a, "rceil"), w[0], w[1], offs) for w in mx],
                "fp8_grouped_mm": [lambda w=w: ops.fp8_grouped_mm(fq, fs, w[0], w[1], offs) for w in f8],
            }
            if ops.mxfp8_grouped_mm_dyn_fits(m, n, k, E):
                forms["mxfp8_grouped_mm_dyn"] = [lambda w=w: ops.mxfp8_grouped_mm_dyn(a, w[0], w[1], offs, "rceil") for w in mx]
            for op, calls, bm, var in ((o, c, b, v) for v in [int(v) for v in args.variants.split(",")] for b in [int(v) for v in args.slab_rows.split(",")] for o, c in forms.items()):
                rec = {"op": op, "shape": name, "N": n, "K": k, "E": E, "tokens_per_expert": t, "M": m, "slab_rows": bm, "variant": var}
                try:
                    lib.ao_gemm8_set_variant(var)
                    lib.ao_gemm8_set_tuning(3, bm)
                    sec = graph_time(calls)
                    rec["us"] = round(sec * 1e6, 2)
                    rec["weight_TBps"] = round(E * n * k * (1 + 1 / 32 if op.startswith("mx") else 1) / sec / 1e12, 3)
                except Exception as e:  # noqa: BLE001
                    rec["error"] = repr(e)[:200]
                finally:
                    lib.ao_gemm8_set_tuning(3, 0)
                    lib.ao_gemm8_set_variant(0)
                print(json.dumps(rec), flush=True)
        del mx, f8


if __name__ == "__main__":
    main()
