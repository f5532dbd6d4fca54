#!/usr/bin/env python3
"""int4 tinygemm linears (g = 128) at every batch size next to PyTorch core's aten::_weight_int4pack_mm, COLD weights, one MI355X.

    python tools/vs_stack_int4.py [--ms 1,2,4,8,16,32,64,128,256,512,2048] > profiles/vs_stack_int4_rNN.jsonl

The packed layouts are bit-identical, so both ops read the SAME tensors: per shape enough distinct copies to exceed the 256 MiB Infinity
Cache, one hipGraph over the copies, best of 5 replays.  One JSON line per (shape, M): us of both and ours-over-core (> 1: ours is faster).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops  # noqa: E402
from tools.bench_dec8 import graph_time  # noqa: E402

SHAPES = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("down", 4096, 14336)]
# shapes no rule was fitted on (--other): Llama-2-13B, Qwen2-7B, Llama-3-70B
OTHER = [("qkv13b", 15360, 5120), ("o13b", 5120, 5120), ("gate13b", 13824, 5120), ("down13b", 5120, 13824),
         ("qkvq7b", 4608, 3584), ("oq7b", 3584, 3584), ("gateq7b", 18944, 3584), ("downq7b", 3584, 18944),
         ("qkv70b", 10240, 8192), ("o70b", 8192, 8192), ("gate70b", 28672, 8192), ("down70b", 8192, 28672)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="1,2,4,8,16,32,64,128,256,512,2048")
    ap.add_argument("--other", action="store_true", help="the shapes of OTHER instead of the Llama-3-8B linears")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for name, n, k in (OTHER if args.other else SHAPES):
        copies = max(2, -(-(300 << 20) // (n * k // 2)))
        ws = []
        for _ in range(copies):
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            ws.append(ops.int4_quantize_tinygemm(w, 128))
            del w
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            rec = {"shape": name, "N": n, "K": k, "M": m, "copies": copies}
            try:
                t = graph_time([lambda q=q, sz=sz: ops.weight_int4pack_mm(x, q, 128, sz) for q, sz in ws])
                rec["us_ours"] = round(t * 1e6, 2)
                t2 = graph_time([lambda q=q, sz=sz: torch.ops.aten._weight_int4pack_mm(x, q, 128, sz) for q, sz in ws])
                rec["us_core"] = round(t2 * 1e6, 2)
                rec["ours_over_core"] = round(t2 / t, 2)
                y = ops.weight_int4pack_mm(x, ws[0][0], 128, ws[0][1]).float()
                y2 = torch.ops.aten._weight_int4pack_mm(x, ws[0][0], 128, ws[0][1]).float()
                rec["rel_vs_core"] = float((y - y2).norm() / y2.norm())
            except Exception as e:  # noqa: BLE001
                rec["error"] = repr(e)[:200]
            print(json.dumps(rec), flush=True)
        del ws


if __name__ == "__main__":
    main()
