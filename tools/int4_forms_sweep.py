#!/usr/bin/env python3
"""int4 tinygemm linears (g = 128): the product dispatch next to forced launch forms (ao_int4_set_tuning) per (shape, M), COLD weights.

    python tools/int4_forms_sweep.py [--ms 8,12,16] [--forms 0:0,4:701,4:702,4:704] [--other] > profiles/int4_forms_rNN.jsonl

A form is wpb:mode (waves per workgroup : A/B mode of include/ao_mi355.h; 0:0 = product).  70S = 16-row slabs of the batched kernel with S
K parts, 71S / 72S / 73S = 32- / 64- / 128-row slabs.  One JSON line per (shape, M, form): us, the kernel name for the product form, and the
norm-relative difference against the product form's output.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402
from tools.bench_dec8 import graph_time  # noqa: E402
from tools.vs_stack_int4 import OTHER, SHAPES  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="8,12,16")
    ap.add_argument("--forms", default="0:0,4:701,4:702,4:704")
    ap.add_argument("--other", action="store_true")
    ap.add_argument("--both", action="store_true")
    ap.add_argument("--shapes", default="", help="name:N:K,... instead of the built-in lists")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    forms = [tuple(int(v) for v in f.split(":")) for f in args.forms.split(",")]
    shapes = (SHAPES + OTHER) if args.both else (OTHER if args.other else SHAPES)
    if args.shapes:
        shapes = [(a, int(b), int(c)) for a, b, c in (t.split(":") for t in args.shapes.split(","))]
    for name, n, k in shapes:
        copies = max(2, -(-(300 << 20) // (n * k // 2)))
        ws = []
        for _ in range(copies):
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            ws.append(ops.int4_quantize_tinygemm(w, 128))
            del w
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            y0 = None
            for wpb, mode in forms:
                rec = {"shape": name, "N": n, "K": k, "M": m, "form": f"{wpb}:{mode}"}
                try:
                    lib.ao_int4_set_tuning(wpb, mode)
                    if mode == 0:
                        rec["kernel"] = lib.ao_int4_mm_kernel_name(m, n, k, 128).decode()
                    y = ops.weight_int4pack_mm(x, ws[0][0], 128, ws[0][1]).float()
                    torch.cuda.synchronize()
                    if y0 is None:
                        y0 = y
                    rec["rel_vs_product"] = float((y - y0).norm() / y0.norm())
                    t = graph_time([lambda q=q, sz=sz: ops.weight_int4pack_mm(x, q, 128, sz) for q, sz in ws])
                    rec["us"] = round(t * 1e6, 2)
                except Exception as e:  # noqa: BLE001
                    rec["error"] = repr(e)[:200]
                finally:
                    lib.ao_int4_set_tuning(0, 0)
                print(json.dumps(rec), flush=True)
        del ws


if __name__ == "__main__":
    main()
