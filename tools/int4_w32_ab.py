#!/usr/bin/env python3
"""A/B of the batched int4 kernels at M = 128 (round 5): per Llama-3-8B shape, cold (rotating weight copies > 256 MiB), the product dispatch
against the 128 x 128 / 32 x 32 x 16 kernel (modes 92S fused, 93S with DMA-producer waves; S = K parts, 0 = auto).

    python tools/int4_w32_ab.py [--ms 128] [--modes 0,920,930,...]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402
from tools.midm_sweep import graph_time  # noqa: E402

SHAPES = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("down", 4096, 14336)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ms", default="128")
    ap.add_argument("--modes", default="0,920,921,922,924,930,931,932,934")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    for name, n, k in SHAPES:
        copies = max(2, -(-(300 << 20) // (n * k // 2)))
        ws = []
        for _ in range(copies):
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            ws.append(ops.int4_quantize_tinygemm(w, 128))
            del w
        for m in [int(v) for v in args.ms.split(",")]:
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            y0 = None
            for mode in [int(v) for v in args.modes.split(",")]:
                lib.ao_int4_set_tuning(0, mode)
                rec = {"shape": name, "N": n, "K": k, "M": m, "mode": mode}
                try:
                    calls = [lambda q=q, sz=sz: ops.weight_int4pack_mm(x, q, 128, sz) for q, sz in ws]
                    y = calls[0]().float()
                    torch.cuda.synchronize()
                    if y0 is None:
                        y0 = y
                    else:
                        rec["rel_vs_product"] = float((y - y0).norm() / y0.norm())
                    t = graph_time(calls)
                    rec["us"] = round(t * 1e6, 2)
                    rec["TFLOPs"] = round(2.0 * m * n * k / t / 1e12, 1)
                except Exception as e:  # noqa: BLE001
                    rec["error"] = repr(e)[:200]
                finally:
                    lib.ao_int4_set_tuning(0, 0)
                print(json.dumps(rec), flush=True)
        del ws


if __name__ == "__main__":
    main()
