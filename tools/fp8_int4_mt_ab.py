#!/usr/bin/env python3
"""fp8-activation x int4-weight linear at 17 <= M <= 512: m-tiles per workgroup forced to 1 (rounds 3-4: one workgroup per 16
rows, the weights re-read per slab), 2 (with one and with two n-tiles per workgroup), 4, and the product rule (0), cold weights, Llama-3-8B shapes, g = 128.  One JSON line per (shape, M)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import _lib, ops  # noqa: E402
from ao_amd.quantization.int4_plain_tensor import Int4Tensor  # noqa: E402
from midm_sweep import graph_time  # noqa: E402

SHAPES = [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("down", 4096, 14336)]


def main():
    lib = _lib.lib()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ms = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64,128,256,512").split(",")]
    for name, n, k in SHAPES:
        copies = max(2, -(-(300 << 20) // (n * k // 2)))
        ws = []
        for _ in range(copies):
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02
            ws.append(Int4Tensor.from_hp(w, [1, 128], activation_dtype=torch.float8_e4m3fn).tile_packed())
            del w
        for m in ms:
            xq, xs = ops.fp8_quantize_rowwise(torch.randn(m, k, device=dev, dtype=torch.bfloat16))
            rec = {"shape": name, "N": n, "K": k, "M": m}
            ref = None
            for mode in (961, 972, 962, 964, 0):
                lib.ao_int4_set_tuning(0, mode)
                try:
                    calls = [lambda q=q, sz=sz: ops.fp8_int4_linear(xq, xs, q, sz, 128) for q, sz in ws]
                    y = calls[0]().float()
                    torch.cuda.synchronize()
                    if ref is None:
                        ref = y
                    key = {961: "mt1", 972: "mt2_nt1", 962: "mt2_nt2", 964: "mt4_nt1", 0: "auto"}[mode]
                    rec[key + "_us"] = round(graph_time(calls) * 1e6, 1)
                    rec[key + "_equal"] = bool(torch.equal(y, ref))
                finally:
                    lib.ao_int4_set_tuning(0, 0)
            print(json.dumps(rec), flush=True)
        del ws


if __name__ == "__main__":
    main()
