#!/usr/bin/env python3
"""How far apart are ao_mxfp8_grouped_mm_pair's outputs and the two single-product launches when the stream-K shares cut the tiles at
different k steps (tools/fuzz_long.py found E = 8, N >= 2112, K >= 2560 cases that are not bit-equal)?  Prints, per case: elements that
differ, the largest difference in bf16 ulps of the single-product value, and both forms' distance from an fp32 dequantised reference."""
import json, os, sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops

DEV = "cuda"


def dq(q, s):  # e4m3 codes [.., K] x E8M0 [.., K/32] -> fp32
    return q.view(torch.float8_e4m3fn).float() * torch.exp2(s.view(torch.uint8).float() - 127.0).repeat_interleave(32, dim=-1)


cases = [([33, 33, 31, 33, 5, 33, 0, 31], 4096, 4096, "floor"), ([16, 0, 16, 1, 1, 1, 16, 1], 4096, 4096, "rceil"), ([33, 48, 48, 31, 33, 5, 5, 31], 4096, 3584, "rceil"),
         ([32, 0, 0, 0, 32, 64, 0, 0], 14336, 4096, "rceil"), ([16] * 8, 14336, 4096, "rceil"), ([33, 1, 48, 1, 1, 5, 48, 33], 4096, 3072, "rceil")]
for sizes, n, k, mode in cases:
    e, m = len(sizes), sum(sizes)
    g = torch.Generator(device=DEV).manual_seed(m + n + k)
    a = torch.randn(m, k, device=DEV, generator=g).to(torch.bfloat16)
    w1 = (torch.randn(e, n, k, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    w3 = (torch.randn(e, n, k, device=DEV, generator=g) * 0.1).to(torch.bfloat16)
    offs = torch.tensor(np.cumsum(sizes), dtype=torch.int32, device=DEV)
    aq, a_s = ops.mxfp8_quantize(a, mode)
    w1q, w1s = ops.mxfp8_quantize(w1, "rceil")
    w3q, w3s = ops.mxfp8_quantize(w3, "rceil")
    y1 = ops.mxfp8_grouped_mm(aq, a_s, w1q, w1s, offs)[:m]
    y3 = ops.mxfp8_grouped_mm(aq, a_s, w3q, w3s, offs)[:m]
    p1, p3 = ops.mxfp8_grouped_mm_pair(a, w1q, w1s, w3q, w3s, offs, mode)
    p1, p3 = p1[:m], p3[:m]
    again, _ = ops.mxfp8_grouped_mm_pair(a, w1q, w1s, w3q, w3s, offs, mode)
    ad = dq(aq, a_s)
    ref = torch.empty(m, n, device=DEV)
    lo = 0
    for i, sz in enumerate(sizes):
        if sz:
            ref[lo:lo + sz] = ad[lo:lo + sz] @ dq(w1q[i], w1s[i]).t()
        lo += sz
    neq = (y1 != p1)
    ulp = torch.exp2(torch.floor(torch.log2(y1.float().abs().clamp_min(1e-30))) - 7)
    d = ((y1.float() - p1.float()).abs() / ulp)
    row = {"sizes": sizes, "N": n, "K": k, "mode": mode, "elements": m * n, "differ": int(neq.sum()), "max_diff_ulps": float(d.max()),
           "single_vs_fp32_rel": float((y1.float() - ref).norm() / ref.norm()), "pair_vs_fp32_rel": float((p1.float() - ref).norm() / ref.norm()),
           "single_max_err_ulps": float(((y1.float() - ref).abs() / ulp).max()), "pair_max_err_ulps": float(((p1.float() - ref).abs() / ulp).max()),
           "pair_reproducible": bool(torch.equal(again[:m], p1))}
    ulp3 = torch.exp2(torch.floor(torch.log2(y3.float().abs().clamp_min(1e-30))) - 7)
    d3 = (y3.float() - p3.float()).abs() / ulp3
    big = y3.float().abs() > 1e-3 * y3.float().abs().max()
    row.update({"differ_3": int((y3 != p3).sum()), "max_diff_ulps_3": float(d3.max()), "max_diff_ulps_3_on_elements_above_1e-3_of_max": float(d3[big].max()),
                "rel_3": float((y3.float() - p3.float()).norm() / y3.float().norm())})
    # which rows / columns differ
    idx = (y3 != p3).nonzero()
    if idx.numel():
        row["rows_3"] = sorted(set(idx[:, 0].tolist()))[:12]
        row["cols_3_min_max"] = [int(idx[:, 1].min()), int(idx[:, 1].max())]
    print(json.dumps(row), flush=True)
