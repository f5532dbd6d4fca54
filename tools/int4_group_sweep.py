#!/usr/bin/env python3
"""int4 tinygemm linears over the group sizes the reference supports (g = 32 / 64 / 128 / 256) at M = 1 .. 128 on the Llama-3-8B shapes, COLD
weights, us per call in a hipGraph: g = 32 carries 25 % more bytes than g = 128 (a dword of qparams per 32 weights) and should cost about that.

    python tools/int4_group_sweep.py > profiles/int4_group_sweep_rNN.jsonl
"""
import json, os, sys, torch
sys.path.insert(0, os.getcwd())
from ao_amd import ops
from tools.bench_dec8 import graph_time
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for name, n, k in [("qkv", 6144, 4096), ("o", 4096, 4096), ("gate", 14336, 4096), ("down", 4096, 14336)]:
    for g in (32, 64, 128, 256):
        copies = max(2, -(-(300 << 20) // (n * k // 2)))
        ws = []
        for _ in range(copies):
            w = torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.05
            ws.append(ops.int4_quantize_tinygemm(w, g)); del w
        for m in (1, 4, 16, 64, 128):
            x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
            t = graph_time([lambda q=q, sz=sz: ops.weight_int4pack_mm(x, q, g, sz) for q, sz in ws])
            nbytes = n * k // 2 + (k // g) * n * 4
            print(json.dumps({"shape": name, "g": g, "M": m, "us": round(t * 1e6, 2), "TBps": round(nbytes / t / 1e12, 2)}), flush=True)
        del ws
