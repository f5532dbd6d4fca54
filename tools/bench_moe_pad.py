"""Micro-bench: MoE token-group pad / unpad kernels (HBM-bound row copies).  The inputs rotate over three distinct 235 MB buffers, so
that the 256 MiB Infinity Cache cannot serve the reads (a single re-used input measured 6.5-6.7 TB/s in round 1: cache-warm)."""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ao_amd import ops

tokens, dim, groups, align = 16384, 7168, 8, 32
rng = np.random.default_rng(0)
cuts = np.sort(rng.integers(0, tokens, size=groups - 1))
ends = torch.tensor(list(cuts) + [tokens], dtype=torch.int32, device="cuda")
xs = [torch.randn(tokens, dim, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
x = xs[0]


def time(fn, reps=21):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


p, st, en = ops.fused_pad_token_groups(x, ends, align)
ps = [ops.fused_pad_token_groups(xi, ends, align)[0] for xi in xs]
t_pad = time(lambda i: ops.fused_pad_token_groups(xs[i % 3], ends, align))
t_unpad = time(lambda i: ops.fused_unpad_token_groups(ps[i % 3], ends, st, tokens, align))
payload = tokens * dim * 2
print(json.dumps({
    "workload": f"pad/unpad {tokens} x {dim} bf16 tokens, {groups} groups, alignment {align}",
    "pad_us": t_pad, "pad_GBps": (payload + p.numel() * 2) / t_pad / 1e3,
    "unpad_us": t_unpad, "unpad_GBps": 2 * payload / t_unpad / 1e3,
    "note": "bytes = rows read + rows written (pad also writes the zero rows); includes the output allocation; inputs rotate over 3 x 235 MB (cold reads)"}))
