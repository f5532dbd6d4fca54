"""Profiling aid: s_memtime stamps of the fp8 rowwise mid-M (weight-streaming) kernel.

    python tools/fp8_rb_trace.py M N K [variant] [tuning, e.g. 1=32,2=4,3=1]
"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ao_amd import ops
from ao_amd._lib import lib as _load

lib = _load()
m, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 101
tun = dict((int(a), int(b)) for a, b in (kv.split("=") for kv in sys.argv[5].split(","))) if len(sys.argv) > 5 and sys.argv[5] else {}
x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.05
xq, xs = ops.fp8_quantize_rowwise(x)
wq, ws = ops.fp8_quantize_rowwise(w)
lib.ao_gemm8_set_variant(variant)
for key in (1, 2, 3, 4, 5, 6):
    lib.ao_gemm8_set_tuning(key, tun.get(key, 0))
for _ in range(3):
    ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t())
torch.cuda.synchronize()
trace = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
lib.ao_int4_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.fp8_scaled_mm(xq, wq.t(), xs, ws.t())
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(0))
lib.ao_gemm8_set_variant(0)
for key in (1, 2, 3, 4, 5, 6):
    lib.ao_gemm8_set_tuning(key, 0)
t = trace.cpu().view(-1, 16).numpy().astype(np.int64)
t = t[t[:, 0] != 0]
names = ["entry", "primed"] + [f"bar{i}" for i in range(8)] + ["loopdone", "met", "exit"]
print(f"M={m} N={n} K={k} variant={variant} tuning={tun}: {len(t)} workgroups; s_memtime ticks relative to each workgroup's entry")
for wg in [0, len(t) // 2, len(t) - 1]:
    row = t[wg]
    print(f"wg {wg:4d}: " + " ".join(f"{names[i]}={int(row[i] - row[0]) if row[i] else -1}" for i in range(13)))
d = np.diff(t[:, 2:10], axis=1)
d = d[(t[:, 2:10] != 0).all(axis=1)]
if len(d):
    print("mean ticks between consecutive step barriers:", d.mean(axis=0).round(0))
st = t[t[:, 11] != 0]
print("per workgroup (ticks): prime", int((t[:, 1] - t[:, 0]).mean()), "first data", int((t[:, 2] - t[:, 1]).mean()),
      "loop", int((t[:, 10] - t[:, 2]).mean()), "tail (meet + store)", int((t[:, 12] - t[:, 10]).mean()))
if len(st):
    print("storing workgroups:", len(st), "loopdone -> met", int((st[:, 11] - st[:, 10]).mean()), "met -> exit", int((st[:, 12] - st[:, 11]).mean()))
t0 = t[:, 0].min()
print("launch: first entry -> last entry", int(t[:, 0].max() - t0), "-> last loopdone", int(t[:, 10].max() - t0), "-> last exit", int(t[:, 12].max() - t0),
      "| entries by decile:", np.percentile(t[:, 0] - t0, [10, 50, 90, 100]).astype(int).tolist())
