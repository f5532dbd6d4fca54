"""Profiling aid: s_memtime stamps of the K-half-wave batched int4 kernel (tuning mode 887: 64 rows x 128 columns), k-blocks 8..11 of every workgroup.
    python tools/kh_trace.py [N K]"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ao_amd import ops
from ao_amd._lib import lib as _load

lib = _load()
dev = "cuda"
m, g = 128, 128
n, k = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14336, 4096)
w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
qdata, sz = ops.int4_quantize_tinygemm(w, g)
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
nwg = (n + 63) // 64
trace = torch.zeros(nwg * 8 * 3 * 16, dtype=torch.int64, device=dev)
lib.ao_int4_set_tuning(0, 887)
for _ in range(3):
    ops.weight_int4pack_mm(x, qdata, g, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.weight_int4pack_mm(x, qdata, g, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(0))
lib.ao_int4_set_tuning(0, 0)
t = trace.cpu().numpy().astype(np.int64).reshape(-1, 3, 16)
t = t[(t[:, 0, 0] != 0)]
print(f"N={n} K={k} M={m}: {len(t)} workgroups traced; mean s_memtime ticks, k-blocks 8..11")
names = {0: ["barrier passed", "dequant issued (first MFMA next)", "all MFMAs issued", "lgkmcnt(0), at the next barrier"],
         1: ["barrier passed", "(first MFMA next)", "all MFMAs issued", "prefetch + dequant(kb+1) done, at the next barrier"],
         2: ["vmcnt wait done", "barrier passed", "stage kb+3 DMAs issued", "-"]}
for wv, label in ((0, "consumer e=0"), (1, "consumer e=1"), (2, "producer")):
    a = t[:, wv, :].reshape(len(t), 4, 4)  # [wg][kb][stamp]
    per_kb = (a[:, 1:, 0] - a[:, :-1, 0]).mean()
    print(f"{label}: k-block period {per_kb:.0f} ticks")
    order = [0, 1, 2, 3] if wv != 2 else [0, 1, 2, 3]
    base = a[:, :, 0] if wv != 2 else a[:, :, 0]
    for i in range(4 if wv != 2 else 3):
        d = (a[:, :, i] - base).mean()
        print(f"    +{d:7.0f}  {names[wv][i]}")
