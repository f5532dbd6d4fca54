"""Profiling aid (round 5): wall time per launch of the 128 x 128 / 32 x 32 x 16 int4 kernel and of its laboratory ablations (modes 941 no
MFMAs, 942 no dequant, 943 no A reads: AO_MI355_LIB=tools/bin/_C_mi355_lab.so), and the s_memtime stamps of the traced build (945).

    AO_MI355_LIB=tools/bin/_C_mi355_lab.so python tools/w32_trace.py N K [M]
"""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ao_amd import ops
from ao_amd._lib import lib as _load
from tools.midm_sweep import graph_time

lib = _load()
n, k = int(sys.argv[1]), int(sys.argv[2])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 128
copies = max(2, -(-(300 << 20) // (n * k // 2)))
ws = [ops.int4_quantize_tinygemm(torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.02, 128) for _ in range(copies)]
x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
for mode in (931, 941, 942, 943, 0):
    lib.ao_int4_set_tuning(0, mode)
    try:
        t = graph_time([lambda q=q, sz=sz: ops.weight_int4pack_mm(x, q, 128, sz) for q, sz in ws])
        print(f"M={m} N={n} K={k} mode {mode}: {t * 1e6:.2f} us")
    finally:
        lib.ao_int4_set_tuning(0, 0)
trace = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
lib.ao_int4_set_tuning(0, 945)
q, sz = ws[0]
for _ in range(3):
    ops.weight_int4pack_mm(x, q, 128, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.weight_int4pack_mm(x, q, 128, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(0))
lib.ao_int4_set_tuning(0, 0)
t = trace.cpu().view(-1, 16).numpy().astype(np.int64)
t = t[t[:, 0] != 0]
d = np.diff(t[:, 2:10], axis=1)
print(f"traced: {len(t)} workgroups; mean ticks between consecutive k-block barriers:", d.mean(axis=0).round(0))
print("per workgroup (ticks): prime", int((t[:, 1] - t[:, 0]).mean()), "first data", int((t[:, 2] - t[:, 1]).mean()), "loop", int((t[:, 10] - t[:, 2]).mean()),
      "tail", int((t[:, 12] - t[:, 10]).mean()))
