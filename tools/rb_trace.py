"""Profiling aid: s_memtime stamps of the register-B batched int4 kernel (tuning mode 65S)."""
import sys, ctypes
import torch
sys.path.insert(0, ".")
from ao_amd import ops
from ao_amd._lib import lib as _load

lib = _load()
dev = "cuda"
m, n, k, g = 128, int(sys.argv[1]), int(sys.argv[2]), 128
split = int(sys.argv[3]) if len(sys.argv) > 3 else 1
w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
qdata, sz = ops.int4_quantize_tinygemm(w, g)
x = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
trace = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
lib.ao_int4_set_tuning(8, 650 + split)
for _ in range(3):
    ops.weight_int4pack_mm(x, qdata, g, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(trace.data_ptr()))
ops.weight_int4pack_mm(x, qdata, g, sz)
torch.cuda.synchronize()
lib.ao_int4_set_trace(ctypes.c_void_p(0))
lib.ao_int4_set_tuning(0, 0)
t = trace.cpu().view(-1, 16)
nwg = ((n + 127) // 128) * split
t = t[:nwg]
t0 = int(t[:, 0].min())
names = ["entry", "primed"] + [f"bar{i}" for i in range(8)] + ["loopdone", "met", "exit"]
print(f"N={n} K={k} split={split} workgroups={nwg}; s_memtime ticks relative to the first entry")
for wg in [0, 1, nwg // 2, nwg - 1]:
    row = t[wg]
    print(f"wg {wg:4d}: " + " ".join(f"{names[i]}={int(row[i] - t0) if row[i] else -1}" for i in range(13)))
import numpy as np
tt = t.numpy().astype(np.int64)
d = np.diff(tt[:, 2:10], axis=1)
print("mean ticks between consecutive k-block barriers:", d.mean(axis=0).round(0))
print("per workgroup (ticks): prime", int((tt[:, 1] - tt[:, 0]).mean()), "first data", int((tt[:, 2] - tt[:, 1]).mean()), "loop", int((tt[:, 10] - tt[:, 2]).mean()), "tail", int((tt[:, 12] - tt[:, 10]).mean()))
