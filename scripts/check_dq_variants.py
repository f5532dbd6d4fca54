"""GPU: bit-exactness of the dequant variants of int4_mm_kernel (G=128, M=1).

x = e_k makes the mm return column k of the dequantised weight exactly; compare
with ao_int4_dequantize (validated bit-exact against the oracle in tests)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ao_amd import ops, _lib

torch.manual_seed(0)
n, k, g = 256, 1024, 128
w = (torch.randn(n, k) * 0.02).to(torch.bfloat16)
w[0, :128] = 0.5; w[1, :128] = 0; w[2, :128] = torch.linspace(-3, 5, 128).to(torch.bfloat16)
w[3] *= 1e-20; w[4] *= 1e10; w[5, 128:256] = -w[5, 128:256].abs()
w = w.cuda()
qdata, sz = ops.int4_quantize_tinygemm(w, g)
dq = ops.int4_dequantize(qdata, sz, g)
ks = list(range(0, 1024, 7)) + [1, 2, 3, 127, 128, 129, 1023]
for mode in [0, 22, 24]:
    _lib.lib().ao_int4_set_tuning(0, mode)
    bad = 0; tot = 0
    for kk in ks:
        x = torch.zeros(1, k, dtype=torch.bfloat16, device="cuda"); x[0, kk] = 1.0
        y = ops.weight_int4pack_mm(x, qdata, g, sz)
        eq = (y[0].view(torch.int16) == dq[:, kk].view(torch.int16))
        bad += int((~eq).sum()); tot += n
        if mode and (~eq).any() and bad < 6:
            i = int((~eq).nonzero()[0])
            print("  mismatch mode", mode, "k", kk, "n", i, float(y[0, i]), float(dq[i, kk]))
    xr = torch.randn(1, k, dtype=torch.bfloat16, device="cuda")
    _lib.lib().ao_int4_set_tuning(0, 0); y0 = ops.weight_int4pack_mm(xr, qdata, g, sz)
    _lib.lib().ao_int4_set_tuning(0, mode); y1 = ops.weight_int4pack_mm(xr, qdata, g, sz)
    print(f"mode {mode}: one-hot mismatches {bad}/{tot}; random-x equal to mode 0: {bool(torch.equal(y0, y1))}")
_lib.lib().ao_int4_set_tuning(0, 0)
