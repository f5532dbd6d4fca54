"""Development aid: one mid8 call per process (so that a hang costs one short timeout).  python tools/mid8_debug.py <mode> <variant> [m n k]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ao_amd import _lib, ops
mode, variant = sys.argv[1], int(sys.argv[2])
m, n, k = (int(v) for v in sys.argv[3:6]) if len(sys.argv) > 5 else (128, 1280, 8192)
lib = _lib.lib()
torch.manual_seed(0)
x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16)
w = torch.randn(n, k, device="cuda", dtype=torch.bfloat16) * 0.05
wq, ws = ops.int8_quantize_rowwise(w)
xq, xs = ops.int8_quantize_rowwise(x)
lib.ao_gemm8_set_variant(300)
ref = ops.int8_scaled_mm(xq, xs, wq, ws)
torch.cuda.synchronize()
lib.ao_gemm8_set_variant(variant)
t = time.time()
if mode == "mm":
    y = ops.int8_scaled_mm(xq, xs, wq, ws)
else:
    y = ops.int8_dynamic_linear(x, wq, ws)
torch.cuda.synchronize()
print(mode, variant, (m, n, k), "equal" if torch.equal(y, ref) else f"DIFF max {float((y.float() - ref.float()).abs().max())}", f"{time.time() - t:.3f}s", flush=True)
