"""Run on the MI355X box: pin the oracle's int4 tile layout against PyTorch core.

aten::_convert_weight_to_int4pack lives in PyTorch core (int4mm.cu), not in the
reference checkout; on the GPU box it is the authority for the bit-exact pack.
Writes gpurun_out/int4pack_gfx950.npz (copy to tests/golden/) and prints how the
oracle and our HIP kernel compare with it, plus a loose cross-check of
aten::_weight_int4pack_mm against ours.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import int4_ref as R  # noqa: E402

out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
print("torch", torch.__version__, "device", torch.cuda.get_device_name(0), torch.cuda.get_device_capability(0))
props = torch.cuda.get_device_properties(0)
print("CUs", props.multi_processor_count, "mem GB", props.total_memory / 2**30, "gcn", getattr(props, "gcnArchName", "?"))

rng = np.random.default_rng(2024)
byte_w = rng.integers(0, 256, size=(64, 512), dtype=np.uint8)  # N=64, K=1024
bw = torch.from_numpy(byte_w).cuda()
try:
    packed = torch.ops.aten._convert_weight_to_int4pack(bw, 8)
    torch.cuda.synchronize()
    packed_np = packed.cpu().numpy()
    print("torch core pack: shape", tuple(packed.shape), packed.dtype)
    np.savez_compressed(os.path.join(out_dir, "int4pack_gfx950.npz"), byte_w=byte_w, packed=packed_np)
    ours = R.convert_weight_to_int4pack(byte_w)
    print("ORACLE == torch core pack:", bool(np.array_equal(ours, packed_np)))
    if not np.array_equal(ours, packed_np):
        # one-hot probe to print the true mapping
        for (nn, kk) in [(0, 0), (1, 0), (0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (0, 32), (0, 64), (0, 128), (8, 0), (16, 0)]:
            q = np.zeros((64, 1024), dtype=np.int32); q[nn, kk] = 15
            p = torch.ops.aten._convert_weight_to_int4pack(torch.from_numpy(R.nibble_pack(q)).cuda(), 8).cpu().numpy()
            nz = np.argwhere(p != 0)
            print("one-hot", (nn, kk), "->", nz.tolist(), [hex(int(p[tuple(i)]) & 0xffffffff) for i in nz])
    from ao_amd import ops
    mine = ops.convert_weight_to_int4pack(bw, 8)
    print("HIP kernel == torch core pack:", bool(torch.equal(mine, packed)))
except Exception as e:  # noqa: BLE001
    print("torch core _convert_weight_to_int4pack FAILED:", repr(e))
    packed = None

try:
    from ao_amd import ops
    g = 128
    torch.manual_seed(0)
    w = (torch.randn(64, 1024) * 0.02).to(torch.bfloat16).cuda()
    qdata, sz = ops.int4_quantize_tinygemm(w, g)
    x = torch.randn(4, 1024, dtype=torch.bfloat16, device="cuda")
    y_mine = ops.weight_int4pack_mm(x, qdata, g, sz)
    y_core = torch.ops.aten._weight_int4pack_mm(x, qdata, g, sz)
    torch.cuda.synchronize()
    rel = ((y_mine.float() - y_core.float()).norm() / y_core.float().norm()).item()
    print("ours vs torch core _weight_int4pack_mm rel diff:", rel)
    dq = ops.int4_dequantize(qdata, sz, g)
    y_dq = torch.nn.functional.linear(x, dq)
    print("ours vs dequant->bf16 matmul (hipBLASLt) rel diff:", ((y_mine.float() - y_dq.float()).norm() / y_dq.float().norm()).item())
    print("torch core vs dequant->bf16 matmul rel diff:", ((y_core.float() - y_dq.float()).norm() / y_dq.float().norm()).item())
except Exception as e:  # noqa: BLE001
    print("torch core _weight_int4pack_mm cross-check FAILED:", repr(e))
