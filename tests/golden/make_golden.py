"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE
(torchao at /root/reference) in the build container.  Run once, commit the .npz:

    PYTHONPATH=/root/reference python tests/golden/make_golden.py

The GPU box has no /root/reference; tests only read the committed fixtures.
bf16 tensors are stored as uint16 bit patterns, fp8/e8m0 as uint8.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def bits(t):
    """bf16 tensor -> uint16 numpy"""
    return t.contiguous().view(torch.int16).numpy().view(np.uint16).copy()


def make_int4():
    from torchao.quantization.quant_primitives import (
        MappingType,
        _choose_qparams_affine_tinygemm,
        _quantize_affine_tinygemm,
    )
    from torchao.quantization.utils import (
        groupwise_affine_dequantize_tensor_from_qparams,
        pack_tinygemm_scales_and_zeros,
    )

    out = {}
    cases = [("g32", 32, 1024, 32, 1), ("g64", 32, 1024, 64, 3), ("g128", 48, 2048, 128, 5), ("g256", 16, 1024, 256, 17)]
    for name, n, k, g, m in cases:
        gen = torch.Generator().manual_seed(1234 + g)
        w = (torch.randn(n, k, generator=gen) * 0.02).to(torch.bfloat16)
        # edge cases the reference's formulas must survive
        w[0, :g] = 0.5          # constant group -> scale clamps to eps
        w[1, :g] = 0.0          # all-zero group
        w[2, :g] = torch.linspace(-3.0, 5.0, g).to(torch.bfloat16)  # wide range
        w[3, g:2 * g] = -w[3, g:2 * g].abs()  # all-negative group
        x = torch.randn(m, k, generator=gen).to(torch.bfloat16)
        s, z = _choose_qparams_affine_tinygemm(
            w, MappingType.ASYMMETRIC, (1, g), torch.int32, 0, 15,
            scale_dtype=torch.bfloat16, zero_point_dtype=torch.bfloat16,
        )
        q = _quantize_affine_tinygemm(w, [1, g], s, z, torch.int32, 0, 15)
        s2, z2 = s.reshape(n, -1), z.reshape(n, -1)
        sz = pack_tinygemm_scales_and_zeros(s2, z2, torch.bfloat16)
        byte_w = (q[:, ::2] << 4 | q[:, 1::2]).to(torch.uint8)
        dq = groupwise_affine_dequantize_tensor_from_qparams(q, s2, z2, 4, g)
        assert dq.dtype == torch.bfloat16
        y = torch.nn.functional.linear(x, dq)  # the reference's dequant -> bf16 matmul path
        out.update({
            f"{name}_w": bits(w), f"{name}_x": bits(x), f"{name}_scale": bits(s2), f"{name}_zero": bits(z2),
            f"{name}_q": q.numpy().astype(np.uint8), f"{name}_byte": byte_w.numpy(),
            f"{name}_sz": bits(sz), f"{name}_dq": bits(dq), f"{name}_y": bits(y),
            f"{name}_group": np.array(g),
        })
    np.savez_compressed(os.path.join(HERE, "int4_tinygemm.npz"), **out)
    print("int4_tinygemm.npz", sum(v.nbytes for v in out.values()), "bytes raw")


if __name__ == "__main__":
    torch.manual_seed(0)
    make_int4()
    if "make_rest" in globals():
        make_rest()  # noqa: F821
