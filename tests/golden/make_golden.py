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


def make_int8_fp8():
    from torchao.quantization.granularity import PerRow
    from torchao.quantization.quantize_.workflows.float8.float8_tensor import Float8Tensor
    from torchao.quantization.quantize_.workflows.int8.int8_tensor import Int8Tensor

    gen = torch.Generator().manual_seed(77)
    m, n, k = 19, 48, 256
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16)
    x[2] *= 300.0          # large row
    x[3] *= 1e-3           # small row
    x[4] = 0               # all-zero row (int8: scale clamps to eps; fp8: scale 0 -> NaN)
    w = (torch.randn(n, k, generator=gen) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, generator=gen).to(torch.bfloat16)
    out = {"x": bits(x), "w": bits(w), "bias": bits(bias)}
    xi, wi = Int8Tensor.from_hp(x, PerRow()), Int8Tensor.from_hp(w, PerRow())
    out.update(int8_xq=xi.qdata.numpy(), int8_xs=xi.scale.flatten().numpy(),
               int8_wq=wi.qdata.numpy(), int8_ws=wi.scale.flatten().numpy())
    # the reference's GPU-path epilogue formulas (int8_tensor.py:305-359, kernels.py:143-144) on CPU tensors
    c = (xi.qdata.to(torch.int32) @ wi.qdata.to(torch.int32).t())
    out["int8_c"] = c.numpy()
    y = (c * xi.scale.reshape(-1, 1)).to(torch.bfloat16)
    y = y * wi.scale.flatten()
    y = y + bias
    out["int8_y"] = bits(y.to(torch.bfloat16))
    xf = x.clone(); xf[4] = torch.randn(k, generator=gen).to(torch.bfloat16) * 1e-6
    out["fp8_x"] = bits(xf)
    xq, wq = Float8Tensor.from_hp(xf, torch.float8_e4m3fn, PerRow()), Float8Tensor.from_hp(w, torch.float8_e4m3fn, PerRow())
    out.update(fp8_xq=xq.qdata.view(torch.uint8).numpy(), fp8_xs=xq.scale.flatten().numpy(),
               fp8_wq=wq.qdata.view(torch.uint8).numpy(), fp8_ws=wq.scale.flatten().numpy())
    # dequant -> fp32 matmul with the reference's dequantize() (CPU _scaled_mm has no rowwise mode)
    yf = (xq.dequantize().float() @ wq.dequantize().float().t()) + bias.float()
    out["fp8_y_dequant_f32"] = yf.numpy()
    x0 = torch.zeros(2, k, dtype=torch.bfloat16)
    z = Float8Tensor.from_hp(x0, torch.float8_e4m3fn, PerRow())
    out.update(fp8_zero_q=z.qdata.view(torch.uint8).numpy(), fp8_zero_s=z.scale.flatten().numpy())
    np.savez_compressed(os.path.join(HERE, "int8_fp8.npz"), **out)
    print("int8_fp8.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_mx():
    from torchao.prototype.moe_training.mxfp8_grouped_mm import _emulated_mxfp8_scaled_grouped_mm_2d_3d
    from torchao.prototype.mx_formats.config import ScaleCalculationMode
    from torchao.prototype.mx_formats.mx_tensor import to_mx
    from torchao.testing import _mxfp8_test_utils as T

    out = {}
    # the reference's own bitwise contract (torchao/testing/_mxfp8_test_utils.py:22-235)
    for mode_name in ("rceil", "floor"):
        c = T.make_mxfp8_semantic_cases(torch.bfloat16, mode_name, device="cpu")
        out[f"sem_{mode_name}_x"] = bits(c.inputs)
        out[f"sem_{mode_name}_data"] = c.expected_data.numpy()
        out[f"sem_{mode_name}_scale"] = c.expected_scales.numpy()
        out[f"sem_{mode_name}_names"] = np.array(c.names)
    vals, exp = T.make_f32_to_e8m0_rceil_cases(device="cpu")
    out["e8m0_rceil_in"] = vals.numpy().view(np.uint32)
    out["e8m0_rceil_out"] = exp.numpy()
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(24, 256, generator=gen).to(torch.bfloat16)
    x[0] *= 1e4; x[1] *= 1e-6; x[2] = 0; x[3, :32] = 448.0; x[4, :32] = 449.0; x[5, 32:64] = -57344.0
    x[6, :32] = torch.tensor([2.0 ** (-i) for i in range(32)]).to(torch.bfloat16)
    x[7, :32] = 1.17549435e-38
    out["x"] = bits(x)
    for name, mode in (("rceil", ScaleCalculationMode.RCEIL), ("floor", ScaleCalculationMode.FLOOR)):
        s, d = to_mx(x, torch.float8_e4m3fn, 32, mode)
        out[f"{name}_scale"] = s.view(torch.uint8).numpy()
        out[f"{name}_data"] = d.view(torch.uint8).numpy()
    # grouped GEMM: E=3, ragged groups (one empty)
    E, N, K, M = 3, 32, 256, 40
    a = torch.randn(M, K, generator=gen).to(torch.bfloat16)
    w = (torch.randn(E, N, K, generator=gen) * 0.1).to(torch.bfloat16)
    offs = torch.tensor([16, 16, 40], dtype=torch.int32)
    a_s, a_d = to_mx(a, torch.float8_e4m3fn, 32, ScaleCalculationMode.RCEIL)
    w_s, w_d = to_mx(w, torch.float8_e4m3fn, 32, ScaleCalculationMode.RCEIL)
    # reference layout: B_data [E, K, N] (K-major), B_scale [E, K/32, N]
    y = _emulated_mxfp8_scaled_grouped_mm_2d_3d(
        a_d, a_s, w_d.transpose(-2, -1), w_s.transpose(-2, -1), offs=offs, out_dtype=torch.bfloat16
    )
    out.update(g_a=bits(a), g_w=bits(w), g_offs=offs.numpy(),
               g_a_data=a_d.view(torch.uint8).numpy(), g_a_scale=a_s.view(torch.uint8).numpy(),
               g_w_data=w_d.view(torch.uint8).numpy(), g_w_scale=w_s.view(torch.uint8).numpy(), g_y=bits(y))
    np.savez_compressed(os.path.join(HERE, "mx.npz"), **out)
    print("mx.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_moe():
    """Outputs of the reference's torch_pad_token_groups / torch_unpad_token_groups (the checker of its CUDA kernels)."""
    from torchao.prototype.moe_training.kernels.mxfp8.quant import torch_pad_token_groups, torch_unpad_token_groups

    out = {}
    gen = torch.Generator().manual_seed(11)
    cases = {
        "ragged": ([5, 5, 37, 64, 64, 70], 32, 24, torch.bfloat16),   # empty groups, one already aligned
        "aligned16": ([16, 48, 64], 16, 40, torch.float32),
        "single": ([3], 32, 8, torch.bfloat16),
        "odd_dim": ([2, 9, 9, 20], 4, 7, torch.bfloat16),
    }
    for name, (ends, align, dim, dtype) in cases.items():
        x = torch.randn(ends[-1], dim, generator=gen).to(dtype)
        offs = torch.tensor(ends, dtype=torch.int32)
        p, s, e = torch_pad_token_groups(x, offs, align)
        u = torch_unpad_token_groups(p, offs, s, ends[-1], align)
        assert torch.equal(u, x)
        as_np = (lambda t: bits(t)) if dtype == torch.bfloat16 else (lambda t: t.numpy())
        out.update({f"{name}_x": as_np(x), f"{name}_offs": offs.numpy(), f"{name}_align": np.int32(align),
                    f"{name}_padded": as_np(p), f"{name}_starts": s.numpy().astype(np.int32), f"{name}_ends": e.numpy().astype(np.int32)})
    np.savez_compressed(os.path.join(HERE, "moe_pad.npz"), **out)
    print("moe_pad.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_mx_blocked():
    """Outputs of the reference's to_blocked (prototype/mx_formats/utils.py:31-72) and torch_to_blocked_2d_M_groups
    (moe_training/kernels/mxfp8/quant.py:136-196: the checker of its CUDA op torchao::mx_block_rearrange_2d_M_groups)."""
    from torchao.prototype.moe_training.kernels.mxfp8.quant import torch_to_blocked_2d_M_groups
    from torchao.prototype.mx_formats.utils import from_blocked, to_blocked

    out = {}
    gen = torch.Generator().manual_seed(23)
    for name, (h, w) in {"one_block": (128, 4), "ragged": (200, 10), "wide": (130, 68), "tiny": (1, 1), "k4096": (256, 128)}.items():
        s = torch.randint(0, 256, (h, w), generator=gen, dtype=torch.uint8)
        b = to_blocked(s)
        assert torch.equal(from_blocked(b, h, w), s)
        out.update({f"tb_{name}_in": s.numpy(), f"tb_{name}_out": b.numpy()})
    cases = {
        "groups": ([5, 5, 133, 256, 256, 300], 16),     # empty groups, a group that is exactly one block, one of 44 rows
        "single": ([70], 8),
        "aligned": ([128, 384], 128),
        "k224": ([3, 40, 41], 7 * 4),
    }
    for name, (ends, cols) in cases.items():
        s = torch.randint(0, 256, (ends[-1], cols), generator=gen, dtype=torch.uint8)
        offs = torch.tensor(ends, dtype=torch.int32)
        blocked, starts = torch_to_blocked_2d_M_groups(s, offs)
        out.update({f"mg_{name}_in": s.numpy(), f"mg_{name}_offs": offs.numpy(), f"mg_{name}_out": blocked.numpy(),
                    f"mg_{name}_starts": starts.numpy().astype(np.int64)})
    np.savez_compressed(os.path.join(HERE, "mx_blocked.npz"), **out)
    print("mx_blocked.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_int4_plain():
    """Int4Tensor (PLAIN) weight preparation: mslk is absent, so the fixtures come from the reference's OWN restatements of its
    numerics -- Int4WeightFakeQuantizer (qat/fake_quantizer.py:148-190) and the GPTQ helpers (prototype/gptq/api.py:167-221)."""
    from torchao.prototype.gptq.api import _int4_row_dequantize_zp, _int4_row_quantize_zp_precomputed_qparams
    from torchao.quantization.qat.fake_quantize_config import Int4WeightFakeQuantizeConfig
    from torchao.quantization.qat.fake_quantizer import Int4WeightFakeQuantizer

    out = {}
    for name, n, k, g in [("g32", 16, 256, 32), ("g128", 32, 1024, 128), ("g256", 8, 512, 256)]:
        gen = torch.Generator().manual_seed(4321 + g)
        w = (torch.randn(n, k, generator=gen) * 0.02).to(torch.bfloat16)
        w[0, :g] = 0.25  # constant group: max - min clamps at 1e-6
        w[1, :g] = 0.0
        w[2, :g] = torch.linspace(-3.0, 5.0, g).to(torch.bfloat16)
        fq_zp = Int4WeightFakeQuantizer(Int4WeightFakeQuantizeConfig(group_size=g, activation_dtype=torch.bfloat16))(w)
        out[f"{name}_w"] = bits(w)
        out[f"{name}_fq_zp"] = bits(fq_zp)  # bf16((q * scale + zero) in fp32)
        # scale / zero as the fake quantizer computes them (fp32), then the GPTQ helpers' quantize / dequantize given them
        wg = w.float().view(n, -1, g)
        mx, mn = wg.amax(-1, keepdim=True), wg.amin(-1, keepdim=True)
        scale = torch.clamp(mx - mn, min=1e-6) / 15
        zero = mn + scale * 8
        s_t, z_t = scale.view(n, -1).t().contiguous(), zero.view(n, -1).t().contiguous()  # [K/g, N] like mslk returns them
        q = _int4_row_quantize_zp_precomputed_qparams(w, s_t, z_t, g)
        out[f"{name}_scale_f32"] = s_t.numpy()
        out[f"{name}_zero_f32"] = z_t.numpy()
        out[f"{name}_q"] = q.numpy()
        out[f"{name}_dq_f32"] = _int4_row_dequantize_zp(q, s_t, z_t, g).numpy()
    np.savez_compressed(os.path.join(HERE, "int4_plain.npz"), **out)
    print("int4_plain.npz:", len(out), "arrays")


def make_hqq():
    """HQQ qparams as Int4TilePackedTo4dTensor.from_hp computes them (int4_tile_packed_to_4d_tensor.py:149-168), with the
    proximal optimizer's dtype forced to float16 -- what the reference uses on a GPU (quant_primitives.py:1832-1833) -- while
    running here on the CPU."""
    from functools import partial

    from torchao.quantization.quant_primitives import _choose_qparams_and_quantize_affine_hqq, optimize_weights_proximal_legacy

    out = {}
    for name, n, k, g in [("g64", 16, 512, 64), ("g128", 32, 1024, 128)]:
        gen = torch.Generator().manual_seed(777 + g)
        w = (torch.randn(n, k, generator=gen) * 0.02).to(torch.bfloat16)
        w[2, :g] = torch.linspace(-1.0, 2.0, g).to(torch.bfloat16)
        q, scale, zero, _ = _choose_qparams_and_quantize_affine_hqq(
            w, nbits=4, group_size=g, axis=1, compute_dtype=torch.bfloat16, device="cpu", verbose=False, raw_output=False,
            optimize_weights=partial(optimize_weights_proximal_legacy, dtype=torch.float16))
        out[f"{name}_w"] = bits(w)
        out[f"{name}_q"] = q.numpy().astype(np.uint8)
        out[f"{name}_scale"] = bits(scale.reshape(n, -1))
        out[f"{name}_zero"] = bits(zero.reshape(n, -1))
    np.savez_compressed(os.path.join(HERE, "hqq.npz"), **out)
    print("hqq.npz:", len(out), "arrays")


def make_int8_fp8_variants():
    """PerTensor granularity (int8, fp8) and the ASYMMETRIC activation mapping of Int8Tensor, by the reference's own from_hp and its
    own F.linear dispatch on CPU tensors (int8_tensor.py:176-359)."""
    import torch.nn.functional as F
    from torchao.quantization.granularity import PerRow, PerTensor
    from torchao.quantization.quant_primitives import MappingType
    from torchao.quantization.quantize_.workflows.float8.float8_tensor import Float8Tensor
    from torchao.quantization.quantize_.workflows.int8.int8_tensor import Int8Tensor, QuantizeTensorToInt8Kwargs

    gen = torch.Generator().manual_seed(91)
    m, n, k = 21, 64, 384
    x = torch.randn(m, k, generator=gen).to(torch.bfloat16)
    x[1] = x[1].abs() + 0.5       # all-positive row: min_val_neg = 0, zero_point = -128
    x[2] = -x[2].abs() - 0.25     # all-negative row: max_val_pos = 0, zero_point = 127
    x[3] *= 250.0
    x[4] *= 2e-3
    x[5] = 0                      # scale clamps to eps, zero_point = -128
    x[6] = x[6] + 3.0             # skewed
    x[7, 0] = 1000.0              # one outlier
    w = (torch.randn(n, k, generator=gen) * 0.05).to(torch.bfloat16)
    bias = torch.randn(n, generator=gen).to(torch.bfloat16)
    out = {"x": bits(x), "w": bits(w), "bias": bits(bias)}
    # --- int8 asymmetric per-row activation
    xa = Int8Tensor.from_hp(x, PerRow(), mapping_type=MappingType.ASYMMETRIC)
    out.update(asym_xq=xa.qdata.numpy(), asym_xs=xa.scale.flatten().numpy(), asym_xzp=xa.zero_point.flatten().numpy().astype(np.int8))
    wt = Int8Tensor.from_hp(w, PerRow(), act_quant_kwargs=QuantizeTensorToInt8Kwargs(granularity=PerRow(), mapping_type=MappingType.ASYMMETRIC))
    out.update(asym_wq=wt.qdata.numpy(), asym_ws=wt.scale.flatten().numpy())
    out["asym_y"] = bits(F.linear(x, wt, bias))
    out["asym_y_nobias"] = bits(F.linear(x, wt))
    # --- int8 per-tensor (symmetric), both operands
    xt = Int8Tensor.from_hp(x, PerTensor())
    wtt = Int8Tensor.from_hp(w, PerTensor(), act_quant_kwargs=QuantizeTensorToInt8Kwargs(granularity=PerTensor()))
    out.update(pt_xq=xt.qdata.numpy(), pt_xs=xt.scale.flatten().numpy(), pt_wq=wtt.qdata.numpy(), pt_ws=wtt.scale.flatten().numpy())
    out["pt_y"] = bits(F.linear(x, wtt, bias))
    # --- fp8 per-tensor
    xf = Float8Tensor.from_hp(x, torch.float8_e4m3fn, PerTensor())
    wf = Float8Tensor.from_hp(w, torch.float8_e4m3fn, PerTensor())
    out.update(fp8pt_xq=xf.qdata.view(torch.uint8).numpy(), fp8pt_xs=xf.scale.flatten().numpy(),
               fp8pt_wq=wf.qdata.view(torch.uint8).numpy(), fp8pt_ws=wf.scale.flatten().numpy())
    out["fp8pt_y_dequant_f32"] = ((xf.dequantize().float() @ wf.dequantize().float().t()) + bias.float()).numpy()
    # --- fp8 per-row with the activation-value bounds of Float8DynamicActivationFloat8WeightConfig(activation_value_lb / _ub)
    lb, ub = 0.37, 900.0  # lb is not bf16-representable on purpose; ub cuts the amax of rows 3 and 7
    xc = Float8Tensor.from_hp(x, torch.float8_e4m3fn, PerRow(), hp_value_lb=lb, hp_value_ub=ub)
    out.update(fp8clamp_xq=xc.qdata.view(torch.uint8).numpy(), fp8clamp_xs=xc.scale.flatten().numpy(), fp8clamp_bounds=np.float64([lb, ub]))
    # --- int8 STATIC activation quantization (Int8StaticActivationInt8WeightConfig): the activation scale (and zero-point) are given
    s_static = torch.tensor([[0.0625]], dtype=torch.float32)          # PerTensor activation scale
    ws_ = Int8Tensor.from_hp(w, PerRow(), act_quant_kwargs=QuantizeTensorToInt8Kwargs(granularity=PerTensor()), act_quant_scale=s_static)
    out["static_sym_scale"] = s_static.numpy()
    out["static_sym_xq"] = Int8Tensor.from_hp(x, PerTensor(), scale=s_static).qdata.numpy()
    out["static_sym_y"] = bits(F.linear(x, ws_, bias))
    s_asym, zp_asym = torch.tensor([[0.04]], dtype=torch.float32), torch.tensor([[-19]], dtype=torch.int8)
    wa_ = Int8Tensor.from_hp(w, PerRow(), act_quant_kwargs=QuantizeTensorToInt8Kwargs(granularity=PerTensor(), mapping_type=MappingType.ASYMMETRIC),
                             act_quant_scale=s_asym, act_quant_zero_point=zp_asym)
    out["static_asym_scale"], out["static_asym_zp"] = s_asym.numpy(), zp_asym.numpy()
    out["static_asym_xq"] = Int8Tensor.from_hp(x, PerTensor(), mapping_type=MappingType.ASYMMETRIC, scale=s_asym, zero_point=zp_asym).qdata.numpy()
    out["static_asym_y"] = bits(F.linear(x, wa_, bias))
    np.savez_compressed(os.path.join(HERE, "int8_fp8_variants.npz"), **out)
    print("int8_fp8_variants.npz:", {k_: v.shape for k_, v in out.items()})


def make_moe_permute():
    """ep/kernels.py generate_permute_indices (its own CPU restatement, use_cpu=True) + ep/permute.py permute_and_pad +
    ep/unpermute.py _unpermute_bf16 on seeded routing tables: the fixtures of the HIP regrouping kernels."""
    from torchao.prototype.moe_training.ep.kernels import generate_permute_indices
    from torchao.prototype.moe_training.ep.permute import permute_and_pad as _unused  # noqa: F401  (import check: same module the mirror cites)

    g = torch.Generator().manual_seed(7)
    out = {}
    cases = [(4, 2, 32, 24), (3, 5, 16, 16), (8, 1, 32, 8), (2, 8, 32, 40)]  # (experts per rank, ranks, alignment, dim)
    for ci, (E, R, align, dim) in enumerate(cases):
        counts = torch.randint(0, 9, (R * E,), generator=g, dtype=torch.int32)
        if ci == 1:
            counts.view(R, E)[:, 2] = 0  # an expert nobody routed to: one aligned block of padding
        T = int(counts.sum())
        max_len = (T + E * align + align - 1) // align * align
        idx, m_sizes, m_offsets = generate_permute_indices(counts, E, R, max_len, align, use_cpu=True)
        x = torch.randn(T, dim, generator=g).to(torch.bfloat16)
        xp = torch.vstack((x, x.new_zeros((1, dim))))
        permuted = xp[idx.long(), :]
        y = torch.randn(max_len, dim, generator=g).to(torch.bfloat16)
        un = y.new_zeros((T + 1, dim))
        un[idx.long(), :] = y
        out[f"c{ci}_meta"] = np.array([E, R, align, dim, T, max_len], dtype=np.int64)
        out[f"c{ci}_counts"] = counts.numpy()
        out[f"c{ci}_idx"], out[f"c{ci}_m_sizes"], out[f"c{ci}_m_offsets"] = idx.numpy(), m_sizes.numpy(), m_offsets.numpy()
        out[f"c{ci}_x"], out[f"c{ci}_permuted"] = bits(x), bits(permuted)
        out[f"c{ci}_y"], out[f"c{ci}_unpermuted"] = bits(y), bits(un[:-1])
    np.savez_compressed(os.path.join(HERE, "moe_permute.npz"), **out)
    print("moe_permute.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_other_dtypes():
    """fp16 / fp32 ACTIVATIONS on the dynamic-activation 8-bit linears (the reference quantizes them in their own dtype): the
    reference's own quantize_() + F.linear on CPU tensors for int8; for fp8 -- whose linear asserts a GPU -- its Float8Tensor.from_hp
    on activation and weight and aten::_scaled_mm's arithmetic (acc * scale_a * scale_b, out in the activation dtype) on its codes."""
    from torchao.quantization import Int8DynamicActivationInt8WeightConfig, PerRow, quantize_
    from torchao.quantization.quantize_.workflows.float8.float8_tensor import Float8Tensor

    gen = torch.Generator().manual_seed(21)
    out = {}
    n, k = 64, 256
    w = (torch.randn(n, k, generator=gen) * 0.05).to(torch.bfloat16)
    out["w"] = bits(w)
    for name, dt in (("f16", torch.float16), ("f32", torch.float32)):
        x = (torch.randn(5, k, generator=gen) * 2).to(dt)
        x[0, :4] = torch.tensor([0.0, -0.0, 1e-4, 60000.0 if dt == torch.float16 else 3e30]).to(dt)
        out[f"x_{name}"] = x.numpy().copy()
        # nn.Linear keeps its weight in the activation dtype, as a model in that dtype would: the reference quantizes THAT tensor
        lin = torch.nn.Linear(k, n, bias=False, dtype=dt)
        with torch.no_grad():
            lin.weight.copy_(w.to(dt))
        quantize_(lin, Int8DynamicActivationInt8WeightConfig())
        out[f"int8_wq_{name}"] = lin.weight.qdata.numpy().copy()
        out[f"int8_ws_{name}"] = lin.weight.scale.float().numpy().copy()
        with torch.no_grad():
            out[f"int8_y_{name}"] = lin(x).float().numpy().copy()
        wq = Float8Tensor.from_hp(w, torch.float8_e4m3fn, PerRow())
        xq = Float8Tensor.from_hp(x, torch.float8_e4m3fn, PerRow())
        out[f"fp8_xq_{name}"] = xq.qdata.view(torch.uint8).numpy().copy()
        out[f"fp8_xs_{name}"] = xq.scale.float().numpy().copy()
        acc = xq.qdata.float() @ wq.qdata.float().t()
        out[f"fp8_y_{name}"] = (acc * xq.scale.float() * wq.scale.float().t()).to(dt).float().numpy().copy()
    out["fp8_wq"] = wq.qdata.view(torch.uint8).numpy().copy()
    out["fp8_ws"] = wq.scale.float().numpy().copy()
    np.savez_compressed(os.path.join(HERE, "other_dtypes.npz"), **out)
    print("other_dtypes.npz", sum(v.nbytes for v in out.values()), "bytes raw")


def make_configs():
    """torchao/core/config.py:69-305 config_to_dict: the {"_type", "_version", "_data"} dicts the REFERENCE writes for the configs of
    the SURVEY section-8 path (tests/test_host_api.py decodes them with this package's config_from_dict and compares its own encoder's
    output with them)."""
    import json

    from torchao.core.config import config_to_dict
    from torchao.quantization import (Float8DynamicActivationFloat8WeightConfig, Float8DynamicActivationInt4WeightConfig, FqnToConfig,
                                      Int4WeightOnlyConfig, Int8DynamicActivationInt8WeightConfig, PerRow)
    from torchao.quantization.quant_primitives import MappingType

    cfgs = {
        "int4_default": Int4WeightOnlyConfig(),
        "int4_tile": Int4WeightOnlyConfig(group_size=32, int4_packing_format="tile_packed_to_4d"),
        "int4_hqq": Int4WeightOnlyConfig(group_size=64, int4_packing_format="tile_packed_to_4d", int4_choose_qparams_algorithm="hqq"),
        "int8_dyn": Int8DynamicActivationInt8WeightConfig(),
        "int8_dyn_asym": Int8DynamicActivationInt8WeightConfig(act_mapping_type=MappingType.ASYMMETRIC, granularity=[PerRow(), PerRow()]),
        "fp8_default": Float8DynamicActivationFloat8WeightConfig(),
        "fp8_row": Float8DynamicActivationFloat8WeightConfig(granularity=PerRow()),
        "fp8_int4": Float8DynamicActivationInt4WeightConfig(),
    }
    cfgs["fqn"] = FqnToConfig({"re:.*q_proj": Int4WeightOnlyConfig(group_size=64), "lm_head": None,
                               "_default": Float8DynamicActivationFloat8WeightConfig(granularity=PerRow())})
    out = {k: config_to_dict(v) for k, v in cfgs.items()}
    with open(os.path.join(HERE, "configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("configs.json", len(out), "configs")


def make_rest():
    make_configs()
    make_other_dtypes()
    make_int8_fp8()
    make_int8_fp8_variants()
    make_mx()
    make_moe()
    make_mx_blocked()
    make_moe_permute()
    make_int4_plain()
    make_hqq()


if __name__ == "__main__":
    torch.manual_seed(0)
    make_int4()
    if "make_rest" in globals():
        make_rest()  # noqa: F821
