"""GPU parity: PerTensor granularity (int8, fp8) and ASYMMETRIC int8 activations -- kernels through the C ABI and the
Int8Tensor / Float8Tensor mirrors against the oracle and the reference-generated fixtures (tests/golden/int8_fp8_variants.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, bf16_bits_to_f32, np_from_torch_bf16, torch_bf16_from_f32
from oracle import bf16, fp8_ref as F, int8_ref as I

pytestmark = pytest.mark.gpu

from ao_amd import ops  # noqa: E402
from ao_amd.quantization import (  # noqa: E402
    Float8DynamicActivationFloat8WeightConfig,
    Float8Tensor,
    Int8DynamicActivationInt8WeightConfig,
    Int8StaticActivationInt8WeightConfig,
    Int8Tensor,
    MappingType,
    PerRow,
    PerTensor,
    quantize_,
)

DEV = "cuda"


def _rel(a, b):
    a, b = a.astype(np.float64), b.astype(np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def _randn_bf16(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16)


def _bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(GOLDEN, "int8_fp8_variants.npz"))


def _t(gv, name):
    return torch_bf16_from_f32(bf16_bits_to_f32(gv[name])).to(DEV)


def test_int8_asym_quantize_golden(gv):
    q, s, zp = ops.int8_quantize_rowwise_asym(_t(gv, "x"))
    assert np.array_equal(q.cpu().numpy(), gv["asym_xq"])
    assert np.array_equal(s.flatten().cpu().numpy(), gv["asym_xs"])
    assert np.array_equal(zp.flatten().cpu().numpy(), gv["asym_xzp"])


@pytest.mark.parametrize("m,k", [(1, 4096), (9, 128), (200, 14336), (33, 8)])
def test_int8_asym_quantize_vs_oracle(m, k):
    x = _randn_bf16((m, k), 3 * m + k, 2.0)
    x[0] = x[0].abs()
    if m > 2:
        x[1] = -x[1].abs()
        x[2] = 0
    q, s, zp = ops.int8_quantize_rowwise_asym(x.to(DEV))
    qo, so, zo = I.quantize_rowwise_asym(x.float().numpy())
    assert np.array_equal(s.flatten().cpu().numpy(), so)
    assert np.array_equal(zp.flatten().cpu().numpy(), zo)
    assert np.array_equal(q.cpu().numpy(), qo)


@pytest.mark.parametrize("n,k", [(16, 16), (48, 256), (4096, 14336), (130, 1040)])
def test_int8_row_sums_exact(n, k):
    g = torch.Generator().manual_seed(n + k)
    wq = torch.randint(-128, 128, (n, k), generator=g, dtype=torch.int8)
    got = ops.int8_row_sums(wq.to(DEV)).cpu().numpy()
    assert np.array_equal(got, wq.numpy().astype(np.int64).sum(axis=1).astype(np.int32))


@pytest.mark.parametrize("with_bias", [True, False])
def test_int8_asym_linear_golden_through_the_subclass(gv, with_bias):
    w = _t(gv, "w")
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=with_bias, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w)
        if with_bias:
            lin.bias.copy_(_t(gv, "bias"))
    quantize_(lin, Int8DynamicActivationInt8WeightConfig(act_mapping_type=MappingType.ASYMMETRIC))
    assert isinstance(lin.weight, Int8Tensor) and lin.weight.act_quant_kwargs.mapping_type == MappingType.ASYMMETRIC
    assert np.array_equal(lin.weight.qdata.cpu().numpy(), gv["asym_wq"])
    y = lin(_t(gv, "x"))
    assert np.array_equal(_bits(y), gv["asym_y" if with_bias else "asym_y_nobias"])  # bit-exact vs the reference's own F.linear


@pytest.mark.parametrize("m,n,k", [(1, 64, 256), (5, 48, 1024), (130, 208, 1040), (512, 512, 4096)])
def test_int8_asym_linear_vs_oracle(m, n, k):
    x = _randn_bf16((m, k), 7 * m + k) + 0.75  # skewed: the zero-point matters
    w = _randn_bf16((n, k), 5 * n + k, 0.05)
    b = _randn_bf16((n,), 9)
    wq, ws = ops.int8_quantize_rowwise(w.to(DEV))
    y = ops.int8_linear_asym(x.to(DEV), wq, ws, ops.int8_row_sums(wq), b.to(DEV))
    y_ref = I.linear_asym(x.float().numpy(), w.float().numpy(), b.float().numpy())
    assert np.array_equal(_bits(y), bf16.to_bits(y_ref))


def test_int8_per_tensor_golden(gv):
    xq, xs = ops.int8_quantize_tensorwise(_t(gv, "x"))
    assert xs.shape == (1, 1) and np.array_equal(xq.cpu().numpy(), gv["pt_xq"]) and float(xs) == float(gv["pt_xs"][0])
    lin = torch.nn.Linear(gv["w"].shape[1], gv["w"].shape[0], bias=True, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(_t(gv, "w"))
        lin.bias.copy_(_t(gv, "bias"))
    quantize_(lin, Int8DynamicActivationInt8WeightConfig(granularity=PerTensor()))
    wt = lin.weight
    assert wt.scale.shape == (1, 1) and wt.block_size == list(wt.shape)
    assert np.array_equal(wt.qdata.cpu().numpy(), gv["pt_wq"]) and float(wt.scale) == float(gv["pt_ws"][0])
    assert np.array_equal(_bits(lin(_t(gv, "x"))), gv["pt_y"])  # bit-exact vs the reference's own F.linear
    assert _rel(np_from_torch_bf16(wt.dequantize()), bf16_bits_to_f32(gv["w"])) < 0.05


def test_int8_mixed_granularity_linear_vs_oracle():
    """[PerTensor activation, PerRow weight]: the reference allows mixing for int8"""
    m, n, k = 37, 96, 512
    x, w = _randn_bf16((m, k), 1), _randn_bf16((n, k), 2, 0.05)
    lin = torch.nn.Linear(k, n, bias=False, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w)
    quantize_(lin, Int8DynamicActivationInt8WeightConfig(granularity=[PerTensor(), PerRow()]))
    y = lin(x.to(DEV))
    xq, xs = I.quantize_tensorwise(x.float().numpy())
    wq, ws = I.quantize_rowwise(w.float().numpy())
    y_ref = I.scaled_mm(xq, np.full(m, xs, np.float32), wq, ws)
    assert np.array_equal(_bits(y), bf16.to_bits(y_ref))


def test_fp8_per_tensor_golden_and_default_config(gv):
    for t in ("x", "w"):
        q, s = ops.fp8_quantize_tensorwise(_t(gv, t))
        assert s.shape == (1, 1) and float(s) == float(gv[f"fp8pt_{t}s"][0])
        assert np.array_equal(q.view(torch.uint8).cpu().numpy(), gv[f"fp8pt_{t}q"])
    lin = torch.nn.Linear(gv["w"].shape[1], gv["w"].shape[0], bias=True, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(_t(gv, "w"))
        lin.bias.copy_(_t(gv, "bias"))
    quantize_(lin, Float8DynamicActivationFloat8WeightConfig())  # the reference's default: PerTensor for both operands
    wt = lin.weight
    assert isinstance(wt, Float8Tensor) and wt.scale.shape == (1, 1) and wt.block_size == list(wt.shape)
    assert isinstance(wt.act_quant_kwargs.granularity, PerTensor)
    y = np_from_torch_bf16(lin(_t(gv, "x")))
    assert _rel(y, gv["fp8pt_y_dequant_f32"]) < 3e-3  # vs the reference's dequantize() -> fp32 matmul; bf16 output rounding
    x, w = bf16_bits_to_f32(gv["x"]), bf16_bits_to_f32(gv["w"])
    xq, xs = F.quantize_tensorwise(x)
    wq, ws = F.quantize_tensorwise(w)
    y_ref = F.scaled_mm(xq, wq, np.full(x.shape[0], xs, np.float32), np.full(w.shape[0], ws, np.float32), bf16_bits_to_f32(gv["bias"]))
    assert _rel(y, y_ref) <= 1e-3  # BASELINE.json tolerance


@pytest.mark.parametrize("m,n,k", [(1, 1024, 8192), (128, 7168, 8192), (300, 256, 1024)])
def test_fp8_per_tensor_linear_vs_oracle(m, n, k):
    x, w = _randn_bf16((m, k), m + k), _randn_bf16((n, k), n + k, 0.03)
    wt = Float8Tensor.from_hp(w.to(DEV), granularity=PerTensor(),
                              act_quant_kwargs=__import__("ao_amd.quantization", fromlist=["x"]).QuantizeTensorToFloat8Kwargs(granularity=PerTensor()))
    y = np_from_torch_bf16(torch.nn.functional.linear(x.to(DEV), wt))
    xq, xs = F.quantize_tensorwise(x.float().numpy())
    wq, ws = F.quantize_tensorwise(w.float().numpy())
    rows = np.arange(m) if m <= 64 else np.random.default_rng(0).choice(m, 48, replace=False)
    y_ref = F.scaled_mm(xq[rows], wq, np.full(len(rows), xs, np.float32), np.full(n, ws, np.float32))
    assert _rel(y[rows], y_ref) <= 1e-3


def test_per_tensor_slices_keep_the_scalar_scale():
    w = _randn_bf16((64, 256), 3, 0.05).to(DEV)
    for cls in (Int8Tensor, Float8Tensor):
        t = cls.from_hp(w, granularity=PerTensor())
        top, left = t[:32], t[:, :128]
        assert top.scale.numel() == 1 and left.scale.numel() == 1 and top.block_size == [32, 256] and left.block_size == [64, 128]
        assert torch.equal(top.qdata.view(torch.uint8), t.qdata[:32].view(torch.uint8))


def test_fp8_activation_value_bounds_through_the_config(gv):
    """Float8DynamicActivationFloat8WeightConfig(granularity=PerRow(), activation_value_lb, activation_value_ub): the activation codes
    the linear multiplies are the reference's (fixture), and the output is the oracle's scaled mm on them."""
    lb, ub = (float(v) for v in gv["fp8clamp_bounds"])
    w = _t(gv, "w")
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=True, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(w)
        lin.bias.copy_(_t(gv, "bias"))
    quantize_(lin, Float8DynamicActivationFloat8WeightConfig(granularity=PerRow(), activation_value_lb=lb, activation_value_ub=ub))
    assert lin.weight.act_quant_kwargs.hp_value_lb == lb and lin.weight.act_quant_kwargs.hp_value_ub == ub
    x = _t(gv, "x")
    y = np_from_torch_bf16(lin(x))
    # the cast the handler runs, on its own: bit-exact against the reference fixture
    amax = ops.rowwise_amax(x).clamp(min=lb, max=ub).to(torch.bfloat16).float()
    xq, xs = ops.fp8_quantize_rowwise_amax(x, amax)
    assert np.array_equal(xq.view(torch.uint8).cpu().numpy(), gv["fp8clamp_xq"]) and np.array_equal(xs.flatten().cpu().numpy(), gv["fp8clamp_xs"])
    wq, ws = F.quantize_rowwise(bf16_bits_to_f32(gv["w"]))
    y_ref = F.scaled_mm(gv["fp8clamp_xq"], wq, gv["fp8clamp_xs"], ws, bf16_bits_to_f32(gv["bias"]))
    assert _rel(y, y_ref) <= 1e-3


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_int8_static_activation_golden_through_the_config(gv, kind):
    """Int8StaticActivationInt8WeightConfig: calibrated activation scale (and zero-point); bit-exact against the reference's F.linear."""
    scale = torch.from_numpy(gv[f"static_{kind}_scale"]).to(DEV)
    zp = torch.from_numpy(gv["static_asym_zp"]).to(DEV) if kind == "asym" else None
    x = _t(gv, "x")
    assert np.array_equal(ops.int8_quantize_static(x, scale, zp).cpu().numpy(), gv[f"static_{kind}_xq"])
    lin = torch.nn.Linear(gv["w"].shape[1], gv["w"].shape[0], bias=True, device=DEV, dtype=torch.bfloat16)
    with torch.no_grad():
        lin.weight.copy_(_t(gv, "w"))
        lin.bias.copy_(_t(gv, "bias"))
    quantize_(lin, Int8StaticActivationInt8WeightConfig(act_quant_scale=scale, act_quant_zero_point=zp, granularity=[PerTensor(), PerRow()],
                                                         act_mapping_type=MappingType.ASYMMETRIC if kind == "asym" else MappingType.SYMMETRIC))
    assert isinstance(lin.weight, Int8Tensor) and lin.weight.act_quant_scale is not None
    assert np.array_equal(_bits(lin(x)), gv[f"static_{kind}_y"])
    # per-row static scales (one per activation row) take the same kernel with stride 1
    rs = torch.linspace(0.03, 0.09, x.shape[0], device=DEV)
    q = ops.int8_quantize_static(x, rs).cpu().numpy()
    ref = np.clip(np.rint(bf16_bits_to_f32(gv["x"]) * (np.float32(1.0) / rs.cpu().numpy())[:, None]), -128, 127).astype(np.int8)
    assert np.array_equal(q, ref)
