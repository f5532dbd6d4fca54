"""CPU: the PerTensor / ASYMMETRIC variants of the int8 and fp8 oracles against fixtures produced by the reference's own
Int8Tensor / Float8Tensor from_hp and F.linear dispatch (tests/golden/make_golden.py:make_int8_fp8_variants)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bf16_bits_to_f32
from oracle import bf16, fp8_ref as F, int8_ref as I


@pytest.fixture(scope="module")
def gv():
    return np.load(os.path.join(GOLDEN, "int8_fp8_variants.npz"))


def test_int8_asymmetric_quantize_bit_exact(gv):
    q, s, zp = I.quantize_rowwise_asym(bf16_bits_to_f32(gv["x"]))
    assert np.array_equal(q, gv["asym_xq"]) and np.array_equal(s, gv["asym_xs"]) and np.array_equal(zp, gv["asym_xzp"])
    # the fixture covers the corner rows: all-positive (zp -128), all-negative (zp 127), all-zero (scale = eps)
    assert gv["asym_xzp"][1] == -128 and gv["asym_xzp"][2] >= 126 and gv["asym_xzp"][5] == -128


@pytest.mark.parametrize("with_bias", [True, False])
def test_int8_asymmetric_linear_bit_exact(gv, with_bias):
    b = bf16_bits_to_f32(gv["bias"]) if with_bias else None
    y = I.scaled_mm_asym(gv["asym_xq"], gv["asym_xs"], gv["asym_xzp"], gv["asym_wq"], gv["asym_ws"], b)
    assert np.array_equal(bf16.to_bits(y), gv["asym_y" if with_bias else "asym_y_nobias"])
    y2 = I.linear_asym(bf16_bits_to_f32(gv["x"]), bf16_bits_to_f32(gv["w"]), b)
    assert np.array_equal(bf16.to_bits(y2), gv["asym_y" if with_bias else "asym_y_nobias"])


def test_int8_per_tensor_bit_exact(gv):
    x, w = bf16_bits_to_f32(gv["x"]), bf16_bits_to_f32(gv["w"])
    xq, xs = I.quantize_tensorwise(x)
    wq, ws = I.quantize_tensorwise(w)
    assert np.array_equal(xq, gv["pt_xq"]) and xs == gv["pt_xs"][0]
    assert np.array_equal(wq, gv["pt_wq"]) and ws == gv["pt_ws"][0]
    y = I.scaled_mm(xq, np.full(x.shape[0], xs, np.float32), wq, np.full(w.shape[0], ws, np.float32), bf16_bits_to_f32(gv["bias"]))
    assert np.array_equal(bf16.to_bits(y), gv["pt_y"])


def test_fp8_per_tensor_bit_exact(gv):
    for t in ("x", "w"):
        q, s = F.quantize_tensorwise(bf16_bits_to_f32(gv[t]))
        assert np.array_equal(q, gv[f"fp8pt_{t}q"]) and s == gv[f"fp8pt_{t}s"][0]
    x, w = bf16_bits_to_f32(gv["x"]), bf16_bits_to_f32(gv["w"])
    xq, xs = F.quantize_tensorwise(x)
    wq, ws = F.quantize_tensorwise(w)
    y = F.scaled_mm(xq, wq, np.full(x.shape[0], xs, np.float32), np.full(w.shape[0], ws, np.float32), bf16_bits_to_f32(gv["bias"]))
    ref = gv["fp8pt_y_dequant_f32"]
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 3e-3  # bf16 output rounding only


def test_fp8_cast_residual_corrected_reciprocal_gives_the_division_codes():
    """The fp8 cast kernels (csrc/quant_math.h fp8_quant8) replace x / s by q0 = x r, q = fma(fma(-q0, s, x), r, q0) with r = 1 / s.
    Every finite bf16 x against bf16-valued scales over 2^-100 .. 2^100: the e4m3 codes are those of the division."""
    old = np.seterr(all="ignore")
    try:
        xs = (np.arange(0x0080, 0x7F80, dtype=np.uint32) << 16).view(np.float32)
        xs = np.concatenate([xs, -xs[::5], np.float32([0.0, -0.0, np.inf, -np.inf])])
        rng = np.random.default_rng(1)
        sc = (rng.integers(0x0D80, 0x7180, size=160, dtype=np.uint32) << 16).view(np.float32)
        sc = np.concatenate([sc, np.float32([1.0, 448.0, 2.0 ** -7, 3.0])])
        for s in sc:
            want = F.f32_to_e4m3(np.clip((xs / s).astype(np.float32), -448, 448))
            r = (np.float32(1.0) / s).astype(np.float32)
            q0 = (xs * r).astype(np.float32)
            e = (xs.astype(np.float64) - q0.astype(np.float64) * np.float64(s)).astype(np.float32)   # fma: one rounding
            q = (q0.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(np.float32)
            q = np.where(np.isfinite(q0) & (q0 != 0), q, q0)
            got = F.f32_to_e4m3(np.clip(q, -448, 448))
            assert np.array_equal(got, want), float(s)
    finally:
        np.seterr(**old)


def test_fp8_activation_value_bounds_bit_exact(gv):
    """hp_value_lb / hp_value_ub (Float8DynamicActivationFloat8WeightConfig(activation_value_lb / _ub)): the clamped amax, then the
    usual per-row cast -- against Float8Tensor.from_hp(..., hp_value_lb, hp_value_ub) of the reference."""
    x = bf16_bits_to_f32(gv["x"])
    lb, ub = (float(v) for v in gv["fp8clamp_bounds"])
    amax = F.clamp_amax(np.abs(x).max(axis=1), lb, ub)
    q, s = F.quantize_rowwise(x, amax=amax)
    assert np.array_equal(q, gv["fp8clamp_xq"]) and np.array_equal(s, gv["fp8clamp_xs"])
    assert s[5] == bf16.div(bf16.bf16_round(np.float32(lb)), np.float32(448.0))  # the all-zero row takes the (bf16-rounded) lower bound


def _quantize_static(x, scale, zp=0):
    """Int8Tensor.from_hp(x, ..., scale=, zero_point=): quantize_affine with given qparams (quant_primitives.py:463-485)."""
    inv = (np.float32(1.0) / np.float32(scale)).astype(np.float32)
    return np.clip(np.rint(x * inv).astype(np.float32) + np.float32(zp), -128, 127).astype(np.int8)


@pytest.mark.parametrize("kind", ["sym", "asym"])
def test_int8_static_activation_bit_exact(gv, kind):
    x, w, b = bf16_bits_to_f32(gv["x"]), bf16_bits_to_f32(gv["w"]), bf16_bits_to_f32(gv["bias"])
    s = float(gv[f"static_{kind}_scale"].reshape(-1)[0])
    zp = int(gv["static_asym_zp"].reshape(-1)[0]) if kind == "asym" else 0
    q = _quantize_static(x, s, zp)
    assert np.array_equal(q, gv[f"static_{kind}_xq"])
    wq, ws = I.quantize_rowwise(w)
    xs = np.full(x.shape[0], s, np.float32)
    y = I.scaled_mm(q, xs, wq, ws, b) if kind == "sym" else I.scaled_mm_asym(q, xs, np.full(x.shape[0], zp, np.int8), wq, ws, b)
    assert np.array_equal(bf16.to_bits(y), gv[f"static_{kind}_y"])
