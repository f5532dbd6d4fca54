"""CPU: int8 / fp8 / MXFP8 numpy oracles against fixtures generated from the
reference's own Python (tests/golden/make_golden.py), including the reference's
MXFP8 bitwise contract (torchao/testing/_mxfp8_test_utils.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bf16_bits_to_f32
from oracle import bf16, fp8_ref as F, int8_ref as I, mx_ref as MX


@pytest.fixture(scope="module")
def g8():
    return np.load(os.path.join(GOLDEN, "int8_fp8.npz"))


@pytest.fixture(scope="module")
def gmx():
    return np.load(os.path.join(GOLDEN, "mx.npz"))


def test_int8_quantize_bit_exact(g8):
    for t in ("x", "w"):
        q, s = I.quantize_rowwise(bf16_bits_to_f32(g8[t]))
        assert np.array_equal(q, g8[f"int8_{t}q"])
        assert np.array_equal(s, g8[f"int8_{t}s"])


def test_int8_linear_epilogue_bit_exact(g8):
    c = I.int_mm(g8["int8_xq"], g8["int8_wq"])
    assert np.array_equal(c, g8["int8_c"])
    y = I.scaled_mm(g8["int8_xq"], g8["int8_xs"], g8["int8_wq"], g8["int8_ws"], bf16_bits_to_f32(g8["bias"]))
    assert np.array_equal(bf16.to_bits(y), g8["int8_y"])


def test_fp8_quantize_bit_exact(g8):
    q, s = F.quantize_rowwise(bf16_bits_to_f32(g8["fp8_x"]))
    assert np.array_equal(q, g8["fp8_xq"]) and np.array_equal(s, g8["fp8_xs"])
    q, s = F.quantize_rowwise(bf16_bits_to_f32(g8["w"]))
    assert np.array_equal(q, g8["fp8_wq"]) and np.array_equal(s, g8["fp8_ws"])
    # all-zero rows: scale 0 -> 0/0 = NaN codes (reference behaviour)
    q, s = F.quantize_rowwise(np.zeros((2, 256), np.float32))
    assert np.array_equal(s, g8["fp8_zero_s"])
    # the NaN sign bit is not part of the contract (x86 0/0 gives 0xff, the oracle emits 0x7f)
    assert np.all((q & 0x7F) == 0x7F) and np.all((g8["fp8_zero_q"] & 0x7F) == 0x7F)


def test_fp8_scaled_mm_matches_reference_dequant(g8):
    y = F.scaled_mm(g8["fp8_xq"], g8["fp8_wq"], g8["fp8_xs"], g8["fp8_ws"], bf16_bits_to_f32(g8["bias"]))
    ref = g8["fp8_y_dequant_f32"]
    assert np.linalg.norm(y - ref) / np.linalg.norm(ref) < 3e-3  # bf16 output rounding only


def test_e8m0_rceil_boundaries(gmx):
    v = gmx["e8m0_rceil_in"].view(np.float32)
    assert np.array_equal(MX.f32_to_e8m0_rceil(v), gmx["e8m0_rceil_out"])


@pytest.mark.parametrize("mode,name", [(MX.RCEIL, "rceil"), (MX.FLOOR, "floor")])
def test_mxfp8_semantic_contract(gmx, mode, name):
    x = bf16_bits_to_f32(gmx[f"sem_{name}_x"])
    data, scale = MX.to_mx(x, mode)
    names = list(gmx[f"sem_{name}_names"])
    for i, nm in enumerate(names):
        assert scale[i, 0] == gmx[f"sem_{name}_scale"][i], nm
        exp = gmx[f"sem_{name}_data"][i]
        got = data[i]
        # NaN payloads: any NaN code matches any NaN code (0x7f / 0xff)
        nan_e, nan_g = (exp & 0x7F) == 0x7F, (got & 0x7F) == 0x7F
        assert np.array_equal(nan_e, nan_g), nm
        assert np.array_equal(exp[~nan_e], got[~nan_g]), nm


@pytest.mark.parametrize("mode,name", [(MX.RCEIL, "rceil"), (MX.FLOOR, "floor")])
def test_to_mx_seeded_bit_exact(gmx, mode, name):
    data, scale = MX.to_mx(bf16_bits_to_f32(gmx["x"]), mode)
    assert np.array_equal(scale, gmx[f"{name}_scale"])
    exp = gmx[f"{name}_data"]
    nan = (exp & 0x7F) == 0x7F
    assert np.array_equal((data & 0x7F) == 0x7F, nan)
    assert np.array_equal(data[~nan], exp[~nan])


def test_grouped_mm_matches_reference_emulated(gmx):
    a_d, a_s = MX.to_mx(bf16_bits_to_f32(gmx["g_a"]), MX.RCEIL)
    w_d, w_s = MX.to_mx(bf16_bits_to_f32(gmx["g_w"]), MX.RCEIL)
    assert np.array_equal(a_d, gmx["g_a_data"]) and np.array_equal(a_s, gmx["g_a_scale"])
    assert np.array_equal(w_d, gmx["g_w_data"]) and np.array_equal(w_s, gmx["g_w_scale"])
    y = MX.grouped_mm(a_d, a_s, w_d, w_s, gmx["g_offs"])
    ref = bf16_bits_to_f32(gmx["g_y"])
    rows = int(gmx["g_offs"][-1])
    assert np.all(np.abs(y[:rows] - ref[:rows]) <= np.abs(ref[:rows]) * 2.0 ** -7 + 1e-6)
    assert np.linalg.norm(y[:rows] - ref[:rows]) / np.linalg.norm(ref[:rows]) < 1e-3
