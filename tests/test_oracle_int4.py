"""CPU: the numpy oracle of the int4 tinygemm path against the fixtures generated
from the reference's own Python (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bf16_bits_to_f32
from oracle import bf16, int4_ref as R

CASES = ["g32", "g64", "g128", "g256"]


@pytest.mark.parametrize("case", CASES)
def test_qparams_quantize_dequant_bit_exact(golden_int4, case):
    d = golden_int4
    g = int(d[f"{case}_group"])
    w = bf16_bits_to_f32(d[f"{case}_w"])
    s, z = R.choose_qparams_tinygemm(w, g)
    assert np.array_equal(bf16.to_bits(s), d[f"{case}_scale"])
    assert np.array_equal(bf16.to_bits(z), d[f"{case}_zero"])
    q = R.quantize_tinygemm(w, s, z, g)
    assert np.array_equal(q, d[f"{case}_q"].astype(np.int32))
    assert np.array_equal(R.nibble_pack(q), d[f"{case}_byte"])
    sz = R.pack_scales_and_zeros(s, z)
    assert np.array_equal(bf16.to_bits(sz), d[f"{case}_sz"])
    dq = R.dequantize_tinygemm(q, sz, g)
    assert np.array_equal(bf16.to_bits(dq), d[f"{case}_dq"])


@pytest.mark.parametrize("case", CASES)
def test_mm_matches_reference_dequant_matmul(golden_int4, case):
    """oracle mm vs the reference's F.linear(x, dequant) in bf16 (CPU BLAS, fp32
    accumulation in its own order): equal up to one final bf16 rounding."""
    d = golden_int4
    g = int(d[f"{case}_group"])
    x = bf16_bits_to_f32(d[f"{case}_x"])
    qdata = R.convert_weight_to_int4pack(d[f"{case}_byte"])
    sz = bf16_bits_to_f32(d[f"{case}_sz"])
    y = R.weight_int4pack_mm(x, qdata, g, sz)
    y_ref = bf16_bits_to_f32(d[f"{case}_y"])
    rel = np.linalg.norm(y - y_ref) / np.linalg.norm(y_ref)
    assert rel < 1e-3, rel
    # elementwise: at most 1 bf16 ulp apart
    ulp = np.abs(y_ref) * 2.0 ** -7 + 1e-30
    assert np.all(np.abs(y - y_ref) <= ulp)


def test_pack_roundtrip_and_shape_pins():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, size=(2048, 2048), dtype=np.int32)
    packed = R.convert_weight_to_int4pack(R.nibble_pack(q))
    # shape pin of the reference (test_int4_tile_packed_to_4d_tensor.py:113-149)
    assert packed.shape == (256, 16, 32, 4) and packed.dtype == np.int32
    assert np.array_equal(R.unpack_int4pack(packed), q)


def test_pack_one_hot_positions():
    """The published ROCm tile order, probed the way SURVEY 8c prescribes (one-hot)."""
    n, k = 32, 256
    for (nn, kk) in [(0, 0), (5, 1), (17, 130), (31, 255), (16, 77)]:
        q = np.zeros((n, k), dtype=np.int32)
        q[nn, kk] = 15
        words = R.convert_weight_to_int4pack(R.nibble_pack(q)).view(np.uint32).reshape(n // 16, k // 128, 64, 4)
        nz = np.argwhere(words != 0)
        assert len(nz) == 1
        nt, ks, t, j = nz[0]
        assert nt == nn // 16 and ks == kk // 128
        assert t % 16 == nn % 16
        kin = kk % 128
        tile, off = kin // 16, kin % 16
        assert t // 16 == off // 4 and j == tile // 2
        v = (tile % 2) * 4 + off % 4          # v0..v7 within the word
        slot = [0, 4, 1, 5, 2, 6, 3, 7][v]    # v0|v2<<4|v4<<8|v6<<12|v1<<16|v3<<20|v5<<24|v7<<28
        assert words[nt, ks, t, j] == np.uint32(15) << np.uint32(4 * slot)


@pytest.mark.skipif(not os.path.exists(os.path.join(GOLDEN, "int4pack_gfx950.npz")), reason="GPU-box fixture not generated yet")
def test_pack_pinned_against_torch_core_on_gfx950():
    """Fixture produced by torch.ops.aten._convert_weight_to_int4pack on the
    MI355X box (scripts/gpu_probe_torch_core.py): the authority for the layout."""
    d = np.load(os.path.join(GOLDEN, "int4pack_gfx950.npz"))
    assert np.array_equal(R.convert_weight_to_int4pack(d["byte_w"]), d["packed"])
