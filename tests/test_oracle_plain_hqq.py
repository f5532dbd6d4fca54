"""CPU: oracle/int4_plain_ref.py and oracle/hqq_ref.py against fixtures generated from the reference's own code
(tests/golden/make_golden.py: make_int4_plain, make_hqq)."""
import os

import numpy as np

from conftest import GOLDEN, bf16_bits_to_f32
from oracle import bf16, hqq_ref as H, int4_plain_ref as P, int4_ref as R


def test_plain_quantize_zp_matches_reference_restatements():
    g = np.load(os.path.join(GOLDEN, "int4_plain.npz"))
    for name, gs in (("g32", 32), ("g128", 128), ("g256", 256)):
        w = bf16_bits_to_f32(g[f"{name}_w"])
        q, s, z = P.quantize_zp(w, gs)
        assert np.array_equal(s, g[f"{name}_scale_f32"]) and np.array_equal(z, g[f"{name}_zero_f32"])
        assert np.array_equal(q, g[f"{name}_q"])  # _int4_row_quantize_zp_precomputed_qparams
        # _int4_row_dequantize_zp: (q + 8) * scale + (zero - 8 scale) in fp32
        sr, zr = np.repeat(s.T, gs, axis=1), np.repeat(z.T, gs, axis=1)
        dq = ((q.astype(np.float32) + 8) * sr + (zr - sr * 8)).astype(np.float32)
        assert np.array_equal(dq, g[f"{name}_dq_f32"])
        # Int4WeightFakeQuantizer._bf16_activations_forward: bf16(q * scale + zero) in fp32
        fq = bf16.bf16_round((q.astype(np.float32) * sr + zr).astype(np.float32))
        assert np.array_equal(fq, bf16_bits_to_f32(g[f"{name}_fq_zp"]))


def test_plain_pack_roundtrip_and_tilepacked_equivalence():
    """pack_int4 / unpack; and the PLAIN (q, scale, zero) re-laid as tinygemm codes dequantises to the same bf16 matrix."""
    rng = np.random.default_rng(0)
    w = bf16.bf16_round((rng.standard_normal((32, 256)) * 0.05).astype(np.float32))
    qdata, s, z = P.from_hp(w, 128)
    q = P.unpack_int4(qdata)
    assert q.min() >= -8 and q.max() <= 7 and np.array_equal(P.pack_int4(q), qdata)
    want = P.dequantize(qdata, s, z, 128)
    # tinygemm convention: unsigned code q + 8, scale_and_zero [K/g, N, 2]
    sz = np.stack([s, z], axis=-1)
    got = R.dequantize_tinygemm((q.astype(np.int32) + 8), sz, 128)
    assert np.array_equal(got, want)
    # symmetric flavour: zero_point is exactly zero, codes in [-8, 7]
    qd2, s2, z2 = P.from_hp(w, 128, symmetric=True)
    assert not z2.any() and np.all(s2 > 0)


def test_hqq_matches_reference_run_in_float16():
    g = np.load(os.path.join(GOLDEN, "hqq.npz"))
    for name, gs in (("g64", 64), ("g128", 128)):
        w = bf16_bits_to_f32(g[f"{name}_w"])
        q, s, z = H.choose_qparams_and_quantize_hqq(w, gs)
        assert np.array_equal(q, g[f"{name}_q"])
        assert np.array_equal(s, bf16_bits_to_f32(g[f"{name}_scale"])) and np.array_equal(z, bf16_bits_to_f32(g[f"{name}_zero"]))
