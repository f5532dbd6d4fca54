"""CPU: the C restatement (used for the timed cpu_baseline) equals the numpy oracle."""
import numpy as np

from oracle import bf16, c_ref, int4_ref as R


def _case(n, k, g, m, seed):
    rng = np.random.default_rng(seed)
    w = bf16.bf16_round((rng.standard_normal((n, k)) * 0.02).astype(np.float32))
    x = bf16.bf16_round(rng.standard_normal((m, k)).astype(np.float32))
    s, z = R.choose_qparams_tinygemm(w, g)
    q = R.quantize_tinygemm(w, s, z, g)
    return x, R.convert_weight_to_int4pack(R.nibble_pack(q)), R.pack_scales_and_zeros(s, z), q


def test_c_dequant_bit_exact():
    for g in (32, 64, 128, 256):
        x, qdata, sz, q = _case(32, 512, g, 1, g)
        got = c_ref.int4_dequantize(qdata, bf16.to_bits(sz), 32, 512, g)
        want = bf16.to_bits(R.dequantize_tinygemm(q, sz, g))
        assert np.array_equal(got, want)


def test_c_linear_matches_numpy_oracle():
    x, qdata, sz, _ = _case(48, 1024, 128, 5, 7)
    got = bf16.from_bits(c_ref.int4_linear(bf16.to_bits(x), qdata, bf16.to_bits(sz), 48, 1024, 128))
    want = R.weight_int4pack_mm(x, qdata, 128, sz)
    assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -7 + 1e-6)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-3
    assert c_ref.num_threads() >= 1
