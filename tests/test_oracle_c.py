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


def _bf16_rand(rng, shape, scale=1.0):
    return bf16.bf16_round((rng.standard_normal(shape) * scale).astype(np.float32))


def test_c_int8_dynamic_linear_bit_exact():
    from oracle import int8_ref

    rng = np.random.default_rng(3)
    x, w = _bf16_rand(rng, (7, 256)), _bf16_rand(rng, (48, 256), 0.02)
    x[2] = 0.0  # all-zero row: scale clamps at eps
    wq, ws = int8_ref.quantize_rowwise(w)
    got = c_ref.int8_dynamic_linear(bf16.to_bits(x), wq, ws)
    xq, xs = int8_ref.quantize_rowwise(x)
    want = bf16.to_bits(int8_ref.scaled_mm(xq, xs, wq, ws))
    assert np.array_equal(got, want)


def test_c_fp8_rowwise_linear_matches_numpy_oracle():
    from oracle import fp8_ref

    rng = np.random.default_rng(4)
    x, w = _bf16_rand(rng, (5, 384)), _bf16_rand(rng, (32, 384), 0.02)
    wq, ws = fp8_ref.quantize_rowwise(w)
    got = bf16.from_bits(c_ref.fp8_rowwise_linear(bf16.to_bits(x), wq, ws))
    xq, xs = fp8_ref.quantize_rowwise(x)
    want = fp8_ref.scaled_mm(xq, wq, xs, ws)
    # same codes, float64 sum on both sides: only the last bf16 rounding can differ (by the fp32 product order)
    assert np.mean(got == want) > 0.99
    assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -7 + 1e-30)


def test_c_mxfp8_grouped_mm_matches_numpy_oracle():
    from oracle import mx_ref

    rng = np.random.default_rng(5)
    E, N, K = 3, 16, 128
    a = _bf16_rand(rng, (64, K))
    w = _bf16_rand(rng, (E, N, K), 0.02)
    wq, wsc = mx_ref.to_mx(w, mx_ref.RCEIL)
    offs = np.array([32, 32, 64], dtype=np.int32)  # empty middle group
    got = bf16.from_bits(c_ref.mxfp8_grouped_mm(bf16.to_bits(a), wq, wsc, offs))
    aq, asc = mx_ref.to_mx(a, mx_ref.RCEIL)
    want, mag = mx_ref.grouped_mm(aq, asc, wq, wsc, offs, return_abs=True)
    assert np.all(np.abs(got - want) <= np.abs(want) * 2.0 ** -7 + mag * 2.0 ** -16)
    assert np.linalg.norm(got - want) / np.linalg.norm(want) < 1e-3
