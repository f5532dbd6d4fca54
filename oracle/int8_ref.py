"""numpy restatement of torchao's int8 dynamic-activation x int8-weight linear.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Paths relative to
/root/reference.  Pinned bit-exactly against tests/golden/int8_fp8.npz, generated
by tests/golden/make_golden.py from the reference's own Python.
"""
import numpy as np

from . import bf16

F32_EPS = np.float32(np.finfo(np.float32).eps)  # eps passed at int8_tensor.py:210


def quantize_rowwise(x, amax=None):
    """Int8Tensor.from_hp(x, PerRow()) with the defaults (SYMMETRIC, int8).
    `amax` (optional, [M]): the rows' amax when x is only a column shard of the tensor the reference would quantize
    (K-sharded TP linears); default: reduced over x's own columns.

    torchao/quantization/quantize_/workflows/int8/int8_tensor.py:191-230 ->
    choose_qparams_affine (quant_primitives.py:1534-1583) and quantize_affine
    (:463-485), input bf16:
        amax  = max(-min(row, 0), max(row, 0))                       (bf16, exact)
        scale = f32( max( bf16(amax / 127.5), bf16(f32_eps) ) )      (division in bf16)
        q     = clamp(round_half_even( f32(x) * f32(1 / scale) ) + 0, -128, 127)
    Returns (q int8 [M,K], scale fp32 [M]).
    """
    x = np.asarray(x, dtype=np.float32)
    assert bf16.is_bf16(x)
    if amax is None:
        mn = np.minimum(x.min(axis=1), 0)
        mx = np.maximum(x.max(axis=1), 0)
        amax = np.maximum(-mn, mx)
    amax = np.asarray(amax, dtype=np.float32)
    scale = bf16.div(amax, np.float32(127.5))
    scale = np.maximum(scale, bf16.bf16_round(F32_EPS)).astype(np.float32)
    inv = (np.float32(1.0) / scale).astype(np.float32)
    q = np.rint(x * inv[:, None]).astype(np.float32)  # fp32 product, half-to-even
    q = np.clip(q, -128, 127).astype(np.int8)
    return q, scale


def dequantize(q, scale):
    return q.astype(np.float32) * scale[:, None]


def int_mm(a, b_t):
    """aten::_int_mm: int32 [M,N] = a[M,K] @ b_t[N,K]^T (exact)."""
    return a.astype(np.int32) @ b_t.astype(np.int32).T


def scaled_mm(xq, x_scale, wq, w_scale, bias=None):
    """The Int8Tensor linear epilogue on the GPU path.

    int8_tensor.py:305-359 + int8/kernels.py:143-144, activation dtype bf16:
        t = bf16( f32(c_int32) * x_scale[m] )      (_int_scaled_matmul(...).to(bf16))
        y = f32(t) * w_scale[n]                    (bf16 tensor * fp32 tensor -> fp32)
        y = y + bias                               (fp32 += bf16)
        return bf16(y)
    """
    c = int_mm(xq, wq).astype(np.float32)  # int32 -> fp32 (RNE above 2^24, like torch)
    t = bf16.bf16_round(c * np.asarray(x_scale, np.float32)[:, None])
    y = t * np.asarray(w_scale, np.float32)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, np.float32)[None, :]
    return bf16.bf16_round(y.astype(np.float32))


def linear(x, w, bias=None):
    """Full dynamic-quant linear from bf16 x [M,K] and bf16 w [N,K]."""
    xq, xs = quantize_rowwise(x)
    wq, ws = quantize_rowwise(w)
    return scaled_mm(xq, xs, wq, ws, bias)


def quantize_tensorwise(x):
    """Int8Tensor.from_hp(x, PerTensor()): the same arithmetic with ONE amax over the whole tensor (block_size = shape;
    int8_tensor.py:191-230).  Returns (q int8 [M,K], scale fp32 scalar)."""
    x = np.asarray(x, dtype=np.float32)
    amax = np.float32(max(-min(x.min(), 0.0), max(x.max(), 0.0)))
    q, s = quantize_rowwise(x, amax=np.full((x.shape[0],), amax, dtype=np.float32))
    return q, s[0]


def quantize_rowwise_asym(x):
    """Int8Tensor.from_hp(x, PerRow(), mapping_type=ASYMMETRIC) (the activation side of
    Int8DynamicActivationInt8WeightConfig(act_mapping_type=ASYMMETRIC)).

    choose_qparams_affine ASYMMETRIC branch (quant_primitives.py:1568-1574), input bf16 so every tensor op rounds to bf16:
        mn = min(min(row), 0); mx = max(max(row), 0)
        scale = max( bf16( bf16(mx - mn) / 255 ), bf16(f32_eps) )
        zp    = clamp( -128 - rint( bf16(mn / scale) ), -128, 127 )                (int8)
    quantize_affine (:463-485) with scale widened to fp32:
        q = clamp( rint( f32(x) * f32(1 / scale) ) + zp, -128, 127 )
    Returns (q int8 [M,K], scale fp32 [M], zero_point int8 [M])."""
    x = np.asarray(x, dtype=np.float32)
    assert bf16.is_bf16(x)
    mn = np.minimum(x.min(axis=1), 0).astype(np.float32)
    mx = np.maximum(x.max(axis=1), 0).astype(np.float32)
    scale = bf16.div(bf16.bf16_round(mx - mn), np.float32(255.0))
    scale = np.maximum(scale, bf16.bf16_round(F32_EPS)).astype(np.float32)
    r = np.rint(bf16.div(mn, scale)).astype(np.float32)
    zp = np.clip(np.float32(-128.0) - r, -128, 127).astype(np.float32)
    inv = (np.float32(1.0) / scale).astype(np.float32)
    q = np.rint(x * inv[:, None]).astype(np.float32) + zp[:, None]
    q = np.clip(q, -128, 127).astype(np.int8)
    return q, scale, zp.astype(np.int8)


def scaled_mm_asym(xq, x_scale, x_zp, wq, w_scale, bias=None):
    """The Int8Tensor linear with an asymmetric activation (int8_tensor.py:305-346):
        t    = bf16( f32(c_int32) * x_scale[m] )
        corr = bf16( (f32(zp[m]) * x_scale[m]) * f32(rowsum(wq)[n]) )
        t    = bf16( t - corr )
        y    = bf16( f32(t) * w_scale[n] (+ bias) )"""
    c = int_mm(xq, wq).astype(np.float32)
    xs = np.asarray(x_scale, np.float32)[:, None]
    t = bf16.bf16_round(c * xs)
    wsum = wq.astype(np.int64).sum(axis=1).astype(np.float32)[None, :]
    corr = bf16.bf16_round((np.asarray(x_zp, np.float32)[:, None] * xs).astype(np.float32) * wsum)
    t = bf16.bf16_round(t - corr)
    y = t * np.asarray(w_scale, np.float32)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, np.float32)[None, :]
    return bf16.bf16_round(y.astype(np.float32))


def linear_asym(x, w, bias=None):
    xq, xs, zp = quantize_rowwise_asym(x)
    wq, ws = quantize_rowwise(w)
    return scaled_mm_asym(xq, xs, zp, wq, ws, bias)
