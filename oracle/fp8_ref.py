"""numpy restatement of torchao's float8 (OCP e4m3fn) rowwise dynamic linear.
TEST INFRASTRUCTURE ONLY.  Paths relative to /root/reference.  Pinned against
tests/golden/int8_fp8.npz (reference Python; the e4m3 cast table comes from
torch.float8_e4m3fn itself).
"""
import numpy as np

from . import bf16

E4M3_MAX = np.float32(448.0)


def _e4m3_table():
    """All 256 e4m3fn codes decoded to fp32 (0x7f / 0xff are NaN)."""
    codes = np.arange(256, dtype=np.uint32)
    sign = np.where(codes & 0x80, -1.0, 1.0).astype(np.float32)
    e = ((codes >> 3) & 0xF).astype(np.int32)
    m = (codes & 0x7).astype(np.float32)
    val = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * np.exp2((e - 7).astype(np.float32)))
    val = (sign * val).astype(np.float32)
    val[(codes & 0x7F) == 0x7F] = np.nan
    return val


E4M3 = _e4m3_table()
_POS = E4M3[:127]  # 0 .. 448, strictly increasing


def f32_to_e4m3(x):
    """fp32 -> e4m3fn code, round-to-nearest-even, inputs already clamped to
    +-448 (what .to(torch.float8_e4m3fn) does for in-range values); NaN -> 0x7f."""
    x = np.asarray(x, dtype=np.float32)
    a = np.abs(x).astype(np.float64)
    idx = np.searchsorted(_POS.astype(np.float64), a, side="left")  # first code >= a
    idx = np.clip(idx, 0, 126)
    lo = np.clip(idx - 1, 0, 126)
    d_hi = np.abs(_POS[idx].astype(np.float64) - a)
    d_lo = np.abs(a - _POS[lo].astype(np.float64))
    pick_lo = (d_lo < d_hi) | ((d_lo == d_hi) & (lo % 2 == 0))
    code = np.where(pick_lo, lo, idx).astype(np.uint8)
    code = np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)
    code = np.where(np.isnan(x), np.uint8(0x7F), code)
    return code


def e4m3_to_f32(code):
    return E4M3[np.asarray(code, dtype=np.uint8)]


def quantize_rowwise(x, amax=None):
    """Float8Tensor.from_hp(x, e4m3, PerRow) (`amax`: see int8_ref.quantize_rowwise):
    torchao/quantization/quantize_/workflows/float8/float8_tensor.py:167-253 ->
    _choose_scale_float8 (quant_primitives.py:2192-2212) and
    _quantize_affine_float8 (:2271-2287), input bf16:
        scale = f32( bf16( amax_row / 448 ) )           (no eps clamp)
        q     = e4m3_rne( clamp( f32(x) / scale, -448, 448 ) )
    Returns (codes uint8 [M,K], scale fp32 [M])."""
    x = np.asarray(x, dtype=np.float32)
    assert bf16.is_bf16(x)
    if amax is None:
        amax = np.abs(x).max(axis=1)
    scale = bf16.div(np.asarray(amax, dtype=np.float32), E4M3_MAX).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (x / scale[:, None]).astype(np.float32)
    t = np.clip(t, -E4M3_MAX, E4M3_MAX)  # NaN stays NaN
    return f32_to_e4m3(t), scale


def scaled_mm(a, b, scale_a, scale_b, bias=None):
    """aten::_scaled_mm with rowwise scales as called at
    torchao/float8/inference.py:104-123 (out bf16):
        y = bf16( (sum_k a[m,k] b[n,k])_f32 * scale_a[m] * scale_b[n] + bias[n] )
    a, b are e4m3 codes ([M,K], [N,K]).  The sum is taken in float64."""
    af = e4m3_to_f32(a).astype(np.float64)
    bf = e4m3_to_f32(b).astype(np.float64)
    y = (af @ bf.T) * np.asarray(scale_a, np.float64)[:, None] * np.asarray(scale_b, np.float64)[None, :]
    if bias is not None:
        y = y + np.asarray(bias, np.float64)[None, :]
    return bf16.bf16_round(y.astype(np.float32))


def linear(x, w, bias=None):
    xq, xs = quantize_rowwise(x)
    wq, ws = quantize_rowwise(w)
    return scaled_mm(xq, wq, xs, ws, bias)


def grouped_mm(a, b, scale_a, scale_b, offs):
    """Float8Tensor's aten::_grouped_mm, rowwise (float8_tensor.py:1085-1122 -> scaled_grouped_mm with RowWise recipes):
    out[offs[e-1]:offs[e]] = bf16((a_rows @ b[e]^T) * scale_a[m] * scale_b[e][n]).  a codes [M, K]; b codes [E, N, K];
    scale_a [M]; scale_b [E, N]; rows past offs[-1] stay zero."""
    out = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    start = 0
    for e in range(b.shape[0]):
        end = int(offs[e])
        if end > start:
            out[start:end] = scaled_mm(a[start:end], b[e], np.asarray(scale_a)[start:end], np.asarray(scale_b)[e])
        start = end
    return out


def quantize_tensorwise(x):
    """Float8Tensor.from_hp(x, e4m3, PerTensor()): one amax over the whole tensor, scale = f32(bf16(amax / 448))
    (quant_primitives.py:2192-2212 with block_size = shape).  Returns (codes uint8 [M,K], scale fp32 scalar)."""
    x = np.asarray(x, dtype=np.float32)
    amax = np.float32(np.abs(x).max())
    q, s = quantize_rowwise(x, amax=np.full((x.shape[0],), amax, dtype=np.float32))
    return q, s[0]


def clamp_amax(amax, lb=None, ub=None):
    """hp_value_lb / hp_value_ub of _choose_scale_float8 (quant_primitives.py:2203-2204): torch.clamp of the bf16 amax with Python
    floats -- compared in fp32, the result rounded back to bf16."""
    a = np.asarray(amax, dtype=np.float32)
    if lb is not None:
        a = np.maximum(a, np.float32(lb))
    if ub is not None:
        a = np.minimum(a, np.float32(ub))
    return bf16.bf16_round(a.astype(np.float32))
