"""numpy restatement of torchao's MXFP8 cast (to_mx) and emulated grouped GEMM.
TEST INFRASTRUCTURE ONLY.  Paths relative to /root/reference.  Pinned against
tests/golden/mx.npz: the reference's own golden vectors
(torchao/testing/_mxfp8_test_utils.py, test/prototype/mx_formats/test_mx_tensor.py)
re-emitted by tests/golden/make_golden.py, plus to_mx run on seeded tensors.
"""
import numpy as np

from . import bf16
from .fp8_ref import E4M3_MAX, e4m3_to_f32, f32_to_e4m3

FLOOR, RCEIL = 0, 1
BLOCK = 32


def f32_to_e8m0_rceil(v):
    """torchao/prototype/mx_formats/mx_tensor.py:111-129: round an fp32 value UP to
    a power of two and return its biased exponent byte (NaN/Inf -> 255)."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    bits = v.view(np.uint32)
    be = ((bits >> 23) & 0xFF).astype(np.int32)
    mant = (bits & 0x7FFFFF).astype(np.int64)
    up = np.where(be == 0, mant > 0x400000, mant != 0)
    e = be + up.astype(np.int32)
    return np.where(np.isfinite(v), e, 255).astype(np.uint8)


def e8m0_reciprocal_f32(e):
    """mx_tensor.py:132-158: fp32 value of the E8M0 byte 254 - e."""
    r = ((254 - e.astype(np.int32)) & 0xFF).astype(np.uint32)
    bits = (r << 23).astype(np.uint32)
    bits = np.where(r == 0, np.uint32(0x00400000), bits)
    bits = np.where(r == 255, np.uint32(0x7F800001), bits)
    return bits.astype(np.uint32).view(np.float32)


def to_mx(x, mode=RCEIL):
    """to_mx(x, float8_e4m3fn, 32, mode) for bf16 x [..., K] (mx_tensor.py:228-409).
    Returns (codes uint8 [..., K], scale e8m0 uint8 [..., K/32])."""
    x = np.asarray(x, dtype=np.float32)
    shp = x.shape
    xb = x.reshape(-1, BLOCK)
    amax = np.abs(xb).max(axis=1).astype(np.float32)  # NaN propagates like torch.amax
    if mode == RCEIL:
        descale = (amax * np.float32(1.0 / 448.0)).astype(np.float32)
        e = f32_to_e8m0_rceil(descale)
    else:
        bits = amax.view(np.uint32)
        ex = ((bits >> 23) & 0xFF).astype(np.int32) - 127 - 8
        e = (np.clip(ex, -127, 128) + 127).astype(np.uint8)
        e = np.where(np.isfinite(amax), e, 255).astype(np.uint8)
    r = e8m0_reciprocal_f32(e)
    with np.errstate(over="ignore", invalid="ignore"):
        d = (xb * r[:, None]).astype(np.float32)
    if mode == FLOOR:
        d = np.where(np.isnan(d), d, np.clip(d, -E4M3_MAX, E4M3_MAX))  # eager saturation, :361-373
    codes = f32_to_e4m3(d)
    return codes.reshape(shp), e.reshape(*shp[:-1], shp[-1] // BLOCK)


def mx_dequant_bf16(codes, scale):
    """to_dtype(..., bf16) (mx_tensor.py:436-471): bf16(fp8) * bf16(2^(e-127))."""
    v = e4m3_to_f32(codes)
    s = np.exp2(scale.astype(np.float32) - 127.0).astype(np.float32)
    s = np.where(scale == 255, np.float32(np.nan), s)
    return bf16.mul(v, np.repeat(s, BLOCK, axis=-1))


def grouped_mm(a, a_scale, b, b_scale, offs, return_abs=False):
    """_emulated_mxfp8_scaled_grouped_mm_2d_3d
    (torchao/prototype/moe_training/mxfp8_grouped_mm.py:959-1023):
    out[offs[e-1]:offs[e]] = dq(a_rows) @ dq(b[e])^T in bf16, fp32 accumulate.
    a [M,K] codes, a_scale [M,K/32]; b [E,N,K] codes, b_scale [E,N,K/32]; offs int [E]."""
    A = mx_dequant_bf16(a, a_scale).astype(np.float64)
    E, N, K = b.shape
    out = np.zeros((a.shape[0], N), dtype=np.float32)
    mag = np.zeros((a.shape[0], N), dtype=np.float32)  # sum_k |a||b| (tolerance scale for the tests)
    start = 0
    for e in range(E):
        end = int(offs[e])
        if end > start:
            Bd = mx_dequant_bf16(b[e], b_scale[e]).astype(np.float64)
            out[start:end] = (A[start:end] @ Bd.T).astype(np.float32)
            if return_abs:
                mag[start:end] = (np.abs(A[start:end]) @ np.abs(Bd).T).astype(np.float32)
        start = end
    if return_abs:
        return bf16.bf16_round(out), mag
    return bf16.bf16_round(out)


# ---- the "128 x 4 blocked" layout of the scales (a data format of the reference; the MI355X GEMMs take row-major scales) ----------------
def to_blocked(scales):
    """prototype/mx_formats/utils.py:31-72 (to_blocked, the torch path): [H, W] bytes -> flat 32 ceil(H/128) x 16 ceil(W/4) bytes.
    Pad to 128-row x 4-column blocks with zeros (:51-63), then per block [128, 4] -> [4, 32, 4] -> transpose(0, 1) -> [32, 16] (:66-68).
    Pinned by tests/golden/mx_blocked.npz (make_golden.py:make_mx_blocked)."""
    s = np.ascontiguousarray(scales)
    h, w = s.shape
    nrb, ncb = -(-h // 128), -(-w // 4)
    p = np.zeros((nrb * 128, ncb * 4), dtype=s.dtype)
    p[:h, :w] = s
    blocks = p.reshape(nrb, 128, ncb, 4).transpose(0, 2, 1, 3)              # :66
    return blocks.reshape(-1, 4, 32, 4).transpose(0, 2, 1, 3).reshape(-1)    # :67-68


def to_blocked_2d_M_groups(scales, group_offs):
    """moe_training/kernels/mxfp8/quant.py:136-196 (torch_to_blocked_2d_M_groups, the checker of torchao::mx_block_rearrange_2d_M_groups;
    output allocation of the CUDA op, csrc/cuda/mx_kernels/mxfp8_extension.cpp:221-228): every non-empty row group is blocked on its own
    and written at the row where the previous groups' 128-row-padded blocks end, inside a zero [rows + 128 G, 4 ceil(cols / 4)] buffer.
    Returns (blocked, start_row_after_padding int64 [G + 1])."""
    s = np.ascontiguousarray(scales)
    rows, cols = s.shape
    offs = [int(v) for v in np.asarray(group_offs).reshape(-1)]
    pcols = -(-cols // 4) * 4
    out = np.zeros((rows + 128 * len(offs), pcols), dtype=s.dtype)           # :159-160
    starts, begin = [0], 0
    for end in offs:                                                         # :163-190
        size = end - begin
        if size == 0:                                                        # :166-168
            starts.append(starts[-1])
            continue
        blocked = to_blocked(s[begin:end])                                   # :172
        padded = -(-size // 128) * 128
        out[starts[-1] : starts[-1] + padded] = blocked.reshape(-1, pcols)   # :181-186
        starts.append(starts[-1] + padded)
        begin = end
    return out, np.asarray(starts, dtype=np.int64)
