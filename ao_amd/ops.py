"""Torch-facing wrappers over the C-ABI, one per reference op on the hot path.

Same argument meaning, output allocation ("allocate and return", inputs
borrowed) and error behaviour as the ops they replace; each docstring cites the
reference call site.  Tensors must live on the GPU ("cuda" == HIP device on
ROCm); there is no CPU fallback.
"""
from typing import Optional

import torch

from . import _lib

__all__ = [
    "convert_weight_to_int4pack",
    "unpack_int4pack",
    "weight_int4pack_mm",
    "int4_dequantize",
    "int4_quantize_tinygemm",
    "int4_plain_quantize",
    "int4_quantize_hqq",
    "int8_quantize_rowwise",
    "int8_scaled_mm",
    "int_mm",
    "fp8_quantize_rowwise",
    "fp8_scaled_mm",
    "mxfp8_quantize",
    "mxfp8_grouped_mm",
    "dynamic_linear_fits",
    "dynamic_linear_preferred",
    "int8_dynamic_linear",
    "fp8_dynamic_linear",
    "fused_pad_token_groups",
    "fused_unpad_token_groups",
    "mx_block_rearrange_2d_M_groups",
    "mx_to_blocked",
    "int8_linear",
    "fp8_linear",
    "mxfp8_quantize_colwise",
    "fp8_grouped_mm",
    "rowwise_amax",
    "int8_quantize_rowwise_amax",
    "fp8_quantize_rowwise_amax",
    "fp8_mm_f32",
    "int8_scale_epilogue",
    "fp8_scale_epilogue",
]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """Device guard for a launch: a no-op when `dev` is already current (torch.cuda.device() costs ~3 us per call, as much as a
    small decode kernel)."""
    return _NO_GUARD if dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


def _require_gpu(name, *tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                f"{name}: expected all tensors on the GPU, got a tensor on {t.device} "
                "(the MI355X backend has no CPU fallback)"
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"{name}: tensors on different devices: {dev} vs {t.device}")
    return dev


def convert_weight_to_int4pack(w_u8: torch.Tensor, inner_k_tiles: int) -> torch.Tensor:
    """aten::_convert_weight_to_int4pack (call site: torchao/quantization/quantize_/
    workflows/int4/int4_tile_packed_to_4d_tensor.py:202).

    w_u8 uint8 [N, K/2] (even k in the high nibble) ->
    int32 [N/8, K/(inner_k_tiles*16), 32, inner_k_tiles/2].
    """
    dev = _require_gpu("convert_weight_to_int4pack", w_u8)
    if w_u8.dim() != 2:
        raise RuntimeError(f"convert_weight_to_int4pack: expected a 2-D tensor, got {w_u8.dim()}-D")
    if w_u8.dtype != torch.uint8:
        raise RuntimeError(f"convert_weight_to_int4pack: expected uint8, got {w_u8.dtype}")
    if not w_u8.is_contiguous():
        raise RuntimeError("convert_weight_to_int4pack: expected a contiguous tensor")
    if inner_k_tiles not in (2, 4, 8):
        raise RuntimeError(f"convert_weight_to_int4pack: innerKTiles must be 2, 4 or 8, got {inner_k_tiles}")
    n, kh = w_u8.shape
    k = kh * 2
    out = torch.empty(
        (n // 8, k // (inner_k_tiles * 16), 32, inner_k_tiles // 2), dtype=torch.int32, device=dev
    )
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int4_convert_weight_to_int4pack(
                _ptr(w_u8), _ptr(out), n, k, inner_k_tiles, _stream()
            )
        )
    return out


def unpack_int4pack(qdata: torch.Tensor, inner_k_tiles: int = 8) -> torch.Tensor:
    """Inverse of convert_weight_to_int4pack: int32 4-D -> uint8 [N, K/2]."""
    dev = _require_gpu("unpack_int4pack", qdata)
    if qdata.dim() != 4 or qdata.dtype != torch.int32 or not qdata.is_contiguous():
        raise RuntimeError("unpack_int4pack: expected a contiguous 4-D int32 tensor")
    n = qdata.shape[0] * 8
    k = qdata.shape[1] * inner_k_tiles * 16
    out = torch.empty((n, k // 2), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int4_unpack_int4pack(_ptr(qdata), _ptr(out), n, k, inner_k_tiles, _stream())
        )
    return out


def _int4_dims(name, qdata, scale_and_zero, group_size):
    if qdata.dim() != 4 or qdata.dtype != torch.int32:
        raise RuntimeError(f"{name}: mat2 must be a 4-D int32 tile-packed tensor")
    if qdata.shape[2] != 32 or qdata.shape[3] != 4:
        raise RuntimeError(f"{name}: mat2 must have shape [N/8, K/128, 32, 4] (innerKTiles = 8), got {tuple(qdata.shape)}")
    if not qdata.is_contiguous():
        raise RuntimeError(f"{name}: mat2 must be contiguous")
    n, k = qdata.shape[0] * 8, qdata.shape[1] * 128
    if group_size not in (32, 64, 128, 256):
        raise RuntimeError(f"{name}: qGroupSize must be one of 32, 64, 128, 256, got {group_size}")
    if scale_and_zero.dtype != torch.bfloat16 or scale_and_zero.dim() != 3:
        raise RuntimeError(f"{name}: qScaleAndZeros must be a 3-D bfloat16 tensor")
    if tuple(scale_and_zero.shape) != (k // group_size, n, 2):
        raise RuntimeError(
            f"{name}: qScaleAndZeros must have shape [K/g, N, 2] = {(k // group_size, n, 2)}, got {tuple(scale_and_zero.shape)}"
        )
    if not scale_and_zero.is_contiguous():
        raise RuntimeError(f"{name}: qScaleAndZeros must be contiguous")
    return n, k


def weight_int4pack_mm(
    x: torch.Tensor, qdata: torch.Tensor, group_size: int, scale_and_zero: torch.Tensor
) -> torch.Tensor:
    """aten::_weight_int4pack_mm(self, mat2, qGroupSize, qScaleAndZeros)
    (call site: int4_tile_packed_to_4d_tensor.py:287).

    x bf16 [M, K]; qdata int32 [N/8, K/128, 32, 4]; scale_and_zero bf16 [K/g, N, 2]
    -> bf16 [M, N].
    """
    dev = _require_gpu("weight_int4pack_mm", x, qdata, scale_and_zero)
    n, k = _int4_dims("weight_int4pack_mm", qdata, scale_and_zero, group_size)
    if x.dim() != 2:
        raise RuntimeError(f"weight_int4pack_mm: self must be 2-D, got {x.dim()}-D")
    if x.dtype != torch.bfloat16:
        raise RuntimeError(f"weight_int4pack_mm: self must be bfloat16, got {x.dtype}")
    if x.shape[1] != k:
        raise RuntimeError(f"weight_int4pack_mm: self has K={x.shape[1]} but mat2 has K={k}")
    if not x.is_contiguous():
        x = x.contiguous()
    m = x.shape[0]
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    if m == 0:
        return y
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int4_weight_int4pack_mm(
                _ptr(x), _ptr(qdata), _ptr(scale_and_zero), _ptr(y), m, n, k, group_size, _stream()
            )
        )
    return y


def int4_dequantize(qdata: torch.Tensor, scale_and_zero: torch.Tensor, group_size: int) -> torch.Tensor:
    """Packed int4 -> bf16 [N, K] with the reference's rounding sequence
    (torchao/quantization/utils.py:445-455, quant_primitives.py:999-1007)."""
    dev = _require_gpu("int4_dequantize", qdata, scale_and_zero)
    n, k = _int4_dims("int4_dequantize", qdata, scale_and_zero, group_size)
    w = torch.empty((n, k), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int4_dequantize(_ptr(qdata), _ptr(scale_and_zero), _ptr(w), n, k, group_size, _stream())
        )
    return w


def int4_quantize_tinygemm(w: torch.Tensor, group_size: int):
    """Fused tinygemm weight prep (int4_tile_packed_to_4d_tensor.py:129-236):
    bf16 [N, K] (N % 16 == 0, K % 128 == 0) -> (qdata int32 [N/8, K/128, 32, 4],
    scale_and_zero bf16 [K/g, N, 2])."""
    dev = _require_gpu("int4_quantize_tinygemm", w)
    if w.dim() != 2 or w.dtype != torch.bfloat16:
        raise RuntimeError("int4_quantize_tinygemm: expected a 2-D bfloat16 tensor")
    if not w.is_contiguous():
        w = w.contiguous()
    n, k = w.shape
    qdata = torch.empty((n // 8, k // 128, 32, 4), dtype=torch.int32, device=dev)
    sz = torch.empty((k // max(group_size, 1), n, 2), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int4_quantize_tinygemm(_ptr(w), _ptr(qdata), _ptr(sz), n, k, group_size, _stream())
        )
    return qdata, sz


def int4_quantize_hqq(w: torch.Tensor, group_size: int):
    """HQQ qparams + codes in the tinygemm format (Int4TilePackedTo4dTensor.from_hp with Int4ChooseQParamsAlgorithm.HQQ,
    int4_tile_packed_to_4d_tensor.py:149-236).  w bf16 [N, K] (N % 16 == 0, K % 128 == 0) ->
    (qdata int32 [N/8, K/128, 32, 4], scale_and_zero bf16 [K/g, N, 2]).  Synchronises the stream (the optimizer's early stop
    reads a global error every iteration, as the reference's does)."""
    dev = _require_gpu("int4_quantize_hqq", w)
    if w.dtype != torch.bfloat16 or w.dim() != 2:
        raise RuntimeError("int4_quantize_hqq: expected a 2-D bfloat16 tensor")
    w = w.contiguous()
    n, k = w.shape
    if n % 16 != 0 or k % 128 != 0 or group_size not in (32, 64, 128, 256) or k % group_size != 0:
        raise ValueError(f"int4_quantize_hqq: shape {tuple(w.shape)} / group_size {group_size} not supported (N % 16, K % 128, g 32|64|128|256)")
    lib = _lib.lib()
    ws = torch.empty((int(lib.ao_int4_hqq_workspace_bytes(n, k, group_size)),), dtype=torch.uint8, device=dev)
    nib = torch.empty((n, k // 2), dtype=torch.uint8, device=dev)
    sz = torch.empty((k // group_size, n, 2), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(lib.ao_int4_quantize_hqq(_ptr(w), _ptr(nib), _ptr(sz), _ptr(ws), n, k, group_size, _stream()))
    return convert_weight_to_int4pack(nib, 8), sz


def int4_plain_quantize(w: torch.Tensor, group_size: int, symmetric: bool = False):
    """Int4Tensor.from_hp's arithmetic, PLAIN packing (quantize_/workflows/int4/int4_tensor.py:130-186; mslk's
    int4_row_quantize_zp / int4_row_quantize + pack_int4 as restated by the reference, qat/fake_quantizer.py:148-190).
    w bf16 [N, K] -> (qdata uint8 [N, K/2] even k in the low nibble, scale bf16 [K/g, N], zero_point bf16 [K/g, N])."""
    dev = _require_gpu("int4_plain_quantize", w)
    if w.dtype != torch.bfloat16 or w.dim() != 2:
        raise RuntimeError("int4_plain_quantize: expected a 2-D bfloat16 tensor")
    w = w.contiguous()
    n, k = w.shape
    if group_size not in (32, 64, 128, 256) or k % 128 != 0 or k % group_size != 0:
        raise ValueError(f"int4_plain_quantize: group_size {group_size} / K={k} not supported (group 32|64|128|256, K % 128 == 0)")
    qdata = torch.empty((n, k // 2), dtype=torch.uint8, device=dev)
    scale = torch.empty((k // group_size, n), dtype=torch.bfloat16, device=dev)
    zero = torch.empty((k // group_size, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int4_plain_quantize(_ptr(w), _ptr(qdata), _ptr(scale), _ptr(zero), n, k, group_size, int(bool(symmetric)),
                                                     _stream()))
    return qdata, scale, zero


# ---------------------------------------------------------------------------
# int8 dynamic activation x int8 weight
# ---------------------------------------------------------------------------
def _as_rows(name, x, dtype):
    if x.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {x.dtype}")
    if x.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D tensor, got {x.dim()}-D")
    return x if x.is_contiguous() else x.contiguous()


def int8_quantize_rowwise(x: torch.Tensor):
    """Int8Tensor.from_hp(x, PerRow()) (symmetric, eps = fp32 eps):
    torchao/quantization/quantize_/workflows/int8/int8_tensor.py:176-248.
    x bf16 [M, K] -> (qdata int8 [M, K], scale fp32 [M, 1])."""
    dev = _require_gpu("int8_quantize_rowwise", x)
    x = _as_rows("int8_quantize_rowwise", x, torch.bfloat16)
    m, k = x.shape
    q = torch.empty((m, k), dtype=torch.int8, device=dev)
    s = torch.empty((m, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_quantize_rowwise(_ptr(x), _ptr(q), _ptr(s), m, k, _stream()))
    return q, s


def int8_scaled_mm(xq, x_scale, wq, w_scale, bias=None):
    """_int_scaled_matmul + weight-scale epilogue of the Int8Tensor linear
    (int8/kernels.py:114-144, int8_tensor.py:305-359).
    xq int8 [M, K]; x_scale fp32 [M(,1)]; wq int8 [N, K]; w_scale fp32 [N(,1)];
    bias bf16 [N] or None -> bf16 [M, N]."""
    dev = _require_gpu("int8_scaled_mm", xq, x_scale, wq, w_scale, bias)
    xq = _as_rows("int8_scaled_mm", xq, torch.int8)
    wq = _as_rows("int8_scaled_mm", wq, torch.int8)
    m, k = xq.shape
    n, k2 = wq.shape
    if k != k2:
        raise RuntimeError(f"int8_scaled_mm: K mismatch {k} vs {k2}")
    x_scale = x_scale.reshape(-1).to(torch.float32).contiguous()
    w_scale = w_scale.reshape(-1).to(torch.float32).contiguous()
    if x_scale.numel() != m or w_scale.numel() != n:
        raise RuntimeError("int8_scaled_mm: scales must be per-row ([M] and [N])")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
        if bias.numel() != n:
            raise RuntimeError("int8_scaled_mm: bias must have N elements")
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_int8_scaled_mm(
                _ptr(xq), _ptr(x_scale), _ptr(wq), _ptr(w_scale), _ptr(bias), _ptr(y), m, n, k, _stream()
            )
        )
    return y


def int_mm(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """aten::_int_mm(self int8 [M, K], mat2 int8 [K, N]) -> int32 [M, N]
    (call sites: int8/kernels.py:38-40,70).  mat2 is expected K-major (the
    `.t()` of a row-major [N, K] weight, as the reference passes it); other
    layouts are made so with one copy."""
    dev = _require_gpu("int_mm", a, b)
    a = _as_rows("int_mm", a, torch.int8)
    if b.dtype != torch.int8 or b.dim() != 2:
        raise RuntimeError("int_mm: mat2 must be a 2-D int8 tensor")
    if a.shape[1] != b.shape[0]:
        raise RuntimeError(f"int_mm: shapes {tuple(a.shape)} and {tuple(b.shape)} cannot be multiplied")
    b_t = b.t()
    if not b_t.is_contiguous():
        b_t = b_t.contiguous()
    m, k = a.shape
    n = b_t.shape[0]
    c = torch.empty((m, n), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_int_mm(_ptr(a), _ptr(b_t), _ptr(c), m, n, k, _stream()))
    return c


# ---------------------------------------------------------------------------
# float8 e4m3fn rowwise
# ---------------------------------------------------------------------------
def fp8_quantize_rowwise(x: torch.Tensor):
    """Float8Tensor.from_hp(x, float8_e4m3fn, PerRow())
    (quantize_/workflows/float8/float8_tensor.py:167-253).
    x bf16 [M, K] -> (qdata float8_e4m3fn [M, K], scale fp32 [M, 1])."""
    dev = _require_gpu("fp8_quantize_rowwise", x)
    x = _as_rows("fp8_quantize_rowwise", x, torch.bfloat16)
    m, k = x.shape
    q = torch.empty((m, k), dtype=torch.uint8, device=dev)
    s = torch.empty((m, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_fp8_quantize_rowwise(_ptr(x), _ptr(q), _ptr(s), m, k, _stream()))
    return q.view(torch.float8_e4m3fn), s


def _fp8_bytes(name, t):
    if t.dtype == torch.float8_e4m3fn:
        t = t.view(torch.uint8)
    elif t.dtype != torch.uint8:
        raise RuntimeError(f"{name}: expected float8_e4m3fn data, got {t.dtype}")
    return t


def fp8_scaled_mm(a, b, scale_a, scale_b, bias=None):
    """aten::_scaled_mm with rowwise scales (torchao/float8/inference.py:104-123),
    out_dtype bf16.  a e4m3 [M, K] row-major; b e4m3 [K, N] column-major (i.e. the
    `.t()` of a row-major [N, K] weight); scale_a fp32 [M, 1]; scale_b fp32 [1, N]."""
    dev = _require_gpu("fp8_scaled_mm", a, b, scale_a, scale_b, bias)
    a = _fp8_bytes("fp8_scaled_mm", a)
    b = _fp8_bytes("fp8_scaled_mm", b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0]:
        raise RuntimeError(f"fp8_scaled_mm: shapes {tuple(a.shape)} and {tuple(b.shape)} cannot be multiplied")
    if not a.is_contiguous():
        a = a.contiguous()
    b_t = b.t()
    if not b_t.is_contiguous():
        b_t = b_t.contiguous()
    m, k = a.shape
    n = b_t.shape[0]
    scale_a = scale_a.reshape(-1).to(torch.float32).contiguous()
    scale_b = scale_b.reshape(-1).to(torch.float32).contiguous()
    if scale_a.numel() != m or scale_b.numel() != n:
        raise RuntimeError("fp8_scaled_mm: only rowwise scaling is implemented (scale_a [M,1], scale_b [1,N])")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
        if bias.numel() != n:
            raise RuntimeError("fp8_scaled_mm: bias must have N elements")
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_fp8_scaled_mm(
                _ptr(a), _ptr(b_t), _ptr(scale_a), _ptr(scale_b), _ptr(bias), _ptr(y), m, n, k, _stream()
            )
        )
    return y


# ---------------------------------------------------------------------------
# K-sharded (row-parallel) 8-bit linears: building blocks (include/ao_mi355.h, last section)
# ---------------------------------------------------------------------------
def _strided_rows(name, x):
    """bf16 [M, K] whose rows are K-contiguous; a column slice of a wider row-major tensor is taken as is (row stride)."""
    if x.dtype != torch.bfloat16 or x.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D bfloat16 tensor, got {x.dim()}-D {x.dtype}")
    if x.shape[1] % 8 != 0:
        raise ValueError(f"{name}: K={x.shape[1]} must be a multiple of 8")
    ok = x.stride(1) == 1 and x.stride(0) >= x.shape[1] and x.stride(0) % 8 == 0 and x.data_ptr() % 16 == 0
    if not ok and x.shape[0] > 0:
        x = x.contiguous()
    return x, (x.stride(0) if x.shape[0] > 1 else max(x.shape[1], x.stride(0)))


def rowwise_amax(x: torch.Tensor) -> torch.Tensor:
    """max |x| per row of a bf16 [M, K] matrix (fp32 [M]): the local half of a full-K activation scale."""
    dev = _require_gpu("rowwise_amax", x)
    x, ldx = _strided_rows("rowwise_amax", x)
    m, k = x.shape
    out = torch.empty((m,), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_rowwise_amax(_ptr(x), ldx, _ptr(out), m, k, _stream()))
    return out


def _quantize_rowwise_amax(name, fn, x, amax, qdtype):
    dev = _require_gpu(name, x, amax)
    x, ldx = _strided_rows(name, x)
    m, k = x.shape
    amax = amax.reshape(-1).to(torch.float32).contiguous()
    if amax.numel() != m:
        raise RuntimeError(f"{name}: amax must have one entry per row ({m}), got {amax.numel()}")
    q = torch.empty((m, k), dtype=qdtype, device=dev)
    s = torch.empty((m, 1), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(fn(_ptr(x), ldx, _ptr(amax), _ptr(q), _ptr(s), m, k, _stream()))
    return q, s


def int8_quantize_rowwise_amax(x: torch.Tensor, amax: torch.Tensor):
    """Int8Tensor.from_hp arithmetic (int8_tensor.py:191-230) on a K shard of the activation, with the rows' amax over the
    full K given: the shard of the unsharded qdata, and the unsharded scale."""
    return _quantize_rowwise_amax("int8_quantize_rowwise_amax", _lib.lib().ao_int8_quantize_rowwise_amax, x, amax, torch.int8)


def fp8_quantize_rowwise_amax(x: torch.Tensor, amax: torch.Tensor):
    """Float8Tensor.from_hp arithmetic (float8_tensor.py:167-253) on a K shard, amax over the full K given."""
    q, s = _quantize_rowwise_amax("fp8_quantize_rowwise_amax", _lib.lib().ao_fp8_quantize_rowwise_amax, x, amax, torch.uint8)
    return q.view(torch.float8_e4m3fn), s


def fp8_mm_f32(a, b):
    """Unscaled e4m3 [M, K] @ e4m3 [K, N] (column-major, the `.t()` of a row-major [N, K] weight) -> fp32 [M, N]: the
    accumulator of aten::_scaled_mm before its scale epilogue."""
    dev = _require_gpu("fp8_mm_f32", a, b)
    a = _fp8_bytes("fp8_mm_f32", a)
    b = _fp8_bytes("fp8_mm_f32", b)
    if a.dim() != 2 or b.dim() != 2 or a.shape[1] != b.shape[0]:
        raise RuntimeError(f"fp8_mm_f32: shapes {tuple(a.shape)} and {tuple(b.shape)} cannot be multiplied")
    if not a.is_contiguous():
        a = a.contiguous()
    b_t = b.t()
    if not b_t.is_contiguous():
        b_t = b_t.contiguous()
    m, k = a.shape
    n = b_t.shape[0]
    c = torch.empty((m, n), dtype=torch.float32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_fp8_mm_f32(_ptr(a), _ptr(b_t), _ptr(c), m, n, k, _stream()))
    return c


def _scale_epilogue(name, fn, acc, acc_dtype, row_scale, col_scale, bias):
    dev = _require_gpu(name, acc, row_scale, col_scale, bias)
    if acc.dtype != acc_dtype or acc.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D {acc_dtype} accumulator, got {acc.dim()}-D {acc.dtype}")
    acc = acc.contiguous()
    m, n = acc.shape
    row_scale = row_scale.reshape(-1).to(torch.float32).contiguous()
    col_scale = col_scale.reshape(-1).to(torch.float32).contiguous()
    if row_scale.numel() != m or col_scale.numel() != n:
        raise RuntimeError(f"{name}: scales must be per-row ([M] and [N])")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
        if bias.numel() != n:
            raise RuntimeError(f"{name}: bias must have N elements")
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(fn(_ptr(acc), _ptr(row_scale), _ptr(col_scale), _ptr(bias), _ptr(y), m, n, _stream()))
    return y


def int8_scale_epilogue(acc, x_scale, w_scale, bias=None):
    """bf16(bf16(acc * x_scale[m]) * w_scale[n] (+ bias)) over int32 accumulators (int8_tensor.py:315-359)."""
    return _scale_epilogue("int8_scale_epilogue", _lib.lib().ao_int8_scale_epilogue, acc, torch.int32, x_scale, w_scale, bias)


def fp8_scale_epilogue(acc, scale_a, scale_b, bias=None):
    """bf16(acc * scale_a[m] * scale_b[n] (+ bias)) over fp32 accumulators (float8/inference.py:104-123)."""
    return _scale_epilogue("fp8_scale_epilogue", _lib.lib().ao_fp8_scale_epilogue, acc, torch.float32, scale_a, scale_b, bias)


# ---- granularity / mapping variants of the dynamic-activation formats (PerTensor; int8 ASYMMETRIC activations) ----------
def _tensor_amax_rows(x):
    """[M] fp32 filled with max |x| over the WHOLE matrix (PerTensor granularity: block_size = shape)."""
    return rowwise_amax(x).amax().expand(x.shape[0])


def int8_quantize_tensorwise(x: torch.Tensor):
    """Int8Tensor.from_hp(x, PerTensor()) (int8_tensor.py:191-230 with block_size = shape): bf16 [M, K] ->
    (qdata int8 [M, K], scale fp32 [1, 1])."""
    q, s = int8_quantize_rowwise_amax(x, _tensor_amax_rows(x))
    return q, s[:1].reshape(1, 1)


def fp8_quantize_tensorwise(x: torch.Tensor):
    """Float8Tensor.from_hp(x, e4m3, PerTensor()) (float8_tensor.py:167-253 with block_size = shape): bf16 [M, K] ->
    (qdata e4m3fn [M, K], scale fp32 [1, 1])."""
    q, s = fp8_quantize_rowwise_amax(x, _tensor_amax_rows(x))
    return q, s[:1].reshape(1, 1)


def int8_quantize_rowwise_asym(x: torch.Tensor):
    """Int8Tensor.from_hp(x, PerRow(), mapping_type=ASYMMETRIC) (int8_tensor.py:191-236): bf16 [M, K] ->
    (qdata int8 [M, K], scale fp32 [M, 1], zero_point int8 [M, 1])."""
    dev = _require_gpu("int8_quantize_rowwise_asym", x)
    x = _as_rows("int8_quantize_rowwise_asym", x, torch.bfloat16)
    m, k = x.shape
    q = torch.empty((m, k), dtype=torch.int8, device=dev)
    s = torch.empty((m, 1), dtype=torch.float32, device=dev)
    zp = torch.empty((m, 1), dtype=torch.int8, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_quantize_rowwise_asym(_ptr(x), _ptr(q), _ptr(s), _ptr(zp), m, k, _stream()))
    return q, s, zp


def int8_quantize_static(x: torch.Tensor, scale: torch.Tensor, zero_point: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Int8Tensor.from_hp(x, ..., scale=scale, zero_point=zero_point) (static quantization, int8_tensor.py:212-231): bf16 [M, K] with an fp32
    scale of 1 or M elements (and an int8 zero-point of the same shape, or None) -> qdata int8 [M, K]."""
    dev = _require_gpu("int8_quantize_static", x, scale, zero_point)
    x = _as_rows("int8_quantize_static", x, torch.bfloat16)
    m, k = x.shape
    scale = scale.reshape(-1).to(torch.float32).contiguous()
    if scale.numel() not in (1, m):
        raise RuntimeError(f"int8_quantize_static: scale must have 1 or M = {m} elements, got {scale.numel()}")
    if zero_point is not None:
        zero_point = zero_point.reshape(-1).to(torch.int8).contiguous()
        if zero_point.numel() != scale.numel():
            raise RuntimeError("int8_quantize_static: zero_point must have the shape of scale")
    q = torch.empty((m, k), dtype=torch.int8, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_quantize_static(_ptr(x), _ptr(scale), _ptr(zero_point), int(scale.numel() == m and m > 1), _ptr(q), m, k, _stream()))
    return q


def int8_linear_static(x2, wq, w_scale, act_scale, act_zero_point=None, w_row_sums=None, bias=None):
    """The Int8Tensor F.linear with STATIC activation qparams (Int8StaticActivationInt8WeightConfig): cast with the given scale /
    zero-point, then the dynamic path's GEMM + epilogue (zero-point-corrected when one is given)."""
    m = x2.shape[0]
    xq = int8_quantize_static(x2, act_scale, act_zero_point)
    xs = act_scale.reshape(-1).to(torch.float32)
    xs = xs.expand(m) if xs.numel() == 1 else xs
    if act_zero_point is None:
        return int8_scaled_mm(xq, xs, wq, w_scale, bias)
    zp = act_zero_point.reshape(-1)
    zp = zp.expand(m) if zp.numel() == 1 else zp
    if w_row_sums is None:
        w_row_sums = int8_row_sums(wq)
    return int8_scale_epilogue_asym(int_mm(xq, wq.t()), xs, zp, w_row_sums, w_scale, bias)


def int8_row_sums(wq: torch.Tensor) -> torch.Tensor:
    """rowsum of an int8 [N, K] weight as int32 [N]: the zero-point correction's `weight_tensor.qdata.sum(dim=-1)` (int8_tensor.py:326)."""
    dev = _require_gpu("int8_row_sums", wq)
    wq = _as_rows("int8_row_sums", wq, torch.int8)
    n, k = wq.shape
    out = torch.empty((n,), dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_row_sums(_ptr(wq), _ptr(out), n, k, _stream()))
    return out


def int8_scale_epilogue_asym(acc, x_scale, x_zero_point, w_row_sums, w_scale, bias=None):
    """bf16(bf16(bf16(acc * sx[m]) - bf16((zp[m] * sx[m]) * wsum[n])) * sw[n] (+ bias)) over int32 accumulators
    (int8_tensor.py:305-346)."""
    name = "int8_scale_epilogue_asym"
    dev = _require_gpu(name, acc, x_scale, x_zero_point, w_row_sums, w_scale, bias)
    if acc.dtype != torch.int32 or acc.dim() != 2:
        raise RuntimeError(f"{name}: expected a 2-D int32 accumulator, got {acc.dim()}-D {acc.dtype}")
    acc = acc.contiguous()
    m, n = acc.shape
    x_scale = x_scale.reshape(-1).to(torch.float32).contiguous()
    x_zero_point = x_zero_point.reshape(-1).to(torch.int8).contiguous()
    w_scale = w_scale.reshape(-1).to(torch.float32).contiguous()
    w_row_sums = w_row_sums.reshape(-1).to(torch.int32).contiguous()
    if x_scale.numel() != m or x_zero_point.numel() != m or w_scale.numel() != n or w_row_sums.numel() != n:
        raise RuntimeError(f"{name}: x_scale / x_zero_point must have M = {m} entries, w_scale / w_row_sums N = {n}")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
        if bias.numel() != n:
            raise RuntimeError(f"{name}: bias must have N elements")
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_int8_scale_epilogue_asym(_ptr(acc), _ptr(x_scale), _ptr(x_zero_point), _ptr(w_row_sums), _ptr(w_scale),
                                                          _ptr(bias), _ptr(y), m, n, _stream()))
    return y


def int8_linear_asym(x2, wq, w_scale, w_row_sums, bias=None):
    """The Int8Tensor F.linear with ASYMMETRIC per-row activation quantization (int8_tensor.py:266-359): cast, int32 GEMM,
    zero-point-corrected scale epilogue (three launches; the symmetric default keeps the fused epilogue)."""
    xq, xs, zp = int8_quantize_rowwise_asym(x2)
    return int8_scale_epilogue_asym(int_mm(xq, wq.t()), xs, zp, w_row_sums, w_scale, bias)


def int8_linear_tensorwise(x2, wq, w_scale, bias=None):
    """The Int8Tensor F.linear with a PerTensor activation (int8_tensor.py:266-359): one amax over the whole activation, the
    fused two-rounding epilogue with that scalar broadcast over the rows.  `w_scale` fp32 [N] (a PerTensor weight scale already
    broadcast)."""
    xq, xs = int8_quantize_tensorwise(x2)
    return int8_scaled_mm(xq, xs.reshape(-1).expand(x2.shape[0]), wq, w_scale, bias)


def fp8_linear_tensorwise(x2, wq, w_scale, bias=None):
    """The Float8Tensor F.linear with PerTensor scales on both operands (tensorwise aten::_scaled_mm, float8/inference.py:68-123):
    `w_scale` is the weight's [1, 1] scale."""
    xq, xs = fp8_quantize_tensorwise(x2)
    n = wq.shape[0]
    return fp8_scaled_mm(xq, wq.t(), xs.reshape(-1).expand(x2.shape[0]), w_scale.reshape(-1).expand(n), bias)


def fp8_linear_clamped(x2, wq, w_scale, bias=None, lb: float = -1.0, ub: float = -1.0, tensorwise: bool = False):
    """The Float8Tensor F.linear with activation-value bounds (Float8DynamicActivationFloat8WeightConfig(activation_value_lb / _ub) ->
    hp_value_lb / hp_value_ub of _choose_scale_float8, quant_primitives.py:2203-2204): the amax (per row, or over the whole
    activation) is clamped to [lb, ub] before the scale is taken from it; values beyond ub saturate at +-448 in the cast.
    A negative bound means "not set" (the dispatcher schema has no optional floats)."""
    amax = rowwise_amax(x2)
    if tensorwise:
        amax = amax.amax().expand(x2.shape[0])
    amax = amax.clamp(min=lb if lb >= 0 else None, max=ub if ub >= 0 else None)
    amax = amax.to(torch.bfloat16).to(torch.float32)  # torch.clamp on the bf16 amax rounds the bound to bf16
    xq, xs = fp8_quantize_rowwise_amax(x2, amax)
    n = wq.shape[0]
    sb = w_scale.reshape(-1).expand(n) if w_scale.numel() == 1 else w_scale.t()
    return fp8_scaled_mm(xq, wq.t(), xs, sb, bias)


# ---------------------------------------------------------------------------
# MXFP8
# ---------------------------------------------------------------------------
MX_SCALE_MODES = {"floor": 0, "rceil": 1}


def mxfp8_quantize(x: torch.Tensor, scaling_mode: str = "rceil"):
    """Rowwise (1x32) MXFP8 cast: torchao::mxfp8_quantize(rowwise=True) /
    to_mx(x, float8_e4m3fn, 32, mode) (prototype/mx_formats/mx_tensor.py:228-409).
    x bf16 [..., C] -> (data float8_e4m3fn [..., C], scale float8_e8m0fnu [..., C/32])."""
    dev = _require_gpu("mxfp8_quantize", x)
    if x.dtype != torch.bfloat16:
        raise RuntimeError(f"mxfp8_quantize: expected bfloat16, got {x.dtype}")
    if not x.is_contiguous():
        raise RuntimeError("mxfp8_quantize: expected a contiguous tensor")  # to_mx asserts the same
    mode = MX_SCALE_MODES.get(str(getattr(scaling_mode, "value", scaling_mode)).lower())
    if mode is None:
        raise RuntimeError(f"mxfp8_quantize: unsupported scaling mode {scaling_mode!r} (floor | rceil)")
    c = x.shape[-1]
    if c % 32 != 0:
        raise RuntimeError(f"mxfp8_quantize: the last dimension of shape {tuple(x.shape)} must be divisible by 32")
    r = x.numel() // c
    q = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    s = torch.empty((*x.shape[:-1], c // 32), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_mxfp8_quantize_rowwise(_ptr(x), _ptr(q), _ptr(s), r, c, mode, _stream()))
    return q.view(torch.float8_e4m3fn), s.view(torch.float8_e8m0fnu)


def fp8_grouped_mm(a, scale_a, b, scale_b, offs):
    """Float8Tensor's aten::_grouped_mm, rowwise scales (float8_tensor.py:1085-1122).  a e4m3 [M, K]; scale_a fp32 [M(,1)];
    b e4m3 [E, N, K]; scale_b fp32 [E, N(,1)]; offs int32 [E] -> bf16 [M, N] (rows past offs[-1] are zero)."""
    dev = _require_gpu("fp8_grouped_mm", a, scale_a, b, scale_b, offs)
    a = _fp8_bytes("fp8_grouped_mm", a).contiguous()
    b = _fp8_bytes("fp8_grouped_mm", b).contiguous()
    if a.dim() != 2 or b.dim() != 3 or a.shape[1] != b.shape[2]:
        raise RuntimeError(f"fp8_grouped_mm: shapes {tuple(a.shape)} and {tuple(b.shape)} are not compatible")
    m, k = a.shape
    e, n, _ = b.shape
    scale_a = scale_a.reshape(-1).to(torch.float32).contiguous()
    scale_b = scale_b.reshape(-1).to(torch.float32).contiguous()
    if scale_a.numel() != m or scale_b.numel() != e * n:
        raise RuntimeError("fp8_grouped_mm: scales must be rowwise ([M] and [E, N])")
    if offs.dtype != torch.int32 or offs.numel() != e:
        raise RuntimeError("fp8_grouped_mm: offs must be int32 [E]")
    out = torch.zeros((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_fp8_grouped_mm(_ptr(a), _ptr(scale_a), _ptr(b), _ptr(scale_b), _ptr(offs.contiguous()), _ptr(out), m, n, k, e,
                                                _stream()))
    return out


def mxfp8_quantize_colwise(x: torch.Tensor, scaling_mode: str = "rceil"):
    """Colwise (32x1) MXFP8 cast: torchao::mxfp8_quantize(colwise=True) == to_mx(x.t()).t()
    (csrc/cuda/mx_kernels/mxfp8_extension.cpp:160-175).  x bf16 [R, C] -> (data float8_e4m3fn {R, C} with strides {1, R},
    scale float8_e8m0fnu {C, R/32} with strides {1, C})."""
    dev = _require_gpu("mxfp8_quantize_colwise", x)
    if x.dtype != torch.bfloat16 or x.dim() != 2 or not x.is_contiguous():
        raise RuntimeError("mxfp8_quantize_colwise: expected a contiguous 2-D bfloat16 tensor")
    mode = MX_SCALE_MODES.get(str(getattr(scaling_mode, "value", scaling_mode)).lower())
    if mode is None:
        raise RuntimeError(f"mxfp8_quantize_colwise: unsupported scaling mode {scaling_mode!r} (floor | rceil)")
    r, c = x.shape
    if r % 32 != 0 or c % 32 != 0:
        raise RuntimeError(f"mxfp8_quantize_colwise: shape {tuple(x.shape)} must be multiples of 32 in both dimensions")
    qt = torch.empty((c, r), dtype=torch.uint8, device=dev)
    st = torch.empty((r // 32, c), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_mxfp8_quantize_colwise(_ptr(x), _ptr(qt), _ptr(st), r, c, mode, _stream()))
    return qt.view(torch.float8_e4m3fn).t(), st.view(torch.float8_e8m0fnu).t()


def mxfp8_quantize_3d(x: torch.Tensor, block_size: int = 32, scale_block_dim1: int = 32, scale_block_dim2: int = 1,
                      scaling_mode: str = "rceil"):
    """mxfp8_quantize_cuda_3d with logical (un-blocked) scales (prototype/moe_training/kernels/mxfp8/quant.py:1413-1440): x bf16
    [E, N, K] -> (data float8_e4m3fn {E, N, K} column-major per expert, i.e. strides {N K, 1, N}; scale float8_e8m0fnu
    {E, K, N/32} with strides {N/32 * K, 1, K}): one scale per 32 values of the MIDDLE dimension.  Only the (32, 1) scale block."""
    dev = _require_gpu("mxfp8_quantize_3d", x)
    if x.dtype != torch.bfloat16 or x.dim() != 3 or not x.is_contiguous():
        raise RuntimeError("mxfp8_quantize_3d: expected a contiguous 3-D bfloat16 tensor")
    if block_size != 32 or (scale_block_dim1, scale_block_dim2) != (32, 1):
        raise NotImplementedError("mxfp8_quantize_3d on MI355X implements block_size 32 with (scale_block_dim1, scale_block_dim2) = (32, 1)")
    mode = MX_SCALE_MODES.get(str(getattr(scaling_mode, "value", scaling_mode)).lower())
    if mode is None:
        raise RuntimeError(f"mxfp8_quantize_3d: unsupported scaling mode {scaling_mode!r} (floor | rceil)")
    e, r, c = x.shape
    if r % 32 != 0 or c % 32 != 0:
        raise RuntimeError(f"mxfp8_quantize_3d: shape {tuple(x.shape)}: N and K must be multiples of 32")
    qt = torch.empty((e, c, r), dtype=torch.uint8, device=dev)
    st = torch.empty((e, r // 32, c), dtype=torch.uint8, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_mxfp8_quantize_colwise_3d(_ptr(x), _ptr(qt), _ptr(st), e, r, c, mode, _stream()))
    return qt.view(torch.float8_e4m3fn).transpose(-2, -1), st.view(torch.float8_e8m0fnu).transpose(-2, -1)


def mxfp8_grouped_mm(a, a_scale, b, b_scale, offs):
    """aten::_scaled_grouped_mm for MXFP8 (mxfp8_grouped_mm.py:541), numerics of the
    emulated path (:959-1023).  a e4m3 [M, K]; a_scale e8m0 [M, K/32]; b e4m3
    [E, N, K]; b_scale e8m0 [E, N, K/32]; offs int32 [E] -> bf16 [M, N]."""
    dev = _require_gpu("mxfp8_grouped_mm", a, a_scale, b, b_scale, offs)
    a = _fp8_bytes("mxfp8_grouped_mm", a).contiguous()
    b = _fp8_bytes("mxfp8_grouped_mm", b).contiguous()
    a_scale = a_scale.view(torch.uint8).contiguous()
    b_scale = b_scale.view(torch.uint8).contiguous()
    if a.dim() != 2 or b.dim() != 3 or a.shape[1] != b.shape[2]:
        raise RuntimeError(f"mxfp8_grouped_mm: A must be [M, K] and B [E, N, K], got {tuple(a.shape)} {tuple(b.shape)}")
    m, k = a.shape
    e, n, _ = b.shape
    if tuple(a_scale.shape) != (m, k // 32) or tuple(b_scale.shape) != (e, n, k // 32):
        raise RuntimeError("mxfp8_grouped_mm: scales must be [M, K/32] and [E, N, K/32]")
    if offs.dtype != torch.int32 or offs.numel() != e:
        raise RuntimeError("mxfp8_grouped_mm: offs must be int32 [E]")
    out = torch.zeros((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_mxfp8_grouped_mm(
                _ptr(a), _ptr(a_scale), _ptr(b), _ptr(b_scale), _ptr(offs.contiguous()), _ptr(out), m, n, k, e, _stream()
            )
        )
    return out


def _mx_mode(scaling_mode) -> int:
    mode = MX_SCALE_MODES.get(str(getattr(scaling_mode, "value", scaling_mode)).lower())
    if mode is None:
        raise RuntimeError(f"unsupported MX scaling mode {scaling_mode!r} (floor | rceil)")
    return mode


def mxfp8_grouped_mm_dyn_fits(m, n, k, e) -> bool:
    """Whether mxfp8_grouped_mm_dyn takes the shape (host logic: decode-size groups, K % 512 == 0)."""
    return bool(_lib.lib().ao_mxfp8_grouped_mm_dyn_fits(int(m), int(n), int(k), int(e)))


def mxfp8_grouped_mm_dyn(a, b, b_scale, offs, scaling_mode="rceil"):
    """to_mx(a) + the MXFP8 grouped mm in ONE launch (mxfp8_grouped_mm.py:330-371): a bf16 [M, K]; b e4m3 [E, N, K];
    b_scale e8m0 [E, N, K/32]; offs int32 [E] -> bf16 [M, N], bit-identical to mxfp8_quantize(a) followed by mxfp8_grouped_mm.  Rows past
    offs[-1] are left unwritten (torch._scaled_grouped_mm's contract; mxfp8_grouped_mm zero-fills them)."""
    dev = _require_gpu("mxfp8_grouped_mm_dyn", a, b, b_scale, offs)
    if a.dtype != torch.bfloat16 or a.dim() != 2:
        raise RuntimeError(f"mxfp8_grouped_mm_dyn: a must be a 2-D bfloat16 tensor, got {a.dtype} {tuple(a.shape)}")
    a = a.contiguous()
    b = _fp8_bytes("mxfp8_grouped_mm_dyn", b).contiguous()
    b_scale = b_scale.view(torch.uint8).contiguous()
    if b.dim() != 3 or a.shape[1] != b.shape[2]:
        raise RuntimeError(f"mxfp8_grouped_mm_dyn: A must be [M, K] and B [E, N, K], got {tuple(a.shape)} {tuple(b.shape)}")
    m, k = a.shape
    e, n, _ = b.shape
    if tuple(b_scale.shape) != (e, n, k // 32):
        raise RuntimeError("mxfp8_grouped_mm_dyn: b_scale must be [E, N, K/32]")
    if offs.dtype != torch.int32 or offs.numel() != e:
        raise RuntimeError("mxfp8_grouped_mm_dyn: offs must be int32 [E]")
    out = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_mxfp8_grouped_mm_dyn(_ptr(a), _ptr(b), _ptr(b_scale), _ptr(offs.contiguous()), _ptr(out), m, n, k, e,
                                                      _mx_mode(scaling_mode), _stream()))
    return out


def mxfp8_grouped_mm_pair_fits(m, n, k, e) -> bool:
    """Whether the pair forms take the shape (host logic: the mxfp8_grouped_mm_dyn conditions with twice the tiles)."""
    return bool(_lib.lib().ao_mxfp8_grouped_mm_pair_fits(int(m), int(n), int(k), int(e)))


def mxfp8_grouped_mm_pair(a, b1, b1_scale, b3, b3_scale, offs, scaling_mode="rceil", a_scale=None):
    """x @ w1 and x @ w3 of an MoE layer in ONE launch: two expert-weight tensors of one shape [E, N, K] (e4m3 + E8M0 [E, N, K/32]) against
    the same activations.  a bf16 [M, K] (the 1 x 32 cast fused into the kernel, as mxfp8_grouped_mm_dyn) or, with a_scale, its e4m3 codes.
    -> (y1, y3) bf16 [M, N]: the two single products' values -- bit for bit where the stream-K shares cut the tiles at the same k steps as the
    single launches do, else up to the fp32 summation order of a cut tile's pieces (single elements one bf16 ulp apart; reproducible from
    launch to launch).  Rows past offs[-1] are left unwritten."""
    dev = _require_gpu("mxfp8_grouped_mm_pair", a, b1, b1_scale, b3, b3_scale, offs)
    b1 = _fp8_bytes("mxfp8_grouped_mm_pair", b1).contiguous()
    b3 = _fp8_bytes("mxfp8_grouped_mm_pair", b3).contiguous()
    b1_scale, b3_scale = b1_scale.view(torch.uint8).contiguous(), b3_scale.view(torch.uint8).contiguous()
    if a.dim() != 2 or b1.dim() != 3 or b1.shape != b3.shape or a.shape[1] != b1.shape[2]:
        raise RuntimeError(f"mxfp8_grouped_mm_pair: A must be [M, K] and both B [E, N, K], got {tuple(a.shape)} {tuple(b1.shape)} {tuple(b3.shape)}")
    m, k = a.shape
    e, n, _ = b1.shape
    if tuple(b1_scale.shape) != (e, n, k // 32) or tuple(b3_scale.shape) != (e, n, k // 32):
        raise RuntimeError("mxfp8_grouped_mm_pair: weight scales must be [E, N, K/32]")
    if offs.dtype != torch.int32 or offs.numel() != e:
        raise RuntimeError("mxfp8_grouped_mm_pair: offs must be int32 [E]")
    y1 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    y3 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        if a_scale is None:
            if a.dtype != torch.bfloat16:
                raise RuntimeError(f"mxfp8_grouped_mm_pair: a must be bfloat16 (or e4m3 codes with a_scale), got {a.dtype}")
            _lib.check(_lib.lib().ao_mxfp8_grouped_mm_dyn_pair(_ptr(a.contiguous()), _ptr(b1), _ptr(b1_scale), _ptr(b3), _ptr(b3_scale), _ptr(offs.contiguous()),
                                                               _ptr(y1), _ptr(y3), m, n, k, e, _mx_mode(scaling_mode), _stream()))
        else:
            aq = _fp8_bytes("mxfp8_grouped_mm_pair", a).contiguous()
            asc = a_scale.view(torch.uint8).contiguous()
            if tuple(asc.shape) != (m, k // 32):
                raise RuntimeError("mxfp8_grouped_mm_pair: a_scale must be [M, K/32]")
            _lib.check(_lib.lib().ao_mxfp8_grouped_mm_pair(_ptr(aq), _ptr(asc), _ptr(b1), _ptr(b1_scale), _ptr(b3), _ptr(b3_scale), _ptr(offs.contiguous()),
                                                           _ptr(y1), _ptr(y3), m, n, k, e, _stream()))
    return y1, y3


def _check_token_groups(name, inputs, offsets):
    if inputs.dim() != 2:
        raise AssertionError("input activations must be 2d")
    if inputs.dtype not in (torch.float32, torch.bfloat16):
        raise AssertionError("inputs must be float32 or bfloat16")
    if offsets.dtype != torch.int32:
        raise AssertionError("offsets must be int32")
    if offsets.dim() != 1 or offsets.numel() == 0:
        raise RuntimeError(f"{name}: offsets must be a non-empty 1-d tensor of group end offsets")


def fused_pad_token_groups(inputs, offsets, alignment_size=32):
    """torchao::fused_pad_token_groups (kernels/mxfp8/quant.py:1244-1283; semantics torch_pad_token_groups,
    quant.py:368-430): every token group is moved to a start that is a multiple of `alignment_size` inside a
    zero-filled buffer of the reference's upper-bound size (no host sync).  Returns (padded_tokens,
    padded_group_start_offsets, padded_group_end_offsets)."""
    dev = _require_gpu("fused_pad_token_groups", inputs, offsets)
    _check_token_groups("fused_pad_token_groups", inputs, offsets)
    inputs = inputs.contiguous()
    offsets = offsets.contiguous()
    tokens, dim = inputs.shape
    groups = offsets.numel()
    if alignment_size <= 0:
        raise ValueError(f"fused_pad_token_groups: alignment_size must be positive, got {alignment_size}")
    rows = _lib.lib().ao_moe_padded_rows(tokens, groups, alignment_size)
    padded = torch.empty((rows, dim), dtype=inputs.dtype, device=dev)  # the kernel writes every row
    starts = torch.empty(groups, dtype=torch.int32, device=dev)
    ends = torch.empty(groups, dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_moe_pad_token_groups(
                _ptr(inputs), _ptr(offsets), _ptr(padded), _ptr(starts), _ptr(ends), tokens, dim, inputs.element_size(), groups,
                alignment_size, _stream()
            )
        )
    return padded, starts, ends


def _scale_bytes(name, scales):
    if scales.dim() != 2:
        raise AssertionError("scales_tensor must be 2D")
    if scales.dtype not in (torch.uint8, torch.float8_e8m0fnu):
        raise AssertionError("scales_tensor must be uint8 or float8_e8m0fnu")
    return scales.contiguous()


def mx_block_rearrange_2d_M_groups(scales_tensor, input_offsets, chunks_per_tb=4):
    """torchao::mx_block_rearrange_2d_M_groups (kernels/mxfp8/quant.py:969-973, 1183-1225; semantics torch_to_blocked_2d_M_groups,
    quant.py:136-196): E8M0 scales [rows, cols] of token groups ending at `input_offsets` -> the reference's 128 x 4 blocked layout, every
    group padded to 128-row blocks, in a [rows + 128 groups, 4 ceil(cols / 4)] tensor (zero elsewhere).  `chunks_per_tb` is the
    reference's launch-shape knob: validated, no meaning here."""
    dev = _require_gpu("mx_block_rearrange_2d_M_groups", scales_tensor, input_offsets)
    s = _scale_bytes("mx_block_rearrange_2d_M_groups", scales_tensor)
    if input_offsets.dtype != torch.int32:
        raise AssertionError("input_offsets must be int32")
    if input_offsets.dim() != 1 or input_offsets.numel() == 0:
        raise RuntimeError("mx_block_rearrange_2d_M_groups: input_offsets must be a non-empty 1-d tensor of group end offsets")
    if chunks_per_tb not in (1, 4, 8, 16):
        raise AssertionError("chunks_per_tb must be 1, 4, 8, or 16")
    rows, cols = s.shape
    groups = input_offsets.numel()
    out = torch.empty((_lib.lib().ao_mx_blocked_rows(rows, groups), (cols + 3) // 4 * 4), dtype=s.dtype, device=dev)  # every byte is written
    with _on(dev):
        _lib.check(_lib.lib().ao_mx_block_rearrange_2d_m_groups(_ptr(s), _ptr(input_offsets.contiguous()), _ptr(out), rows, cols, groups, _stream()))
    return out


def mx_to_blocked(scales_tensor):
    """to_blocked (prototype/mx_formats/utils.py:31-72): E8M0 scales [H, W] -> flat 32 ceil(H / 128) x 16 ceil(W / 4) bytes in the 128 x 4
    blocked layout."""
    dev = _require_gpu("mx_to_blocked", scales_tensor)
    s = _scale_bytes("mx_to_blocked", scales_tensor)
    rows, cols = s.shape
    out = torch.empty(((rows + 127) // 128 * 128) * ((cols + 3) // 4 * 4), dtype=s.dtype, device=dev)
    if out.numel():
        with _on(dev):
            _lib.check(_lib.lib().ao_mx_to_blocked(_ptr(s), _ptr(out), rows, cols, _stream()))
    return out


def fp8_int4_linear(xq, x_scale, qdata, scale_and_zero, group_size, bias=None):
    """mslk.f8i4bf16_rowwise's contract (int4_tensor.py:213-229): e4m3 activations [M, K] with per-row fp32 scales x int4 weights in the
    tinygemm tile order (offset-8 codes; scale_and_zero bf16 [K/g, N, 2]) -> bf16 [M, N]; the group scale multiplies fp32 group sums."""
    dev = _require_gpu("fp8_int4_linear", xq, x_scale, qdata, scale_and_zero)
    xq = _fp8_bytes("fp8_int4_linear", xq)
    if xq.dim() != 2 or qdata.dim() != 4 or qdata.dtype != torch.int32 or scale_and_zero.dtype != torch.bfloat16:
        raise RuntimeError("fp8_int4_linear: expected xq [M, K] e4m3, qdata int32 [N/8, K/128, 32, 4], scale_and_zero bf16 [K/g, N, 2]")
    m, k = xq.shape
    n = qdata.shape[0] * 8
    if qdata.shape[1] * 128 != k or tuple(scale_and_zero.shape) != (k // group_size, n, 2):
        raise RuntimeError(f"fp8_int4_linear: shapes do not agree: xq {tuple(xq.shape)}, qdata {tuple(qdata.shape)}, scale_and_zero {tuple(scale_and_zero.shape)}")
    xs = x_scale.reshape(-1).to(torch.float32).contiguous()
    if xs.numel() != m:
        raise RuntimeError("fp8_int4_linear: one activation scale per row expected")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_fp8_int4_linear(_ptr(xq.contiguous()), _ptr(xs), _ptr(qdata.contiguous()), _ptr(scale_and_zero.contiguous()),
                                                 _ptr(bias) if bias is not None else None, _ptr(y), m, n, k, group_size, _stream()))
    return y


def fp8_int4_dynamic_fits(m: int, n: int, k: int) -> bool:
    """Whether the fp8-activation x int4 kernel with the activation cast fused in takes this shape (M <= 16, codes within 64 KiB of LDS)."""
    return bool(_lib.lib().ao_fp8_int4_dynamic_fits(m, n, k))


def fp8_int4_act_linear(x, qdata, scale_and_zero, group_size, bias=None, fused=None):
    """Float8DynamicActivationInt4WeightConfig's F.linear on a 2-D bf16 activation (int4_tensor.py:205-235): per-row e4m3 cast of x, then
    mslk.f8i4bf16_rowwise's contract.  One row runs cast + matmul in ONE launch (round 4: the stand-alone cast cost this path 28 %);
    everything else ao_fp8_quantize_rowwise + ao_fp8_int4_linear.  Same bits either way."""
    dev = _require_gpu("fp8_int4_act_linear", x, qdata, scale_and_zero)
    if x.dim() != 2 or x.dtype != torch.bfloat16:
        raise RuntimeError(f"fp8_int4_act_linear: x must be a 2-D bfloat16 tensor, got {tuple(x.shape)} {x.dtype}")
    m, k = x.shape
    n = qdata.shape[0] * 8
    # (round 6, tools/bench_fp8_int4.py --batch 1 .. 16, the 160 linears of a Llama-3-8B token, profiles/f3_fused_vs_cast_r06.jsonl: the
    # fused launch is 1.23 x ahead of cast + matmul at one row, level at two, and 1.17 / 1.32 / 1.72 x BEHIND at 3 / 4 / 8 rows -- every
    # workgroup casts all rows itself -- so it serves one row only)
    # `fused`: None = that rule; True = the one-launch form wherever the kernel takes the shape (tests, A/B)
    if not fp8_int4_dynamic_fits(m, n, k) or not (fused if fused is not None else m == 1):
        xq, xs = fp8_quantize_rowwise(x.contiguous())
        return fp8_int4_linear(xq, xs, qdata, scale_and_zero, group_size, bias)
    if qdata.dim() != 4 or qdata.dtype != torch.int32 or scale_and_zero.dtype != torch.bfloat16 or qdata.shape[1] * 128 != k \
            or tuple(scale_and_zero.shape) != (k // group_size, n, 2):
        raise RuntimeError(f"fp8_int4_act_linear: shapes do not agree: x {tuple(x.shape)}, qdata {tuple(qdata.shape)}, scale_and_zero {tuple(scale_and_zero.shape)}")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_fp8_int4_dynamic_linear(_ptr(x.contiguous()), _ptr(qdata.contiguous()), _ptr(scale_and_zero.contiguous()),
                                                         _ptr(bias) if bias is not None else None, _ptr(y), m, n, k, group_size, _stream()))
    return y


def generate_permute_indices(tokens_per_expert_group, experts_per_rank, num_ranks, max_len, alignment):
    """torchao.prototype.moe_training.ep.kernels.generate_permute_indices (kernels.py:132-214): expert-major gather indices with every
    expert's group padded to `alignment` rows.  Returns (permuted_indices int32 [max_len] with -1 for padding, m_sizes int32 [E],
    m_offsets int32 [E])."""
    dev = _require_gpu("generate_permute_indices", tokens_per_expert_group)
    counts = tokens_per_expert_group.to(torch.int32).contiguous()
    if counts.numel() != experts_per_rank * num_ranks:
        raise ValueError(f"generate_permute_indices: expected {experts_per_rank * num_ranks} counts, got {counts.numel()}")
    idx = torch.empty(max_len, dtype=torch.int32, device=dev)
    start = torch.empty(counts.numel(), dtype=torch.int32, device=dev)
    m_sizes = torch.empty(experts_per_rank, dtype=torch.int32, device=dev)
    m_offsets = torch.empty(experts_per_rank, dtype=torch.int32, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_moe_permute_indices(_ptr(counts), _ptr(start), _ptr(idx), _ptr(m_sizes), _ptr(m_offsets), experts_per_rank,
                                                     num_ranks, max_len, alignment, _stream()))
    return idx, m_sizes, m_offsets


def _rows_u8(t):
    """[rows, row_bytes] byte view of a 2-D tensor of any dtype"""
    t = t.contiguous()
    return t, t.shape[0], t.shape[1] * t.element_size()


def gather_rows(x, indices, num_out=None):
    """out[i] = x[indices[i]] for 0 <= indices[i] < rows(x), zeros otherwise (`vstack(x, 0)[indices]`, ep/permute.py:86-96)."""
    dev = _require_gpu("gather_rows", x, indices)
    if x.dim() != 2 or indices.dim() != 1 or indices.dtype != torch.int32:
        raise ValueError("gather_rows: expected a 2-D tensor and int32 1-D indices")
    x, rows, row_bytes = _rows_u8(x)
    indices = indices.contiguous()
    n_out = indices.numel() if num_out is None else num_out
    out = torch.empty((n_out, x.shape[1]), dtype=x.dtype, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_moe_gather_rows(_ptr(x), _ptr(indices), _ptr(out), rows, n_out, row_bytes, _stream()))
    return out


def scatter_rows(y, indices, num_rows_out):
    """out[indices[i]] = y[i] for 0 <= indices[i] < num_rows_out (`out = empty(rows + 1); out[indices] = y; out[:-1]`,
    ep/unpermute.py:36-41).  Rows no index names are left uninitialised, like the reference's."""
    dev = _require_gpu("scatter_rows", y, indices)
    if y.dim() != 2 or indices.dim() != 1 or indices.dtype != torch.int32 or indices.numel() != y.shape[0]:
        raise ValueError("scatter_rows: expected a 2-D tensor and one int32 index per row")
    y, rows, row_bytes = _rows_u8(y)
    indices = indices.contiguous()
    out = torch.empty((num_rows_out, y.shape[1]), dtype=y.dtype, device=dev)
    with _on(dev):
        _lib.check(_lib.lib().ao_moe_scatter_rows(_ptr(y), _ptr(indices), _ptr(out), rows, num_rows_out, row_bytes, _stream()))
    return out


def fused_unpad_token_groups(inputs, offsets, padded_group_start_offsets, num_tokens, alignment_size=32):
    """torchao::fused_unpad_token_groups (kernels/mxfp8/quant.py:1319-1363; semantics torch_unpad_token_groups,
    quant.py:433-480): gathers the `num_tokens` real rows back out of a padded buffer."""
    dev = _require_gpu("fused_unpad_token_groups", inputs, offsets, padded_group_start_offsets)
    _check_token_groups("fused_unpad_token_groups", inputs, offsets)
    if padded_group_start_offsets.dtype != torch.int32:
        raise AssertionError("padded_group_start_offsets must be int32")
    if padded_group_start_offsets.numel() != offsets.numel():
        raise RuntimeError("fused_unpad_token_groups: offsets and padded_group_start_offsets must have one entry per group")
    inputs = inputs.contiguous()
    dim = inputs.shape[1]
    out = torch.empty((num_tokens, dim), dtype=inputs.dtype, device=dev)
    with _on(dev):
        _lib.check(
            _lib.lib().ao_moe_unpad_token_groups(
                _ptr(inputs), _ptr(offsets.contiguous()), _ptr(padded_group_start_offsets.contiguous()), _ptr(out), num_tokens, dim,
                inputs.element_size(), offsets.numel(), _stream()
            )
        )
    return out


def dynamic_linear_fits(m: int, n: int, k: int) -> bool:
    """Whether the fused cast + matmul kernels take this shape (M <= 16: the cast activation must fit LDS; 16 < M <= 256: weights with
    few output tiles and K % 512 == 0)."""
    return bool(_lib.lib().ao_dyn_linear_fits(m, n, k))


def dynamic_linear_preferred(m: int, n: int, k: int) -> bool:
    """Whether the fused kernel beats cast + matmul: every workgroup (one per 16 output columns) casts the whole activation itself, so
    the redundant work grows with M while the stand-alone cast stays one ~3 us launch.  Round 6 re-measured both forms of the whole linear
    on a bf16 activation (tools/bench_dec8.py --linear --all-shapes, 8 shapes x int8 / fp8 x M = 1 .. 16, cold weights,
    profiles/dyn_vs_two_r06.jsonl): fused wins at M = 1 on every shape (qkv shard 5.0 vs 7.5 us, o 3.7 vs 6.2, gate_up 13.7 vs 15.7, down
    4096 x 14336 15.8 vs 16.9), M = 2 is level (+- 5 %), and from 3 rows on cast + matmul is 1.2 - 1.6 x ahead wherever the rule of
    rounds 2 - 5 (M x N / 16 <= 1024) still took the fused kernel (qkv shard at M = 6: 14.4 vs 9.1 us) -- the decode matmul got faster
    since (full-line ring, 8-row tiles, the LDS bound), the in-kernel cast did not.  16 < M <= 256 (mid8_kernels.hip) is reachable but was
    never preferred (profiles/mid8_sweep_r04.txt)."""
    return m == 1 and dynamic_linear_fits(m, n, k)


def int8_linear(x2: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias=None) -> torch.Tensor:
    """The Int8Tensor dynamic-activation F.linear on a 2-D bf16 activation (int8_tensor.py:266-359): decode sizes take the
    fused cast + matmul kernel, everything else ao_int8_quantize_rowwise + ao_int8_scaled_mm.  Same bits either way."""
    if dynamic_linear_preferred(x2.shape[0], wq.shape[0], x2.shape[1]):
        return int8_dynamic_linear(x2, wq, w_scale, bias)
    xq, xs = int8_quantize_rowwise(x2)
    return int8_scaled_mm(xq, xs, wq, w_scale, bias)


def fp8_linear(x2: torch.Tensor, wq: torch.Tensor, w_scale: torch.Tensor, bias=None) -> torch.Tensor:
    """The Float8Tensor rowwise dynamic-activation F.linear on a 2-D bf16 activation (float8_tensor.py:338-469)."""
    if dynamic_linear_preferred(x2.shape[0], wq.shape[0], x2.shape[1]):
        return fp8_dynamic_linear(x2, wq, w_scale, bias)
    xq, xs = fp8_quantize_rowwise(x2)
    return fp8_scaled_mm(xq, wq.t(), xs, w_scale.t(), bias)


def _dynamic_linear(name, entry, x, wq, w_scale, bias, wdtype):
    dev = _require_gpu(name, x, wq, w_scale, bias)
    if x.dim() != 2 or x.dtype != torch.bfloat16:
        raise RuntimeError(f"{name}: x must be a 2-D bfloat16 tensor, got {tuple(x.shape)} {x.dtype}")
    if wq.dim() != 2 or wq.dtype != wdtype:
        raise RuntimeError(f"{name}: weight must be a 2-D {wdtype} tensor [N, K]")
    x = x.contiguous()
    wq = wq.contiguous()
    m, k = x.shape
    n, k2 = wq.shape
    if k != k2:
        raise RuntimeError(f"{name}: K mismatch {k} vs {k2}")
    w_scale = w_scale.reshape(-1).to(torch.float32).contiguous()
    if w_scale.numel() != n:
        raise RuntimeError(f"{name}: weight scale must be per-row ([N])")
    if bias is not None:
        bias = bias.to(torch.bfloat16).contiguous()
        if bias.numel() != n:
            raise RuntimeError(f"{name}: bias must have N elements")
    y = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    with _on(dev):
        _lib.check(entry(_ptr(x), _ptr(wq.view(torch.uint8) if wdtype != torch.int8 else wq), _ptr(w_scale), _ptr(bias), _ptr(y), m, n, k, _stream()))
    return y


def int8_dynamic_linear(x, wq, w_scale, bias=None):
    """Int8Tensor F.linear with dynamic per-row activation quantisation in ONE launch (int8_tensor.py:176-248, 305-359): same
    bits as int8_quantize_rowwise + int8_scaled_mm.  Shapes accepted: dynamic_linear_fits(M, N, K)."""
    return _dynamic_linear("int8_dynamic_linear", _lib.lib().ao_int8_dynamic_linear, x, wq, w_scale, bias, torch.int8)


def fp8_dynamic_linear(x, wq, w_scale, bias=None):
    """Float8Tensor F.linear with dynamic per-row e4m3 activation quantisation in ONE launch (float8_tensor.py:167-253,
    float8/inference.py:104-123): same bits as fp8_quantize_rowwise + fp8_scaled_mm at these sizes."""
    return _dynamic_linear("fp8_dynamic_linear", _lib.lib().ao_fp8_dynamic_linear, x, wq, w_scale, bias, torch.float8_e4m3fn)
