"""Torch-facing wrappers over the C-ABI, one per reference op on the hot path.

Same argument meaning, output allocation ("allocate and return", inputs
borrowed) and error behaviour as the ops they replace; each docstring cites the
reference call site.  Tensors must live on the GPU ("cuda" == HIP device on
ROCm); there is no CPU fallback.
"""
import torch

from . import _lib

__all__ = [
    "convert_weight_to_int4pack",
    "unpack_int4pack",
    "weight_int4pack_mm",
    "int4_dequantize",
    "int4_quantize_tinygemm",
]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return t.data_ptr() if t is not None else None


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
    with torch.cuda.device(dev):
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
    with torch.cuda.device(dev):
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
    with torch.cuda.device(dev):
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
    with torch.cuda.device(dev):
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
    with torch.cuda.device(dev):
        _lib.check(
            _lib.lib().ao_int4_quantize_tinygemm(_ptr(w), _ptr(qdata), _ptr(sz), n, k, group_size, _stream())
        )
    return qdata, sz
