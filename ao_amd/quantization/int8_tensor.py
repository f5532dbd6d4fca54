"""Int8Tensor: int8 per-row weight with dynamic per-token int8 activations, MI355X-native.

Host-side mirror of torchao/quantization/quantize_/workflows/int8/int8_tensor.py
(same attribute names, from_hp / linear / slice semantics for the path SURVEY.md 8(a7, a8) scopes:
PerRow symmetric weight, optional PerRow symmetric dynamic activation).  Arithmetic: the HIP
kernels behind ao_amd.ops (ao_int8_quantize_rowwise, ao_int8_scaled_mm).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import ops
from .base_tensor import LowBitTensorBase, aten
from .granularity import Granularity, PerRow

__all__ = ["Int8Tensor", "QuantizeTensorToInt8Kwargs"]


@dataclass
class QuantizeTensorToInt8Kwargs:
    """reference int8_tensor.py:41-56 (only the PerRow / symmetric defaults are implemented)"""

    granularity: Granularity = field(default_factory=PerRow)
    mapping_type: str = "symmetric"
    reduce_range: bool = False


def _check_per_row(granularity, what):
    if not isinstance(granularity, PerRow):
        raise NotImplementedError(
            f"Int8Tensor on MI355X implements PerRow {what} quantization only, got {granularity} "
            "(per-tensor / per-group are outside the SURVEY.md section 8 path)"
        )


class Int8Tensor(LowBitTensorBase):
    """
    Tensor attributes (reference :59-88):
      qdata  int8 [N, K]
      scale  fp32 [N, 1]   (reference keeps the scale in the hp dtype widened on use; values are
                            bf16-representable, computed in bf16 like the reference -- oracle A.3)
    Non-tensor attributes: block_size ([1, K]), dtype (the original hp dtype),
    act_quant_kwargs (None = weight only).
    """

    tensor_data_names = ["qdata", "scale"]
    tensor_attribute_names = ["block_size", "dtype_", "act_quant_kwargs"]
    optional_tensor_data_names = ["act_pre_scale"]

    def __new__(cls, qdata, scale, block_size, dtype_, act_quant_kwargs=None, act_pre_scale=None):
        kwargs = dict(device=qdata.device, dtype=dtype_, requires_grad=False)
        return torch.Tensor._make_wrapper_subclass(cls, qdata.shape, **kwargs)

    def __init__(self, qdata, scale, block_size, dtype_, act_quant_kwargs=None, act_pre_scale=None):
        self.qdata = qdata
        self.scale = scale
        self.block_size = list(block_size)
        self.dtype_ = dtype_
        self.act_quant_kwargs = act_quant_kwargs
        self.act_pre_scale = act_pre_scale

    def _quantization_type(self):
        return (f"act_quant_kwargs={self.act_quant_kwargs}, block_size={self.block_size}, "
                f"shape={tuple(self.shape)}, device={self.device}, dtype={self.dtype}")

    @classmethod
    def from_hp(cls, hp_tensor: torch.Tensor, granularity: Granularity = None,
                act_quant_kwargs: Optional[QuantizeTensorToInt8Kwargs] = None):
        """reference from_hp (:176-248): symmetric, scale = amax / 127.5 clamped at fp32 eps."""
        granularity = PerRow() if granularity is None else granularity
        _check_per_row(granularity, "weight")
        if hp_tensor.dtype != torch.bfloat16:
            raise NotImplementedError(f"Int8Tensor.from_hp on MI355X takes bfloat16, got {hp_tensor.dtype}")
        if hp_tensor.dim() != 2:
            raise NotImplementedError("Int8Tensor.from_hp on MI355X takes 2-D tensors")
        qdata, scale = ops.int8_quantize_rowwise(hp_tensor.contiguous())
        return cls(qdata, scale, [1, hp_tensor.shape[-1]], hp_tensor.dtype, act_quant_kwargs=act_quant_kwargs)

    def dequantize(self, output_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """reference :250-263: qdata * scale in fp32, then cast"""
        return (self.qdata.to(torch.float32) * self.scale.to(torch.float32)).to(output_dtype or self.dtype)


implements = Int8Tensor.implements
implements_torch_function = Int8Tensor.implements_torch_function


@implements(aten.linear.default)
@implements_torch_function(F.linear)
def _(func, types, args, kwargs):
    """reference :266-359 (dynamic-activation branch); the activation cast and the GEMM with its
    two-stage scale epilogue are one HIP launch each."""
    x, w = args[0], args[1]
    bias = args[2] if len(args) > 2 else kwargs.get("bias", None)
    assert isinstance(w, Int8Tensor), f"Expected weight to be Int8Tensor, got {type(w)}"
    out_dtype = x.dtype
    if w.act_pre_scale is not None:
        x = x * w.act_pre_scale
    if w.act_quant_kwargs is None:
        raise NotImplementedError(
            "Int8Tensor weight-only linear is not on the MI355X hot path (SURVEY.md section 8): "
            "use Int8DynamicActivationInt8WeightConfig"
        )
    _check_per_row(w.act_quant_kwargs.granularity, "activation")
    x2 = x.reshape(-1, x.shape[-1]).to(torch.bfloat16).contiguous()
    n = w.qdata.shape[0]
    if x2.shape[0] == 0:
        y = x2.new_zeros((0, n))
    else:
        from ..torch_ops import kernels  # dispatcher ops (with fake kernels) while tracing, the direct C-ABI calls otherwise
        y = kernels(x2).int8_linear(x2, w.qdata, w.scale, bias)
        bias = None
    y = y.reshape(*x.shape[:-1], n)
    if bias is not None:
        y = y + bias.to(y.dtype)
    return y.to(out_dtype)


@implements(aten.slice.Tensor)
def _(func, types, args, kwargs):
    """reference :362-422: rows slice qdata and scale, columns slice qdata only (the per-row scale
    was computed over the full K, as a K-sharded TP linear needs -- SURVEY.md 8(e))"""
    self, dim = args[0], args[1] if len(args) > 1 else 0
    start = args[2] if len(args) > 2 and args[2] is not None else 0
    end = args[3] if len(args) > 3 and args[3] is not None else self.shape[dim]
    step = args[4] if len(args) > 4 else 1
    assert step == 1 and dim in (0, 1)
    end = min(end, self.shape[dim])
    pre = self.act_pre_scale
    if dim == 0:
        q, s = self.qdata[start:end].contiguous(), self.scale[start:end].contiguous()
    else:
        q, s = self.qdata[:, start:end].contiguous(), self.scale
        if pre is not None and pre.numel() == self.shape[1]:  # per-input-feature pre-scale follows the K slice
            pre = pre.reshape(-1)[start:end]
    return Int8Tensor(q, s, [1, q.shape[1]], self.dtype_, self.act_quant_kwargs, pre)


torch.serialization.add_safe_globals([Int8Tensor, QuantizeTensorToInt8Kwargs])
