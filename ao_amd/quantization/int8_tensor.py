"""Int8Tensor: int8 weight with dynamic int8 activations, MI355X-native.

Host-side mirror of torchao/quantization/quantize_/workflows/int8/int8_tensor.py (same attribute names, from_hp / linear /
slice semantics for the path SURVEY.md 8(a7, a8) scopes): PerRow (default) or PerTensor symmetric weights; dynamic
activations PerRow / PerTensor symmetric, or PerRow ASYMMETRIC with the zero-point correction of :318-331.  Arithmetic: the
HIP kernels behind ao_amd.ops (ao_int8_quantize_rowwise[_amax|_asym], ao_int8_scaled_mm, ao_int8_int_mm +
ao_int8_scale_epilogue_asym).
"""
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import ops
from .base_tensor import LowBitTensorBase, aten
from .granularity import Granularity, PerRow, PerTensor
from .quant_primitives import MappingType

__all__ = ["Int8Tensor", "QuantizeTensorToInt8Kwargs"]


def _require_bf16_activation(x, what):
    """The reference quantizes the activation in ITS dtype (fp16 scales are upcast on purpose, int8_tensor.py:311-317); the fused MI355X
    casts take bfloat16 only, and a silent .to(bfloat16) would round fp16 / fp32 activations before the scale is taken -- not the
    reference's arithmetic.  The default (PerRow, symmetric, dynamic) linear has a correct slow path for fp16 / fp32
    (`_linear_other_dtype`); the other variants refuse instead of approximating."""
    if x.dtype != torch.bfloat16:
        raise NotImplementedError(f"{what} on MI355X takes bfloat16 activations, got {x.dtype}: cast the activation explicitly "
                                  "(x.to(torch.bfloat16)) if that rounding is acceptable")


def _tracing_now(x):
    """fp16 / fp32 activations run raw C-ABI GEMMs that have no fake kernels: while tracing they refuse like before (NotImplementedError
    from _require_bf16_activation) instead of failing on a FakeTensor's data_ptr."""
    from ..torch_ops import tracing

    return tracing(x)


def _linear_other_dtype(x2, w, bias):
    """fp16 / fp32 activations (the reference supports them): Int8Tensor.from_hp's arithmetic in the activation's OWN dtype as
    torch ops on the device -- choose_qparams_affine SYMMETRIC (quant_primitives.py:1534-1562: amax / 127.5, clamped at fp32 eps, fp32
    scale) and quantize_affine (:479-481: round(x * (1 / scale))) -- then this library's int8 x int8 -> int32 GEMM and the reference's
    epilogue (int8_tensor.py:311-357: the row scale applied in fp32, cast to the activation dtype, times the weight scale)."""
    mn, mx = x2.amin(dim=-1, keepdim=True), x2.amax(dim=-1, keepdim=True)
    zero = torch.zeros_like(mn)
    amax = torch.max(-torch.min(mn, zero), torch.max(mx, zero))
    scale = torch.clamp(amax / 127.5, min=torch.finfo(torch.float32).eps).to(torch.float32)
    q = torch.clamp(torch.round(x2 * (1.0 / scale)), -128, 127).to(torch.int8)
    c = ops.int_mm(q, w.qdata.t())
    y = (c.to(torch.float32) * scale).to(x2.dtype)
    y = y * w._row_scale().flatten()
    if bias is not None:
        y = y + bias
    return y.to(x2.dtype)


@dataclass
class QuantizeTensorToInt8Kwargs:
    """reference int8_tensor.py:41-56"""

    granularity: Granularity = field(default_factory=PerRow)
    mapping_type: MappingType = MappingType.SYMMETRIC
    reduce_range: bool = False


def _mapping(mapping_type) -> MappingType:
    if isinstance(mapping_type, str):  # round-1 checkpoints stored the lowercase name
        return MappingType[mapping_type.upper()]
    return mapping_type


def _check_granularity(granularity, what):
    if not isinstance(granularity, (PerRow, PerTensor)):
        raise NotImplementedError(
            f"Int8Tensor on MI355X implements PerRow / PerTensor {what} quantization, got {granularity} "
            "(per-group int8 is outside the SURVEY.md section 8 path)"
        )
    if isinstance(granularity, PerRow) and granularity.dim not in (-1,):
        raise NotImplementedError(f"Int8Tensor on MI355X implements PerRow(dim=-1) only, got {granularity}")


class Int8Tensor(LowBitTensorBase):
    """
    Tensor attributes (reference :59-88):
      qdata       int8 [N, K]
      scale       fp32 [N, 1] (PerRow) or [1, 1] (PerTensor)   (values are bf16-representable, computed in bf16 like the
                                                                reference -- oracle A.3)
      zero_point  int8, same shape as scale, or None (symmetric)
    Non-tensor attributes: block_size ([1, K] or [N, K]), dtype (the original hp dtype), act_quant_kwargs (None = weight only).
    """

    tensor_data_names = ["qdata", "scale"]
    tensor_attribute_names = ["block_size", "dtype_", "act_quant_kwargs"]
    # w_row_sums (int32 [N]): rowsum(qdata), the operand of the asymmetric-activation correction (int8_tensor.py:326 recomputes it
    # on every call); kept next to the weight, built at from_hp when the activation mapping is ASYMMETRIC
    optional_tensor_data_names = ["act_pre_scale", "zero_point", "w_row_sums", "act_quant_scale", "act_quant_zero_point"]

    def __new__(cls, qdata, scale, block_size, dtype_, act_quant_kwargs=None, act_pre_scale=None, zero_point=None, w_row_sums=None,
                act_quant_scale=None, act_quant_zero_point=None):
        kwargs = dict(device=qdata.device, dtype=dtype_, requires_grad=False)
        return torch.Tensor._make_wrapper_subclass(cls, qdata.shape, **kwargs)

    def __init__(self, qdata, scale, block_size, dtype_, act_quant_kwargs=None, act_pre_scale=None, zero_point=None, w_row_sums=None,
                 act_quant_scale=None, act_quant_zero_point=None):
        self.qdata = qdata
        self.scale = scale
        self.block_size = list(block_size)
        self.dtype_ = dtype_
        self.act_quant_kwargs = act_quant_kwargs
        self.act_pre_scale = act_pre_scale
        self.zero_point = zero_point
        self.w_row_sums = w_row_sums
        self.act_quant_scale = act_quant_scale              # static activation quantization: given, not measured (reference :80-84)
        self.act_quant_zero_point = act_quant_zero_point

    def _quantization_type(self):
        return (f"act_quant_kwargs={self.act_quant_kwargs}, block_size={self.block_size}, "
                f"shape={tuple(self.shape)}, device={self.device}, dtype={self.dtype}")

    @classmethod
    def from_hp(cls, hp_tensor: torch.Tensor, granularity: Granularity = None, mapping_type=MappingType.SYMMETRIC,
                act_quant_kwargs: Optional[QuantizeTensorToInt8Kwargs] = None, act_quant_scale: Optional[torch.Tensor] = None,
                act_quant_zero_point: Optional[torch.Tensor] = None):
        """reference from_hp (:176-248).  SYMMETRIC: scale = amax / 127.5 clamped at fp32 eps, over the row or the whole
        tensor; ASYMMETRIC (PerRow): scale = (max - min) / 255, integer zero-point."""
        granularity = PerRow() if granularity is None else granularity
        mapping_type = _mapping(mapping_type)
        _check_granularity(granularity, "tensor")
        if hp_tensor.dtype != torch.bfloat16:
            raise NotImplementedError(f"Int8Tensor.from_hp on MI355X takes bfloat16, got {hp_tensor.dtype}")
        if hp_tensor.dim() == 3:
            # per-expert (MoE) weight: every expert quantized on its own, qdata [E, N, K], scale [E, N, 1] (PerRow) / [E, 1, 1];
            # aten.select.int(0, e) (reference :492-517) hands expert e to F.linear
            if act_quant_scale is not None or mapping_type != MappingType.SYMMETRIC:
                raise NotImplementedError("3-D Int8Tensor weights on MI355X: symmetric, dynamic or weight-only")
            per = [cls.from_hp(hp_tensor[i], granularity, mapping_type, act_quant_kwargs) for i in range(hp_tensor.shape[0])]
            sums = None if per[0].w_row_sums is None else torch.stack([p.w_row_sums for p in per])
            return cls(torch.stack([p.qdata for p in per]), torch.stack([p.scale for p in per]), [1] + per[0].block_size, hp_tensor.dtype,
                       act_quant_kwargs=act_quant_kwargs, w_row_sums=sums)
        if hp_tensor.dim() != 2:
            raise NotImplementedError("Int8Tensor.from_hp on MI355X takes 2-D tensors or 3-D [experts, N, K] weights")
        x = hp_tensor.contiguous()
        zero_point = None
        if mapping_type == MappingType.ASYMMETRIC:
            if not isinstance(granularity, PerRow):
                raise NotImplementedError("Int8Tensor on MI355X implements ASYMMETRIC quantization per row only")
            qdata, scale, zero_point = ops.int8_quantize_rowwise_asym(x)
        elif mapping_type != MappingType.SYMMETRIC:
            raise NotImplementedError(f"Int8Tensor on MI355X implements SYMMETRIC / ASYMMETRIC mapping, got {mapping_type}")
        elif isinstance(granularity, PerTensor):
            qdata, scale = ops.int8_quantize_tensorwise(x)
        else:
            qdata, scale = ops.int8_quantize_rowwise(x)
        block_size = list(hp_tensor.shape) if isinstance(granularity, PerTensor) else [1, hp_tensor.shape[-1]]
        w_row_sums = None
        if act_quant_kwargs is not None and (_mapping(act_quant_kwargs.mapping_type) == MappingType.ASYMMETRIC or act_quant_zero_point is not None):
            w_row_sums = ops.int8_row_sums(qdata)
        return cls(qdata, scale, block_size, hp_tensor.dtype, act_quant_kwargs=act_quant_kwargs, zero_point=zero_point, w_row_sums=w_row_sums,
                   act_quant_scale=act_quant_scale, act_quant_zero_point=act_quant_zero_point)

    def dequantize(self, output_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """reference :250-263: (qdata - zero_point) * scale in fp32, then cast"""
        q = self.qdata.to(torch.float32)
        if self.zero_point is not None:
            q = q - self.zero_point.to(torch.float32)
        return (q * self.scale.to(torch.float32)).to(output_dtype or self.dtype)

    def _row_scale(self) -> torch.Tensor:
        """fp32 [N]: the per-row view of the scale (a PerTensor scale is broadcast)"""
        return self.scale.reshape(-1).expand(self.qdata.shape[0]) if self.scale.numel() == 1 else self.scale.reshape(-1)


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
    act = w.act_quant_kwargs
    _check_granularity(act.granularity, "activation")
    if w.zero_point is not None:
        raise NotImplementedError("Int8Tensor linear on MI355X takes symmetric weights (asymmetric is an ACTIVATION option in the reference)")
    assert w.qdata.dim() == 2, "F.linear takes a 2-D weight: select an expert of a 3-D weight first (weight[e])"
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    n = w.qdata.shape[0]
    if (x.dtype in (torch.float16, torch.float32) and w.act_quant_scale is None and _mapping(act.mapping_type) == MappingType.SYMMETRIC
            and isinstance(act.granularity, PerRow) and x2.shape[0] > 0 and not _tracing_now(x2)):
        return _linear_other_dtype(x2, w, bias).reshape(*x.shape[:-1], n).to(out_dtype)
    _require_bf16_activation(x, "Int8Tensor dynamic-activation linear")
    if x2.shape[0] == 0:
        y = x2.new_zeros((0, n))
    else:
        from ..torch_ops import kernels  # dispatcher ops (with fake kernels) while tracing, the direct C-ABI calls otherwise
        k = kernels(x2)
        if w.act_quant_scale is not None:  # static: the activation qparams were calibrated, not measured per call
            zp = w.act_quant_zero_point
            if zp is not None and w.w_row_sums is None:
                w.w_row_sums = ops.int8_row_sums(w.qdata)
            y = k.int8_linear_static(x2, w.qdata, w._row_scale(), w.act_quant_scale, zp, w.w_row_sums, bias)
        elif _mapping(act.mapping_type) == MappingType.ASYMMETRIC:
            if not isinstance(act.granularity, PerRow):
                raise NotImplementedError("Int8Tensor on MI355X implements ASYMMETRIC activation quantization per row only")
            if w.w_row_sums is None:  # a weight built without from_hp (e.g. loaded from a reference checkpoint): once, eagerly
                w.w_row_sums = ops.int8_row_sums(w.qdata)
            y = k.int8_linear_asym(x2, w.qdata, w._row_scale(), w.w_row_sums, bias)
        elif isinstance(act.granularity, PerTensor):
            y = k.int8_linear_tensorwise(x2, w.qdata, w._row_scale(), bias)
        else:
            y = k.int8_linear(x2, w.qdata, w._row_scale(), bias)
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
    per_tensor = self.scale.numel() == 1
    zp = self.zero_point
    if dim == 0:
        q = self.qdata[start:end].contiguous()
        s = self.scale if per_tensor else self.scale[start:end].contiguous()
        zp = zp if (zp is None or per_tensor) else zp[start:end].contiguous()
        sums = None if self.w_row_sums is None else self.w_row_sums[start:end].contiguous()
    else:
        q, s = self.qdata[:, start:end].contiguous(), self.scale
        sums = None if self.w_row_sums is None else ops.int8_row_sums(q)  # a K slice has its own row sums
        if pre is not None and pre.numel() == self.shape[1]:  # per-input-feature pre-scale follows the K slice
            pre = pre.reshape(-1)[start:end]
    block_size = list(q.shape) if per_tensor else [1, q.shape[1]]
    return Int8Tensor(q, s, block_size, self.dtype_, self.act_quant_kwargs, pre, zp, sums, self.act_quant_scale, self.act_quant_zero_point)


@implements(aten.select.int)
def _(func, types, args, kwargs):
    """reference :492-517: expert selection on a 3-D (MoE) weight"""
    self, dim, index = args
    assert dim == 0, f"Int8Tensor aten.select.int with {dim=} is not yet supported"
    assert len(self.qdata.shape) == len(self.scale.shape), "unsupported"
    assert len(self.qdata.shape) == len(self.block_size), "unsupported"
    zp = None if self.zero_point is None else self.zero_point[index]
    sums = None if self.w_row_sums is None else self.w_row_sums[index]
    return Int8Tensor(self.qdata[index], self.scale[index], self.block_size[1:], self.dtype_, self.act_quant_kwargs, self.act_pre_scale, zp, sums,
                      self.act_quant_scale, self.act_quant_zero_point)


torch.serialization.add_safe_globals([Int8Tensor, QuantizeTensorToInt8Kwargs, MappingType])
