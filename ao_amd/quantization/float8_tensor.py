"""Float8Tensor: float8 e4m3 weight (PerRow or PerTensor scales) with dynamic activations of the same granularity, MI355X-native.

Host-side mirror of torchao/quantization/quantize_/workflows/float8/float8_tensor.py for the branch
SURVEY.md 8(a9, a10) scopes -- PerRow, KernelPreference TORCH/AUTO on AMD, i.e. what
`_float8_addmm_impl` (:338-469) sends to aten::_scaled_mm through
torchao/float8/inference.py:86-123.  gfx950 uses OCP e4m3fn (max 448), as the reference selects for
MI350 (torchao/float8/config.py:80-83).  Arithmetic: ao_fp8_quantize_rowwise, ao_fp8_scaled_mm.
"""
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn.functional as F

from .. import ops
from .base_tensor import LowBitTensorBase, aten
from .granularity import Granularity, PerRow, PerTensor

__all__ = ["Float8Tensor", "QuantizeTensorToFloat8Kwargs"]


def _require_bf16_activation(x, what):
    """The reference quantizes the activation in ITS dtype (fp16 scales are upcast on purpose, int8_tensor.py:311-317); the fused MI355X
    casts take bfloat16 only, and a silent .to(bfloat16) would round fp16 / fp32 activations before the scale is taken -- not the
    reference's arithmetic.  The default PerRow dynamic linear has a correct slow path for fp16 / fp32 (`_linear_other_dtype`); the
    variants that do not (static / bounded / tensorwise / grouped) refuse instead of approximating."""
    if x.dtype != torch.bfloat16:
        raise NotImplementedError(f"{what} on MI355X takes bfloat16 activations, got {x.dtype}: cast the activation explicitly "
                                  "(x.to(torch.bfloat16)) if that rounding is acceptable")


def _tracing_now(x):
    """fp16 / fp32 activations run raw C-ABI GEMMs that have no fake kernels: while tracing they refuse like before (NotImplementedError
    from _require_bf16_activation) instead of failing on a FakeTensor's data_ptr."""
    from ..torch_ops import tracing

    return tracing(x)


def _linear_other_dtype(x2, w, bias):
    """fp16 / fp32 activations (the reference supports them, float8_tensor.py:349-355 + :167-253): the cast in the activation's
    OWN dtype with the reference's op sequence -- scale = (amax / 448) in x.dtype, widened to fp32 (_choose_scale_float8); codes =
    sat_cast(f32(x) / scale) (_quantize_affine_float8) -- as torch ops on the device, then the raw e4m3 x e4m3 -> fp32 GEMM of this
    library and aten::_scaled_mm's epilogue (acc * scale_a * scale_b + bias) in fp32, out in the activation's dtype.  Three extra
    elementwise passes: the fused bf16 kernels stay the fast path."""
    amax = x2.abs().amax(dim=-1, keepdim=True)
    scale = (amax / 448.0).to(torch.float32)
    xq = (x2.to(torch.float32) / scale).clamp(min=-448.0, max=448.0).to(torch.float8_e4m3fn)
    acc = ops.fp8_mm_f32(xq, w.qdata.t())
    y = acc * scale * w.scale.reshape(1, -1).to(torch.float32)
    if bias is not None:
        y = y + bias.to(torch.float32)
    return y.to(x2.dtype)


@dataclass
class QuantizeTensorToFloat8Kwargs:
    """reference float8_tensor.py:50-81 (PerRow / PerTensor, e4m3fn)"""

    float8_dtype: torch.dtype = torch.float8_e4m3fn
    granularity: Granularity = field(default_factory=PerRow)
    hp_value_lb: Optional[float] = None  # bounds on the amax the activation scale is taken from (reference :64-66)
    hp_value_ub: Optional[float] = None


def _check(granularity, float8_dtype):
    if not isinstance(granularity, (PerRow, PerTensor)):
        raise NotImplementedError(f"Float8Tensor on MI355X implements PerRow / PerTensor, got {granularity} (blockwise scaling is outside SURVEY.md section 8)")
    if isinstance(granularity, PerRow) and granularity.dim != -1:
        raise NotImplementedError(f"Float8Tensor on MI355X implements PerRow(dim=-1) only, got {granularity}")
    if float8_dtype != torch.float8_e4m3fn:
        raise NotImplementedError(f"Float8Tensor on MI355X implements float8_e4m3fn only, got {float8_dtype}")


class Float8Tensor(LowBitTensorBase):
    """
    Tensor attributes (reference :84-140):
      qdata  float8_e4m3fn [N, K]
      scale  fp32 [N, 1]
    Non-tensor attributes: block_size ([1, K]), dtype (original hp dtype), act_quant_kwargs.
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
    def from_hp(cls, hp_tensor: torch.Tensor, float8_dtype: torch.dtype = torch.float8_e4m3fn,
                granularity: Granularity = None, act_quant_kwargs: Optional[QuantizeTensorToFloat8Kwargs] = None):
        """reference from_hp (:167-253), kernel_choice "torch": scale = amax / 448 in the input dtype
        (no eps clamp: an all-zero row gives scale 0 and NaN data, like the reference), data =
        saturating e4m3 cast of x / scale."""
        granularity = PerRow() if granularity is None else granularity
        _check(granularity, float8_dtype)
        if hp_tensor.dtype != torch.bfloat16:
            # reference quant_api.py:1211-1216: PerRow quantization only works for bfloat16 precision; the MI355X kernels take
            # bfloat16 for PerTensor too
            raise AssertionError("PerRow quantization only works for bfloat16 precision input weight")
        if hp_tensor.dim() not in (2, 3):
            raise NotImplementedError("Float8Tensor.from_hp on MI355X takes 2-D weights or 3-D [E, N, K] expert weights")
        k = hp_tensor.shape[-1]
        rows = hp_tensor.contiguous().reshape(-1, k)
        if isinstance(granularity, PerTensor):  # one scale for the whole tensor (block_size = shape)
            qdata, scale = ops.fp8_quantize_tensorwise(rows)
            return cls(qdata.reshape(hp_tensor.shape), scale.reshape([1] * hp_tensor.dim()), list(hp_tensor.shape), hp_tensor.dtype,
                       act_quant_kwargs=act_quant_kwargs)
        qdata, scale = ops.fp8_quantize_rowwise(rows)  # PerRow: one scale per row of every expert
        qdata, scale = qdata.reshape(hp_tensor.shape), scale.reshape(*hp_tensor.shape[:-1], 1)
        return cls(qdata, scale, [1] * (hp_tensor.dim() - 1) + [k], hp_tensor.dtype, act_quant_kwargs=act_quant_kwargs)

    def dequantize(self, output_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """reference :255-275: fp8 -> fp32 * scale, then cast"""
        return (self.qdata.to(torch.float32) * self.scale.to(torch.float32)).to(output_dtype or self.dtype)


implements = Float8Tensor.implements
implements_torch_function = Float8Tensor.implements_torch_function


@implements(aten.linear.default)
@implements_torch_function(F.linear)
def _(func, types, args, kwargs):
    return _float8_linear(args[0], args[1], args[2] if len(args) > 2 else kwargs.get("bias", None))


def _as_weight(w_t):
    """mm / matmul / addmm_ receive the weight as the reference's callers pass it: the TRANSPOSED view [K, N] of a Float8Tensor
    quantized along K (`weight.t()`); the kernels want [N, K] K-contiguous, which is that view transposed back (no copy)."""
    assert isinstance(w_t, Float8Tensor) and w_t.qdata.dim() == 2, "expected a 2-D Float8Tensor operand"
    assert w_t.qdata.stride(0) == 1 or w_t.qdata.shape[0] == 1 or w_t.qdata.shape[1] == 1, (
        "Float8Tensor mm on MI355X takes the transposed view of an [N, K] weight (K-contiguous), as _float8_addmm_impl's callers pass it")
    return w_t.transpose(0, 1)


@implements(aten.t.default)
def _(func, types, args, kwargs):
    """reference :863-885 (aten.t): transpose(0, 1) of a 2-D tensor"""
    self = args[0]
    assert self.qdata.dim() == 2
    bs = list(self.block_size)
    return Float8Tensor(self.qdata.t(), self.scale.t(), [bs[1], bs[0]], self.dtype_, self.act_quant_kwargs, self.act_pre_scale)


@implements(aten.mm.default)
@implements_torch_function(torch.mm)
def _(func, types, args, kwargs):
    """reference :296-300: _float8_addmm_impl(input, weight_t)"""
    return _float8_linear(args[0], _as_weight(args[1]), None)


@implements(aten.matmul.default)
@implements_torch_function(torch.matmul)
def _(func, types, args, kwargs):
    """reference :289-293"""
    return _float8_linear(args[0], _as_weight(args[1]), None)


@implements(aten.addmm_.default)
def _(func, types, args, kwargs):
    """reference :303-314: bias.add_(input @ weight_t), alpha = beta = 1 only"""
    bias_tensor, x, w_t = args[0], args[1], args[2]
    assert kwargs.get("alpha", 1) == 1, "only alpha=1 is supported"
    assert kwargs.get("beta", 1) == 1, "only beta=1 is supported"
    return bias_tensor.add_(_float8_linear(x, _as_weight(w_t), None))


@implements(aten.cat.default)
def _(func, types, args, kwargs):
    """reference :790-846 (merged-weight loaders: q / k / v or gate / up quantized separately, concatenated along N): along a
    dimension whose block size is 1 the scales are concatenated too, along a blocked dimension they must be equal."""
    tensors = args[0]
    dim = (args[1] if len(args) > 1 else kwargs.get("dim", 0)) % tensors[0].dim()
    t0 = tensors[0]
    for t in tensors[1:]:
        assert t0.qdata.dim() == t.qdata.dim() and t0.scale.dim() == t.scale.dim()
        assert list(t0.block_size) == list(t.block_size) and t0.act_quant_kwargs == t.act_quant_kwargs
    qdata = torch.cat([t.qdata for t in tensors], dim=dim)
    if t0.block_size[dim] == 1:
        scale = torch.cat([t.scale for t in tensors], dim=dim)
    else:
        for t in tensors[1:]:
            assert torch.equal(t0.scale, t.scale)
        scale = t0.scale
    block_size = [qdata.shape[i] // scale.shape[i] for i in range(qdata.dim())]
    return Float8Tensor(qdata, scale, block_size, t0.dtype_, t0.act_quant_kwargs, t0.act_pre_scale)


def _like(self, qdata, scale):
    """A Float8Tensor over (qdata, scale) with this one's kwargs; block sizes follow from the two shapes."""
    bs = [qdata.shape[i] // scale.shape[i] for i in range(qdata.dim())]
    return Float8Tensor(qdata, scale, bs, self.dtype_, self.act_quant_kwargs, self.act_pre_scale)


@implements(aten.view.default)
def _(func, types, args, kwargs):
    """reference :863-906: 3-D <-> 2-D with the last dimension kept (MoE experts flattened / restored), or a same-rank view with
    matching (or -1) sizes; the scale is reshaped block for block."""
    self, size = args[0], list(args[1])
    shape = list(self.shape)
    if len(shape) == 3 and len(size) == 2:
        assert shape[-1] == size[-1], f"Only support reshaping when last dimension matches, requested: reshaping from {shape} to {size}"
        return _like(self, self.qdata.reshape(*size), self.scale.reshape(-1, self.scale.shape[-1]))
    if len(shape) == 2 and len(size) == 3:
        assert shape[-1] == size[-1], f"Only support reshaping when last dimension matches, requested: reshaping from {shape} to {size}"
        q = self.qdata.reshape(*size)
        bs = [1, self.block_size[0], self.block_size[1]]
        return _like(self, q, self.scale.reshape(*[q.shape[i] // bs[i] for i in range(3)]))
    assert len(shape) == len(size) and all(x == y or y == -1 for x, y in zip(shape, size)), (
        f"Only support viewing with match dimensions or -1, got: {shape}, {size}")
    return _like(self, self.qdata.reshape(*size), self.scale)


@implements(aten.squeeze.dim)
def _(func, types, args, kwargs):
    """reference :909-928"""
    self, dim = args[0], args[1]
    assert dim == 0, f"Only dim == 0 is supported, got: {dim}"
    return _like(self, self.qdata.squeeze(dim=dim), self.scale.squeeze(dim=dim))


@implements(aten.unsqueeze.default)
def _(func, types, args, kwargs):
    """reference :953-971"""
    self, dim = args[0], args[1]
    return _like(self, self.qdata.unsqueeze(dim=dim), self.scale.unsqueeze(dim=dim))


@implements(aten.split.Tensor)
def _(func, types, args, kwargs):
    """reference :1013-1078 (torch.chunk / split of a merged weight back into its parts): along a dimension with one scale per element
    the scales are split too, along a blocked dimension every chunk keeps the scale (its block is the chunk)."""
    self, size = args[0], args[1]
    dim = args[2] if len(args) > 2 else kwargs.get("dim", 0)
    assert isinstance(size, int), "unimplemented"
    dim = dim % self.dim()
    qs = torch.split(self.qdata, size, dim)
    if self.scale.shape[dim] == 1 and self.block_size[dim] == self.shape[dim]:
        scales = [self.scale] * len(qs)
    elif self.scale.shape[dim] == self.shape[dim] and self.block_size[dim] == 1:
        scales = torch.split(self.scale, size, dim)
    else:
        raise AssertionError(f"`aten.split.Tensor` with {dim=} and {self.scale.shape=} is not yet implemented")
    return [_like(self, q, sc) for q, sc in zip(qs, scales)]


def _float8_linear(x, w, bias):
    """reference :278-469 -> preprocess_data / preprocess_scale -> _scaled_mm(A row-major,
    B = W.t() column-major, scale_a [M,1], scale_b [1,N], bias, out_dtype, use_fast_accum)."""
    assert isinstance(w, Float8Tensor), f"Expected weight to be Float8Tensor, got {type(w)}"
    out_dtype = x.dtype
    if w.act_pre_scale is not None:
        x = x * w.act_pre_scale
    if w.act_quant_kwargs is None:
        raise NotImplementedError(
            "Float8Tensor weight-only linear is not on the MI355X hot path (SURVEY.md section 8): "
            "use Float8DynamicActivationFloat8WeightConfig"
        )
    _check(w.act_quant_kwargs.granularity, w.act_quant_kwargs.float8_dtype)
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    n = w.qdata.shape[0]
    w_tensorwise = w.scale.numel() == 1
    act0 = w.act_quant_kwargs
    if (x.dtype in (torch.float16, torch.float32) and isinstance(act0.granularity, PerRow) and not w_tensorwise and w.qdata.dim() == 2
            and act0.hp_value_lb is None and act0.hp_value_ub is None and x2.shape[0] > 0 and not _tracing_now(x2)):
        return _linear_other_dtype(x2, w, bias).reshape(*x.shape[:-1], n).to(out_dtype)
    _require_bf16_activation(x, "Float8Tensor dynamic-activation linear")
    if isinstance(w.act_quant_kwargs.granularity, PerTensor) != w_tensorwise:
        # reference quant_api.py:1123: "Currently both quantizations need to be the same type"
        raise NotImplementedError("Float8Tensor linear: activation and weight granularities must both be PerRow or both PerTensor")
    if x2.shape[0] == 0:
        y = x2.new_zeros((0, n))
    else:
        from ..torch_ops import kernels  # dispatcher ops (with fake kernels) while tracing, the direct C-ABI calls otherwise
        # tensorwise-scaled _scaled_mm (float8/inference.py:68-123 with [1, 1] scales) is the rowwise epilogue with the two
        # scalars broadcast over rows / columns
        act = w.act_quant_kwargs
        if act.hp_value_lb is not None or act.hp_value_ub is not None:
            lb = -1.0 if act.hp_value_lb is None else float(act.hp_value_lb)
            ub = -1.0 if act.hp_value_ub is None else float(act.hp_value_ub)
            assert lb >= 0 or act.hp_value_lb is None, "hp_value_lb bounds an absolute value: it cannot be negative"
            y = kernels(x2).fp8_linear_clamped(x2, w.qdata, w.scale, bias, lb, ub, w_tensorwise)
        else:
            y = (kernels(x2).fp8_linear_tensorwise if w_tensorwise else kernels(x2).fp8_linear)(x2, w.qdata, w.scale, bias)
        bias = None
    y = y.reshape(*x.shape[:-1], n)
    if bias is not None:
        y = y + bias.to(y.dtype)
    return y.to(out_dtype)


@implements(aten.slice.Tensor)
def _(func, types, args, kwargs):
    """reference :732-790: rows slice qdata and scale; columns slice qdata only (rowwise scales are
    over the full K -- what a row-parallel TP shard needs)"""
    self, dim = args[0], args[1] if len(args) > 1 else 0
    start = args[2] if len(args) > 2 and args[2] is not None else 0
    end = args[3] if len(args) > 3 and args[3] is not None else self.shape[dim]
    step = args[4] if len(args) > 4 else 1
    assert step == 1 and dim in (0, 1)
    end = min(end, self.shape[dim])
    pre = self.act_pre_scale
    per_tensor = self.scale.numel() == 1
    if dim == 0:
        q = self.qdata[start:end].contiguous()
        s = self.scale if per_tensor else self.scale[start:end].contiguous()
    else:
        q, s = self.qdata[:, start:end].contiguous(), self.scale
        if pre is not None and pre.numel() == self.shape[1]:  # per-input-feature pre-scale follows the K slice
            pre = pre.reshape(-1)[start:end]
    return Float8Tensor(q, s, list(q.shape) if per_tensor else [1, q.shape[1]], self.dtype_, self.act_quant_kwargs, pre)


@implements(aten.transpose.int)
def _(func, types, args, kwargs):
    """reference :970-1000: a view with qdata / scale transposed and block_size swapped (MoE callers pass w.transpose(-2, -1))"""
    self, d0, d1 = args[0], args[1], args[2]
    bs = list(self.block_size)
    bs[d0], bs[d1] = bs[d1], bs[d0]
    return Float8Tensor(self.qdata.transpose(d0, d1), self.scale.transpose(d0, d1), bs, self.dtype_, self.act_quant_kwargs, self.act_pre_scale)


@implements(aten._grouped_mm.default)
def _(func, types, args, kwargs):
    """reference float8_grouped_mm (:1085-1122): mat_b is the transposed view [E, K, N] of expert weights quantized PerRow along K;
    the token rows are cast per row to e4m3 and every group multiplies its expert: scaled_grouped_mm with RowWise scales."""
    mat_a, mat_b = args[0], args[1]
    offs = args[2] if len(args) > 2 else kwargs.get("offs", None)
    assert isinstance(mat_b, Float8Tensor)
    assert offs is not None, "offs is required for _grouped_mm"
    is_b_transposed = mat_b.qdata.stride(-2) < mat_b.qdata.stride(-1)
    assert is_b_transposed and mat_b.qdata.dim() == 3 and mat_a.dim() == 2, "unsupported"
    output_dtype = mat_a.dtype
    wq = mat_b.qdata.transpose(-2, -1)  # [E, N, K], K-contiguous: no copy for the reference's layout
    ws = mat_b.scale.transpose(-2, -1)  # [E, N, 1]
    if mat_b.act_quant_kwargs is None:
        raise NotImplementedError("weight-only Float8Tensor _grouped_mm is not on the MI355X hot path: use dynamic activation quantization")
    _check(mat_b.act_quant_kwargs.granularity, mat_b.act_quant_kwargs.float8_dtype)
    # PerRow only, like the reference (float8_tensor.py:1098-1101 asserts rowwise scales on both operands)
    if not isinstance(mat_b.act_quant_kwargs.granularity, PerRow) or ws.shape[-2] != wq.shape[-2]:
        raise NotImplementedError("Float8Tensor _grouped_mm implements PerRow activations and PerRow weight scales only "
                                  f"(got activations {mat_b.act_quant_kwargs.granularity}, weight scale {tuple(mat_b.scale.shape)})")
    _require_bf16_activation(mat_a, "Float8Tensor _grouped_mm")
    aq, a_s = ops.fp8_quantize_rowwise(mat_a.contiguous())
    return ops.fp8_grouped_mm(aq, a_s, wq, ws, offs.to(torch.int32)).to(output_dtype)


@implements(aten.select.int)
def _(func, types, args, kwargs):
    """reference float8_tensor.py:936-955: expert selection on a 3-D (MoE) weight"""
    self, dim, index = args
    assert dim == 0, f"Float8Tensor aten.select.int with {dim=} is not yet supported"
    assert len(self.qdata.shape) == len(self.scale.shape), "unsupported"
    assert len(self.qdata.shape) == len(self.block_size), "unsupported"
    return Float8Tensor(self.qdata[index], self.scale[index], self.block_size[1:], self.dtype_, self.act_quant_kwargs, self.act_pre_scale)


torch.serialization.add_safe_globals([Float8Tensor, QuantizeTensorToFloat8Kwargs])
