"""Int4Tensor: int4 groupwise weights in the PLAIN packing format -- the reference's DEFAULT Int4WeightOnlyConfig format and the
one Float8DynamicActivationInt4WeightConfig uses -- MI355X-native.

Host-side mirror of torchao/quantization/quantize_/workflows/int4/int4_tensor.py (same attribute names, from_hp / linear /
slice semantics).  The reference needs the un-vendored `mslk` package for both the weight preparation and the GEMMs
(int4_tensor.py:22-27, 139-140, 213-229) and only enables them on NVIDIA sm90; here

* from_hp runs the HIP kernel behind ops.int4_plain_quantize (mslk's int4_row_quantize_zp / int4_row_quantize + pack_int4 as
  the reference itself restates them, csrc/int4_plain_kernels.hip);
* F.linear uses the tinygemm kernels: a PLAIN weight and a tile-packed weight describe the same dequantised matrix
  bf16(bf16(q * scale) + zero_point) (zero_point is the value of the middle code in both), so the packed nibbles and the
  scale / zero_point pair are re-laid ONCE into the tile-packed layout (a side buffer next to the checkpoint-format data)
  and `_weight_int4pack_mm` serves it -- same oracle, same kernels, same speed as Int4TilePackedTo4dTensor;
* activation_dtype = float8_e4m3fn: the activation is cast per row to e4m3 (Float8Tensor's cast) and multiplied with the int4 codes
  by the fp8 x int4 MFMA kernel (ops.fp8_int4_linear, csrc/fp8_int4_kernels.hip): group scales on fp32 group sums, the row scale on
  the result -- mslk.f8i4bf16_rowwise's contract.
"""
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import ops
from .base_tensor import LowBitTensorBase, aten

__all__ = ["Int4Tensor"]


class Int4Tensor(LowBitTensorBase):
    """
    Tensor attributes (reference :56-88):
      qdata       uint8 [N, K/2], two codes per byte, even k in the LOW nibble (two's-complement nibbles in [-8, 7])
      scale       bf16 [K/g, N]
      zero_point  bf16 [K/g, N]
    Non-tensor attributes: block_size ([1, g]), shape; optional act_pre_scale, activation_dtype (bf16 | float8_e4m3fn).
    """

    tensor_data_names = ["qdata", "scale", "zero_point"]
    tensor_attribute_names = ["block_size", "shape_"]
    # tp_qdata / tp_scale_and_zero: the compute layout (tinygemm tile order, padded to N % 16 == 0 and K % 128 == 0).  They are
    # inner tensors like the rest -- built at from_hp, carried through flatten / unflatten / detach / to(), so torch.compile and
    # AOT tracing see ready buffers and never the re-layout.  A tensor built without them (a reference checkpoint)
    # gets them at its first eager use.
    optional_tensor_data_names = ["act_pre_scale", "tp_qdata", "tp_scale_and_zero"]

    def __new__(cls, qdata, scale, zero_point, block_size, shape_, act_pre_scale=None, activation_dtype=None, tp_qdata=None,
                tp_scale_and_zero=None):
        kwargs = dict(device=qdata.device, dtype=scale.dtype, requires_grad=False)
        return torch.Tensor._make_wrapper_subclass(cls, torch.Size(shape_), **kwargs)

    def __init__(self, qdata, scale, zero_point, block_size, shape_, act_pre_scale=None, activation_dtype=None, tp_qdata=None,
                 tp_scale_and_zero=None):
        self.qdata = qdata
        self.scale = scale
        self.zero_point = zero_point
        self.block_size = list(block_size)
        self.shape_ = torch.Size(shape_)
        self.act_pre_scale = act_pre_scale
        self.activation_dtype = activation_dtype if activation_dtype is not None else torch.bfloat16
        self.tp_qdata = tp_qdata                    # int32 [Np/8, Kp/128, 32, 4]
        self.tp_scale_and_zero = tp_scale_and_zero  # bf16 [Kp/g, Np, 2]

    def __tensor_flatten__(self):
        return self._data_names(), [self.block_size, self.shape_, self.activation_dtype]

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size, outer_stride):
        block_size, shape_, act_dtype = tensor_attributes
        return cls(tensor_data_dict["qdata"], tensor_data_dict["scale"], tensor_data_dict["zero_point"], block_size, shape_,
                   act_pre_scale=tensor_data_dict.get("act_pre_scale"), activation_dtype=act_dtype,
                   tp_qdata=tensor_data_dict.get("tp_qdata"), tp_scale_and_zero=tensor_data_dict.get("tp_scale_and_zero"))

    def _apply_fn_to_data(self, fn):
        opt = lambda t: fn(t) if t is not None else None  # noqa: E731
        return Int4Tensor(fn(self.qdata), fn(self.scale), fn(self.zero_point), self.block_size, self.shape_, opt(self.act_pre_scale),
                          self.activation_dtype, opt(self.tp_qdata), opt(self.tp_scale_and_zero))

    def release_plain_(self):
        """Serving-only: drop the PLAIN nibbles (the checkpoint layout) and keep the compute layout -- the default keeps both, i.e.
        2 x N K / 2 bytes per weight.  After this the tensor can still run F.linear and dequantize(); slicing and saving in the
        reference's format need the PLAIN data and raise."""
        self.tile_packed()
        self.qdata = self.qdata.new_empty((0,))
        return self

    def _quantization_type(self):
        s = f"shape={tuple(self.shape)}, block_size={self.block_size}, device={self.device}, activation_dtype={self.activation_dtype}"
        if self.act_pre_scale is not None:
            s += f", act_pre_scale.shape={tuple(self.act_pre_scale.shape)}"
        return s

    @classmethod
    def from_hp(cls, w: torch.Tensor, block_size: List[int], activation_dtype: torch.dtype = torch.bfloat16):
        """reference from_hp (:130-186): symmetric codes for fp8 activations, min/max (zero-point) codes for bf16."""
        assert len(block_size) == w.ndim, (
            f"Expecting the length of block_size to be equal to the dimension of the weight, got {block_size=} and {w.ndim=}"
        )
        assert activation_dtype in (torch.bfloat16, torch.float8_e4m3fn), (
            f"activation dtype {activation_dtype} is not supported, supported ones are: bfloat16, float8_e4m3fn"
        )
        assert all(x == 1 for x in block_size[:-1]) and block_size[-1] != 1, "Only groupwise quant is supported right now"
        if w.dtype != torch.bfloat16:
            raise NotImplementedError(f"Int4Tensor.from_hp on MI355X takes bfloat16, got {w.dtype}")
        if w.dim() != 2:
            raise NotImplementedError("Int4Tensor.from_hp on MI355X takes 2-D weights (per-expert 3-D weights: quantize each expert)")
        g = block_size[-1]
        n, k = w.shape
        assert k % g == 0, f"K={k} must be a multiple of the group size {g} (quantize_() leaves such weights unquantized)"
        kpad = (-k) % 128  # the kernel walks 128-k runs; whole zero groups are appended and cut off again
        wq = F.pad(w, (0, kpad)) if kpad else w
        qdata, scale, zero = ops.int4_plain_quantize(wq.contiguous(), g, symmetric=activation_dtype == torch.float8_e4m3fn)
        if kpad:
            qdata, scale, zero = qdata[:, : k // 2].contiguous(), scale[: k // g].contiguous(), zero[: k // g].contiguous()
        out = cls(qdata, scale, zero, list(block_size), w.shape, act_pre_scale=None, activation_dtype=activation_dtype)
        out.tile_packed()  # the compute layout, now (not inside the first forward, which may be a trace)
        return out

    def dequantize(self, output_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """q.to(bf16) * scale + zero_point in the weight dtype (reference formula gptq/api.py:200-221)."""
        qdata_tp, sz = self.tile_packed()
        w = ops.int4_dequantize(qdata_tp, sz, self.block_size[-1])[: self.shape[0], : self.shape[1]]
        return w if output_dtype is None else w.to(output_dtype)

    def tile_packed(self):
        """The compute layout: the same codes and (scale, zero_point) in the tinygemm tile-packed format (gfx950:
        [N/16][K/128][64 lanes][16 B]), N padded to a multiple of 16 and K to a multiple of 128 (padded codes / qparams are zero: they
        meet zero activations or columns that are sliced off) -- built once, kept beside the PLAIN data."""
        if self.tp_qdata is None:
            from torch._subclasses.fake_tensor import is_fake

            if torch.compiler.is_compiling() or is_fake(self.qdata):
                raise RuntimeError("Int4Tensor: the compute layout is missing while tracing; build it eagerly first (from_hp does; for a "
                                   "tensor loaded from a reference checkpoint call weight.tile_packed() once before torch.compile)")
            if self.qdata.numel() == 0:
                raise RuntimeError("Int4Tensor: PLAIN data was released (release_plain_) and no compute layout is present")
            n, k = self.shape
            g = self.block_size[-1]
            npad, kpad = (-n) % 16, (-k) % max(128, g)
            b = self.qdata ^ 0x88                      # two's-complement nibble -> offset-8 code (0..15)
            b = ((b << 4) | (b >> 4))                  # tinygemm's nibble pack keeps even k in the HIGH nibble
            scale, zero = self.scale, self.zero_point
            if npad or kpad:
                b = F.pad(b, (0, kpad // 2, 0, npad), value=0x88)  # code 8 = value 0 * scale + zero_point
                scale = F.pad(scale, (0, npad, 0, kpad // g))
                zero = F.pad(zero, (0, npad, 0, kpad // g))
            self.tp_qdata = ops.convert_weight_to_int4pack(b.contiguous(), 8)
            self.tp_scale_and_zero = torch.stack([scale, zero], dim=-1).to(torch.bfloat16).contiguous()
            assert self.tp_scale_and_zero.shape == ((k + kpad) // g, n + npad, 2)
        return self.tp_qdata, self.tp_scale_and_zero


implements = Int4Tensor.implements
implements_torch_function = Int4Tensor.implements_torch_function


@implements(aten.linear.default)
@implements_torch_function(F.linear)
def _(func, types, args, kwargs):
    """reference :189-234"""
    input_tensor, weight_tensor = args[0], args[1]
    bias = args[2] if len(args) > 2 else kwargs.get("bias", None)
    assert isinstance(weight_tensor, Int4Tensor)
    assert weight_tensor.qdata.is_contiguous(), "Expected qdata to be contiguous"
    assert weight_tensor.scale.is_contiguous(), "Expected scale to be contiguous"
    assert weight_tensor.zero_point.is_contiguous(), "Expected zero_point to be contiguous"
    if weight_tensor.act_pre_scale is not None:
        input_tensor = input_tensor * weight_tensor.act_pre_scale
    orig_act_size = input_tensor.size()
    orig_dtype = input_tensor.dtype
    n_out = weight_tensor.shape[-2]
    x2 = input_tensor.reshape(-1, input_tensor.shape[-1]).to(torch.bfloat16)
    qdata_tp, sz = weight_tensor.tile_packed()
    g = weight_tensor.block_size[-1]
    kp = qdata_tp.shape[1] * 128
    if kp != x2.shape[-1]:  # the compute layout pads K to a multiple of 128
        x2 = F.pad(x2, (0, kp - x2.shape[-1]))
    from ..torch_ops import kernels

    k = kernels(x2)
    if x2.shape[0] == 0:
        res = x2.new_zeros((0, n_out))
    elif weight_tensor.activation_dtype == torch.float8_e4m3fn:
        # dynamic rowwise fp8 activation, then the fp8 x int4 MFMA kernel (mslk.f8i4bf16_rowwise's contract: the e4m3 codes meet the
        # int4 codes on the matrix pipe, group scales multiply fp32 group sums, the row scale the result)
        # (one op: decode sizes run the cast inside the matmul launch, ops.fp8_int4_act_linear)
        res = k.fp8_int4_act_linear(x2.contiguous(), qdata_tp, sz, g, None)
    else:
        res = k.weight_int4pack_mm(x2.contiguous(), qdata_tp, g, sz)
    res = res[:, :n_out].reshape(*orig_act_size[:-1], n_out)
    if bias is not None:
        res = res + bias.to(res.dtype)
    return res.to(orig_dtype)


@implements(aten.slice.Tensor)
def _(func, types, args, kwargs):
    """reference :262-348: dim 0 slices qdata rows and scale / zero_point columns, dim 1 slices packed bytes and groups."""
    self, dim = args[0], args[1] if len(args) > 1 else 0
    start = args[2] if len(args) > 2 and args[2] is not None else 0
    end = args[3] if len(args) > 3 and args[3] is not None else self.shape[dim]
    step = args[4] if len(args) > 4 else 1
    assert step == 1
    assert dim in (0, 1), f"Only dim==0 or 1 are supported, got: {dim}"
    if self.qdata.numel() == 0:
        raise RuntimeError("Int4Tensor: slicing needs the PLAIN data, which release_plain_() dropped")
    end = min(end, self.shape[dim])
    g = self.block_size[-1]
    pre = self.act_pre_scale
    if dim == 0:
        qdata, scale, zero = self.qdata[start:end], self.scale[:, start:end], self.zero_point[:, start:end]
        new_shape = (end - start, self.shape[1])
    else:
        assert start % g == 0 and (end % g == 0 or end == self.shape[1]), f"K slices must align to the group size {g}"
        qdata = self.qdata[:, start // 2 : end // 2]
        scale, zero = self.scale[start // g : (end + g - 1) // g], self.zero_point[start // g : (end + g - 1) // g]
        new_shape = (self.shape[0], end - start)
        if pre is not None and pre.numel() == self.shape[1]:
            pre = pre.reshape(-1)[start:end]
    out = Int4Tensor(qdata.contiguous(), scale.contiguous(), zero.contiguous(), self.block_size, new_shape, pre, self.activation_dtype)
    from torch._subclasses.fake_tensor import is_fake
    if not (torch.compiler.is_compiling() or is_fake(qdata)):
        out.tile_packed()  # a shard is a weight like any other: its compute layout is built with it
    return out


torch.serialization.add_safe_globals([Int4Tensor])
