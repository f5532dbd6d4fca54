"""Int4TilePackedTo4dTensor: tinygemm int4 weight, MI355X-native.

Host-side mirror of torchao/quantization/quantize_/workflows/int4/
int4_tile_packed_to_4d_tensor.py (same attribute names, same from_hp / linear /
slice semantics); the arithmetic runs in the HIP kernels behind ao_amd.ops.
"""
from typing import List, Optional

import torch
import torch.nn.functional as F

from .. import ops
from .base_tensor import LowBitTensorBase, aten

__all__ = ["Int4TilePackedTo4dTensor"]

INNER_K_TILES = 8  # reference fixes this (int4_tile_packed_to_4d_tensor.py:120)


def find_multiple(n: int, k: int) -> int:
    """Smallest multiple of k that is >= n (reference: torchao/utils.py:202)."""
    return n if n % k == 0 else n + k - (n % k)


class Int4TilePackedTo4dTensor(LowBitTensorBase):
    """
    Tensor attributes (reference :34-52):
      qdata           int32 [N/8, K/128, 32, 4], gfx950 tile order
      scale_and_zero  bf16 [K/g, N, 2]
    Non-tensor attributes: block_size (e.g. [1, 128]), shape (original, unpadded).
    Optional: act_pre_scale (multiplied into the activation before the mm).
    """

    tensor_data_names = ["qdata", "scale_and_zero"]
    tensor_attribute_names = ["block_size", "shape"]
    optional_tensor_data_names = ["act_pre_scale"]

    def __new__(cls, qdata, scale_and_zero, block_size, shape, act_pre_scale=None):
        kwargs = dict(device=qdata.device, dtype=torch.bfloat16, requires_grad=False)
        return torch.Tensor._make_wrapper_subclass(cls, shape, **kwargs)

    def __init__(self, qdata, scale_and_zero, block_size, shape, act_pre_scale=None):
        self.qdata = qdata
        self.scale_and_zero = scale_and_zero
        self.block_size = list(block_size)
        self.act_pre_scale = act_pre_scale

    def _quantization_type(self):
        s = f"shape={tuple(self.shape)}, block_size={self.block_size}, device={self.device}"
        if self.act_pre_scale is not None:
            s += f", act_pre_scale.shape={tuple(self.act_pre_scale.shape)}"
        return s

    @classmethod
    def from_hp(cls, hp_tensor: torch.Tensor, block_size: List[int], int4_choose_qparams_algorithm="tinygemm",
                ntile_size: Optional[int] = 16):
        """Quantize a bf16 [N, K] weight (reference from_hp, :96-236).  Pads K to a multiple of 1024 and N to a multiple
        of `ntile_size` (16 on ROCm, quant_api.py:514), then runs the fused choose_qparams + quantize + tile-pack kernel
        (TINYGEMM qparams) or the HQQ optimizer kernels + tile pack (int4_choose_qparams_algorithm = "hqq", :149-168)."""
        assert len(block_size) == hp_tensor.ndim, (
            f"Expecting the length of block_size to be equal to the dimension of the weight, got {block_size=} and {hp_tensor.ndim=}"
        )
        assert all(x == 1 for x in block_size[:-1]), (
            f"Only per group quantization is supported, got block_size: {block_size}"
        )
        assert hp_tensor.dtype == torch.bfloat16, (
            f"Only bfloat16 is supported for Int4TilePackedTo4dTensor, got {hp_tensor.dtype}"
        )
        assert hp_tensor.ndim in (2, 3), "2-D weights, or 3-D [experts, N, K] MoE weights (reference :206-217), are supported"
        if not hp_tensor.is_cuda:
            # reference: test_cant_initialize_in_cpu (needs a GPU device)
            raise RuntimeError("Int4TilePackedTo4dTensor.from_hp requires a GPU tensor")
        if hp_tensor.ndim == 3:
            # "for moe quant" (reference :206-217): every expert is packed on its own; qdata [E, N/8, K/128, 32, 4],
            # scale_and_zero [E, K/g, N, 2]; aten.select.int(0, e) hands expert e to F.linear as a 2-D weight
            per = [cls.from_hp(hp_tensor[i], list(block_size[1:]), int4_choose_qparams_algorithm, ntile_size) for i in range(hp_tensor.shape[0])]
            return cls(torch.stack([p.qdata for p in per]), torch.stack([p.scale_and_zero for p in per]), list(block_size), hp_tensor.shape,
                       act_pre_scale=None)
        group_size = block_size[-1]
        original_shape = hp_tensor.shape
        n0, k0 = original_shape
        nt = 16 if ntile_size is None else max(int(ntile_size), 16)
        k = find_multiple(k0, 1024)
        n = find_multiple(n0, nt)
        w = F.pad(hp_tensor, (0, k - k0, 0, n - n0)) if (k != k0 or n != n0) else hp_tensor
        algo = str(getattr(int4_choose_qparams_algorithm, "value", int4_choose_qparams_algorithm)).lower()
        assert algo in ("tinygemm", "hqq"), f"Unsupported Int4ChooseQParamsAlgorithm: {int4_choose_qparams_algorithm}"
        if algo == "hqq":
            qdata, scale_and_zero = ops.int4_quantize_hqq(w.contiguous(), group_size)
        else:
            qdata, scale_and_zero = ops.int4_quantize_tinygemm(w.contiguous(), group_size)
        return cls(qdata, scale_and_zero, list(block_size), original_shape, act_pre_scale=None)

    def dequantize(self) -> torch.Tensor:
        """bf16 [N, K] (unpadded) with the reference dequant rounding."""
        if self.qdata.dim() == 5:
            return torch.stack([self[i].dequantize() for i in range(self.shape[0])])
        w = ops.int4_dequantize(self.qdata, self.scale_and_zero, self.block_size[-1])
        return w[: self.shape[0], : self.shape[1]]


implements = Int4TilePackedTo4dTensor.implements
implements_torch_function = Int4TilePackedTo4dTensor.implements_torch_function


@implements(aten.linear.default)
@implements_torch_function(F.linear)
def _(func, types, args, kwargs):
    """reference :243-299"""
    input_tensor, weight_tensor = args[0], args[1]
    bias = args[2] if len(args) > 2 else kwargs.get("bias", None)
    assert weight_tensor.qdata.dim() == 4, "F.linear takes a 2-D weight: select an expert of a 3-D weight first (weight[e])"
    assert weight_tensor.qdata.is_contiguous(), "Expected qdata to be contiguous"
    assert weight_tensor.scale_and_zero.is_contiguous(), "Expected scale_and_zero to be contiguous"
    assert weight_tensor.block_size[0] == 1, (
        f"Requires groupwise quantization, got block_size: {weight_tensor.block_size}"
    )
    assert input_tensor.shape[-1] == weight_tensor.shape[1], (
        f"need input_tensor shape: {input_tensor.shape} final"
        f"dim to match weight_tensor shape: {weight_tensor.shape} second dim "
    )
    if weight_tensor.act_pre_scale is not None:
        input_tensor = input_tensor * weight_tensor.act_pre_scale

    orig_act_size = input_tensor.size()
    orig_dtype = input_tensor.dtype
    act_mat = input_tensor.reshape(-1, input_tensor.shape[-1]).to(torch.bfloat16)
    # the reference pads to a multiple of 1024 (what from_hp padded the weight to); a K-sliced
    # shard keeps its own (128-aligned) extent, so pad to what the packed weight actually holds
    pad_size = weight_tensor.qdata.shape[1] * 128
    assert pad_size >= act_mat.shape[-1], "activation is wider than the packed weight"
    if pad_size != act_mat.shape[-1]:
        act_mat = F.pad(act_mat, (0, pad_size - act_mat.shape[-1]))
    groupsize = weight_tensor.block_size[-1]
    n_out = weight_tensor.shape[-2]
    if act_mat.numel() == 0:
        y = act_mat.new_zeros((act_mat.shape[0], n_out))
    else:
        from ..torch_ops import kernels  # dispatcher ops (with fake kernels) while tracing, the direct C-ABI calls otherwise
        y = kernels(act_mat).weight_int4pack_mm(act_mat, weight_tensor.qdata, groupsize, weight_tensor.scale_and_zero)
        y = y[:, :n_out]
    y = y.reshape(*orig_act_size[:-1], n_out)
    if bias is not None:
        y = y + bias.to(y.dtype)
    return y.to(orig_dtype)


@implements(aten.slice.Tensor)
def _(func, types, args, kwargs):
    """Slice along N (dim 0) or K (dim 1) in units the packed layout allows
    (reference :302-367; on gfx950 an N slice must start/end on a multiple of
    16 rows and a K slice on a multiple of 128 and of the group size)."""
    self, dim = args[0], args[1] if len(args) > 1 else 0
    start = args[2] if len(args) > 2 and args[2] is not None else 0
    end = args[3] if len(args) > 3 and args[3] is not None else self.shape[dim]
    step = args[4] if len(args) > 4 else 1
    assert step == 1, "only step=1 slices are supported"
    assert dim in (0, 1), f"Int4TilePackedTo4dTensor slice: dim={dim} is not supported"
    end = min(end, self.shape[dim])
    g = self.block_size[-1]
    if dim == 0:
        assert start % 16 == 0 and (end % 16 == 0 or end == self.shape[0]), "N slices must align to 16 rows"
        qdata = self.qdata[start // 8 : find_multiple(end, 16) // 8].contiguous()
        sz = self.scale_and_zero[:, start : find_multiple(end, 16)].contiguous()
    else:
        unit = max(128, g)
        assert start % unit == 0 and (end % unit == 0 or end == self.shape[1]), (
            f"K slices must align to {unit} columns"
        )
        kend = find_multiple(end, unit)
        # the 4-D shape [N/8, K/128, 32, 4] is nominal on ROCm: memory is
        # [N/16][K/128][64][4] (one wavefront tile per 16 rows x 128 k), so a K
        # slice has to be taken in that view and re-labelled afterwards
        n16, kb = self.qdata.shape[0] // 2, self.qdata.shape[1]
        q = self.qdata.reshape(n16, kb, 64, 4)[:, start // 128 : kend // 128].contiguous()
        qdata = q.reshape(n16 * 2, q.shape[1], 32, 4)
        sz = self.scale_and_zero[start // g : kend // g].contiguous()
    new_shape = list(self.shape)
    new_shape[dim] = end - start
    block_size = list(self.block_size)
    block_size[dim] = min(block_size[dim], new_shape[dim])
    return Int4TilePackedTo4dTensor(qdata, sz, block_size, torch.Size(new_shape), act_pre_scale=self.act_pre_scale)


@implements(aten.select.int)
def _(func, types, args, kwargs):
    """reference :363-385: expert selection on a 3-D (MoE) weight"""
    self, dim, index = args
    assert dim == 0, f"Int4TilePackedTo4dTensor aten.select.int with {dim=} is not yet supported"
    assert self.qdata.dim() == 5, "aten.select.int needs a 3-D (per-expert) weight"
    new_shape = list(self.shape)
    new_shape.pop(dim)
    block_size = list(self.block_size)
    block_size.pop(dim)
    return Int4TilePackedTo4dTensor(self.qdata[index], self.scale_and_zero[index], block_size, torch.Size(new_shape), act_pre_scale=self.act_pre_scale)


torch.serialization.add_safe_globals([Int4TilePackedTo4dTensor])
