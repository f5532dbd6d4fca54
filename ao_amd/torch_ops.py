"""PyTorch-dispatcher registration of the MI355X kernels.

Two layers:

1. `ao_mi355::*` custom ops with fake (meta) kernels, so that `torch.compile` / export can trace
   through the tensor subclasses: same tensor contracts as the ATen / torchao ops they replace.
2. `install_aten_overrides()`: registers the int4 kernels as the CUDA-key (= HIP on ROCm) kernels
   of the EXISTING ATen schemas `aten::_convert_weight_to_int4pack` and `aten::_weight_int4pack_mm`
   -- the names torchao's `Int4TilePackedTo4dTensor` calls
   (quantize_/workflows/int4/int4_tile_packed_to_4d_tensor.py:202,287) -- so an unmodified torchao
   picks the MI355X kernels up through PyTorch-ROCm's dispatcher.  Opt-in (it replaces PyTorch
   core's own kernels for every caller in the process); `aten::_int_mm` is offered the same way.
"""
import torch

from . import ops

__all__ = ["install_aten_overrides", "aten_overrides_installed"]

_lib_def = torch.library.Library("ao_mi355", "DEF")
_lib_def.define("weight_int4pack_mm(Tensor x, Tensor qdata, int group_size, Tensor scale_and_zero) -> Tensor")
_lib_def.define("convert_weight_to_int4pack(Tensor w_u8, int inner_k_tiles) -> Tensor")
_lib_def.define("int8_scaled_mm(Tensor xq, Tensor x_scale, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_scaled_mm(Tensor a, Tensor b, Tensor scale_a, Tensor scale_b, Tensor? bias) -> Tensor")
_lib_def.define("int8_dynamic_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_dynamic_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("mxfp8_quantize(Tensor x, str scaling_mode) -> (Tensor, Tensor)")
_lib_def.define("mxfp8_grouped_mm(Tensor a, Tensor a_scale, Tensor b, Tensor b_scale, Tensor offs) -> Tensor")
# same schemas as torchao::fused_pad_token_groups / fused_unpad_token_groups (kernels/mxfp8/quant.py:1244-1246, 1319-1321)
_lib_def.define("fused_pad_token_groups(Tensor inputs, Tensor offsets, int alignment_size) -> (Tensor, Tensor, Tensor)")
_lib_def.define(
    "fused_unpad_token_groups(Tensor inputs, Tensor offsets, Tensor padded_group_start_offsets, int num_tokens, int alignment_size) -> Tensor"
)

_lib_impl = torch.library.Library("ao_mi355", "IMPL", "CUDA")
_lib_impl.impl("weight_int4pack_mm", ops.weight_int4pack_mm)
_lib_impl.impl("convert_weight_to_int4pack", ops.convert_weight_to_int4pack)
_lib_impl.impl("int8_scaled_mm", ops.int8_scaled_mm)
_lib_impl.impl("fp8_scaled_mm", ops.fp8_scaled_mm)
_lib_impl.impl("int8_dynamic_linear", ops.int8_dynamic_linear)
_lib_impl.impl("fp8_dynamic_linear", ops.fp8_dynamic_linear)
_lib_impl.impl("mxfp8_quantize", lambda x, mode: ops.mxfp8_quantize(x, mode))
_lib_impl.impl("mxfp8_grouped_mm", ops.mxfp8_grouped_mm)
_lib_impl.impl("fused_pad_token_groups", ops.fused_pad_token_groups)
_lib_impl.impl("fused_unpad_token_groups", ops.fused_unpad_token_groups)


@torch.library.register_fake("ao_mi355::weight_int4pack_mm")
def _(x, qdata, group_size, scale_and_zero):
    return x.new_empty((x.shape[0], qdata.shape[0] * 8), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::convert_weight_to_int4pack")
def _(w_u8, inner_k_tiles):
    n, kh = w_u8.shape
    return w_u8.new_empty((n // 8, (kh * 2) // (inner_k_tiles * 16), 32, inner_k_tiles // 2), dtype=torch.int32)


@torch.library.register_fake("ao_mi355::int8_scaled_mm")
def _(xq, x_scale, wq, w_scale, bias):
    return xq.new_empty((xq.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_scaled_mm")
def _(a, b, scale_a, scale_b, bias):
    return a.new_empty((a.shape[0], b.shape[1]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::int8_dynamic_linear")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_dynamic_linear")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::mxfp8_quantize")
def _(x, scaling_mode):
    return (x.new_empty(x.shape, dtype=torch.float8_e4m3fn),
            x.new_empty((*x.shape[:-1], x.shape[-1] // 32), dtype=torch.float8_e8m0fnu))


@torch.library.register_fake("ao_mi355::mxfp8_grouped_mm")
def _(a, a_scale, b, b_scale, offs):
    return a.new_empty((a.shape[0], b.shape[1]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fused_pad_token_groups")
def _(inputs, offsets, alignment_size):
    rows = inputs.shape[0] + offsets.shape[0] * alignment_size
    rows = (rows + alignment_size - 1) // alignment_size * alignment_size
    return (inputs.new_empty((rows, inputs.shape[1])), offsets.new_empty(offsets.shape), offsets.new_empty(offsets.shape))


@torch.library.register_fake("ao_mi355::fused_unpad_token_groups")
def _(inputs, offsets, padded_group_start_offsets, num_tokens, alignment_size):
    return inputs.new_empty((num_tokens, inputs.shape[1]))


_aten_impl = None


def aten_overrides_installed() -> bool:
    return _aten_impl is not None


def install_aten_overrides(int_mm: bool = False) -> None:
    """Make `torch.ops.aten._weight_int4pack_mm` / `_convert_weight_to_int4pack` (and optionally
    `_int_mm`) run the MI355X kernels for GPU tensors.  Idempotent."""
    global _aten_impl
    if _aten_impl is not None:
        return
    import warnings

    lib = torch.library.Library("aten", "IMPL")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "Overriding a previously registered kernel": that is the point
        lib.impl("_weight_int4pack_mm", lambda x, q, g, sz: ops.weight_int4pack_mm(x, q, g, sz), "CUDA")
        lib.impl("_convert_weight_to_int4pack", lambda w, ikt: ops.convert_weight_to_int4pack(w, ikt), "CUDA")
        if int_mm:
            lib.impl("_int_mm", lambda a, b: ops.int_mm(a, b), "CUDA")
    _aten_impl = lib
