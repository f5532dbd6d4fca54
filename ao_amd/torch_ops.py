"""PyTorch-dispatcher registration of the MI355X kernels.

Two layers:

0. `_C_mi355_ops.so` (csrc_torch/binding.cpp): C++ TORCH_LIBRARY_IMPL registrations under the reference's own op names --
   `torchao::mxfp8_quantize`, `torchao::fused_(un)pad_token_groups`, and (with AO_MI355_OVERRIDE_ATEN=1 in the environment
   when it is loaded) `aten::_weight_int4pack_mm`, `_convert_weight_to_int4pack`, `_int_mm`, `_scaled_mm` (rowwise e4m3),
   `_scaled_grouped_mm` (MXFP8).  `load_ops_library()` loads it; importing this module does so when it has been built.
1. `ao_mi355::*` custom ops with fake (meta) kernels, so that `torch.compile` / export can trace
   through the tensor subclasses: same tensor contracts as the ATen / torchao ops they replace.
2. `install_aten_overrides()`: registers the int4 kernels as the CUDA-key (= HIP on ROCm) kernels
   of the EXISTING ATen schemas `aten::_convert_weight_to_int4pack` and `aten::_weight_int4pack_mm`
   -- the names torchao's `Int4TilePackedTo4dTensor` calls
   (quantize_/workflows/int4/int4_tile_packed_to_4d_tensor.py:202,287) -- so an unmodified torchao
   picks the MI355X kernels up through PyTorch-ROCm's dispatcher.  Opt-in (it replaces PyTorch
   core's own kernels for every caller in the process); `aten::_int_mm` is offered the same way.
"""
import os

import torch

from . import _lib, ops

__all__ = ["install_aten_overrides", "aten_overrides_installed", "kernels", "load_ops_library", "OPS_LIB_PATH"]

OPS_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_C_mi355_ops.so")


def _tracing(x) -> bool:
    """True when `x` is not a real tensor being computed on now: Dynamo / AOT / export tracing (FakeTensor, functional
    wrappers, any active dispatch mode)."""
    from torch._subclasses.fake_tensor import is_fake
    from torch.utils._python_dispatch import _get_current_dispatch_mode

    return (torch.compiler.is_compiling() or is_fake(x) or torch._is_functional_tensor(x)
            or _get_current_dispatch_mode() is not None)


def tracing(x=None) -> bool:
    """torch.compile / export / a FakeTensor in hand: only ops with fake kernels may be called."""
    return torch.compiler.is_compiling() or (x is not None and _tracing(x))


def kernels(x=None):
    """Where the subclasses' F.linear implementations get their kernels: while tracing (torch.compile / export; `x` is the
    activation) the `ao_mi355::` dispatcher ops -- they have fake kernels, so FakeTensors flow through, and inductor calls
    them as extern kernels like the reference's `extern_kernels._int_mm` (test_int8_tensor.py:276-278) -- and in eager mode
    the C-ABI wrappers directly (the dispatcher hop costs as much as a decode kernel)."""
    if torch.compiler.is_compiling() or (x is not None and _tracing(x)):
        return torch.ops.ao_mi355
    return ops

_lib_def = torch.library.Library("ao_mi355", "DEF")
_lib_def.define("weight_int4pack_mm(Tensor x, Tensor qdata, int group_size, Tensor scale_and_zero) -> Tensor")
_lib_def.define("convert_weight_to_int4pack(Tensor w_u8, int inner_k_tiles) -> Tensor")
_lib_def.define("int8_scaled_mm(Tensor xq, Tensor x_scale, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_scaled_mm(Tensor a, Tensor b, Tensor scale_a, Tensor scale_b, Tensor? bias) -> Tensor")
_lib_def.define("int8_dynamic_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_dynamic_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("int8_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_linear(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("int8_linear_asym(Tensor x, Tensor wq, Tensor w_scale, Tensor w_row_sums, Tensor? bias) -> Tensor")
_lib_def.define("int8_linear_tensorwise(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_linear_tensorwise(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias) -> Tensor")
_lib_def.define("fp8_linear_clamped(Tensor x, Tensor wq, Tensor w_scale, Tensor? bias, float lb, float ub, bool tensorwise) -> Tensor")
_lib_def.define("int8_linear_static(Tensor x, Tensor wq, Tensor w_scale, Tensor act_scale, Tensor? act_zero_point, Tensor? w_row_sums, Tensor? bias) -> Tensor")
_lib_def.define("fp8_int4_linear(Tensor xq, Tensor x_scale, Tensor qdata, Tensor scale_and_zero, int group_size, Tensor? bias) -> Tensor")
_lib_def.define("fp8_int4_act_linear(Tensor x, Tensor qdata, Tensor scale_and_zero, int group_size, Tensor? bias) -> Tensor")
_lib_def.define("int8_quantize_rowwise(Tensor x) -> (Tensor, Tensor)")
_lib_def.define("fp8_quantize_rowwise(Tensor x) -> (Tensor, Tensor)")
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
_lib_impl.impl("int8_linear", ops.int8_linear)
_lib_impl.impl("fp8_linear", ops.fp8_linear)
_lib_impl.impl("int8_linear_asym", ops.int8_linear_asym)
_lib_impl.impl("int8_linear_tensorwise", ops.int8_linear_tensorwise)
_lib_impl.impl("fp8_linear_tensorwise", ops.fp8_linear_tensorwise)
_lib_impl.impl("fp8_linear_clamped", ops.fp8_linear_clamped)
_lib_impl.impl("int8_linear_static", ops.int8_linear_static)
_lib_impl.impl("fp8_int4_linear", ops.fp8_int4_linear)
_lib_impl.impl("fp8_int4_act_linear", ops.fp8_int4_act_linear)
_lib_impl.impl("int8_quantize_rowwise", ops.int8_quantize_rowwise)
_lib_impl.impl("fp8_quantize_rowwise", ops.fp8_quantize_rowwise)
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


@torch.library.register_fake("ao_mi355::int8_linear")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_linear")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::int8_linear_asym")
def _(x, wq, w_scale, w_row_sums, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::int8_linear_tensorwise")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_linear_tensorwise")
def _(x, wq, w_scale, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_linear_clamped")
def _(x, wq, w_scale, bias, lb, ub, tensorwise):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::int8_linear_static")
def _(x, wq, w_scale, act_scale, act_zero_point, w_row_sums, bias):
    return x.new_empty((x.shape[0], wq.shape[0]), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_int4_linear")
def _(xq, x_scale, qdata, scale_and_zero, group_size, bias):
    return xq.new_empty((xq.shape[0], qdata.shape[0] * 8), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::fp8_int4_act_linear")
def _(x, qdata, scale_and_zero, group_size, bias):
    return x.new_empty((x.shape[0], qdata.shape[0] * 8), dtype=torch.bfloat16)


@torch.library.register_fake("ao_mi355::int8_quantize_rowwise")
def _(x):
    return x.new_empty(x.shape, dtype=torch.int8), x.new_empty((x.shape[0], 1), dtype=torch.float32)


@torch.library.register_fake("ao_mi355::fp8_quantize_rowwise")
def _(x):
    return x.new_empty(x.shape, dtype=torch.float8_e4m3fn), x.new_empty((x.shape[0], 1), dtype=torch.float32)


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


# ---- autograd: these are inference kernels ------------------------------------------------------------------------------------
# Without an Autograd-key registration PyTorch warns on every call that sees a tensor requiring grad ("an autograd kernel was not
# registered": 34 warnings in the round-2 GPU log).  The ops get an explicit FALLTHROUGH on the Autograd key: forward works on any
# input, the outputs are cut off from the graph -- what the reference's inference subclasses do too (their aten kernels run on
# `requires_grad=False` parameters).  (A backward stub that raises was tried first: AOT autograd traces the joint graph of a
# module whose bias requires grad and would hit it at torch.compile time.)
_lib_autograd = torch.library.Library("ao_mi355", "IMPL", "Autograd")
for _name in ("weight_int4pack_mm", "convert_weight_to_int4pack", "int8_scaled_mm", "fp8_scaled_mm", "int8_dynamic_linear", "fp8_dynamic_linear",
              "int8_linear", "fp8_linear", "int8_linear_asym", "int8_linear_tensorwise", "fp8_linear_tensorwise", "fp8_linear_clamped",
              "int8_linear_static", "fp8_int4_linear", "fp8_int4_act_linear", "int8_quantize_rowwise", "fp8_quantize_rowwise", "mxfp8_quantize", "mxfp8_grouped_mm",
              "fused_pad_token_groups", "fused_unpad_token_groups"):
    _lib_autograd.impl(_name, torch.library.fallthrough_kernel)


# ---- the C++ registrations (ao_amd/csrc_torch/binding.cpp -> _C_mi355_ops.so) ------------------------------------------
# torchao::mxfp8_quantize / fused_pad_token_groups / fused_unpad_token_groups / mx_block_rearrange_2d_M_groups are IMPLEMENTED in the .so under the
# reference's names (TORCH_LIBRARY_IMPL(torchao, CUDA)); their schemas are defined by torchao's Python when it is imported
# (prototype/mx_formats/kernels.py:1022-1026, moe_training/kernels/mxfp8/quant.py:1244-1246, 1319-1321) and by this module
# otherwise.  Dropped into a torchao checkout as torchao/_C_mi355_ops.so the library is found by torchao's own loader
# (torchao/__init__.py:89-94).
_TORCHAO_SCHEMAS = {
    "mxfp8_quantize": "mxfp8_quantize(Tensor input, bool rowwise, bool colwise, int scale_dim_x, int scale_dim_y, str fp8_format, "
                      "str scaling_mode) -> (Tensor, Tensor, Tensor, Tensor)",
    "fused_pad_token_groups": "fused_pad_token_groups(Tensor inputs, Tensor group_offsets, int alignment_size) -> (Tensor, Tensor, Tensor)",
    "fused_unpad_token_groups": "fused_unpad_token_groups(Tensor inputs, Tensor group_offsets, Tensor padded_group_start_offsets, "
                                "int num_tokens, int alignment_size) -> Tensor",
    # (kernels/mxfp8/quant.py:969-973)
    "mx_block_rearrange_2d_M_groups": "mx_block_rearrange_2d_M_groups(Tensor scales_tensor, Tensor input_offsets, int chunks_per_tb) -> Tensor",
}
_ops_lib_loaded = False
_torchao_def = None


def load_ops_library() -> bool:
    """torch.ops.load_library(_C_mi355_ops.so) once; defines the torchao:: schemas if nobody has.  Returns False when the
    library has not been built (python -m ao_amd.build)."""
    global _ops_lib_loaded, _torchao_def
    if _ops_lib_loaded:
        return True
    if not os.path.exists(OPS_LIB_PATH):
        return False
    _lib.lib()  # the C-ABI library the ops library links against (same directory; loaded first so that it resolves)
    torch.ops.load_library(OPS_LIB_PATH)
    _torchao_def = torch.library.Library("torchao", "FRAGMENT")
    for name, schema in _TORCHAO_SCHEMAS.items():
        if not _schema_defined(f"torchao::{name}"):
            _torchao_def.define(schema)
    _register_torchao_fakes()
    _ops_lib_loaded = True
    return True


def _schema_defined(qualname: str) -> bool:
    try:
        torch._C._dispatch_find_schema_or_throw(qualname, "")
        return True
    except RuntimeError:
        return False


def _register_torchao_fakes():
    def _try(name, fn):
        try:
            torch.library.register_fake(f"torchao::{name}")(fn)
        except RuntimeError:
            pass  # torchao's own Python registered one already

    def mx_fake(input, rowwise, colwise, scale_dim_x, scale_dim_y, fp8_format, scaling_mode):
        r, c = input.shape
        e = lambda *shape, dt: input.new_empty(shape, dtype=dt)  # noqa: E731
        out_r = e(r, c, dt=torch.float8_e4m3fn) if rowwise else e(0, dt=torch.float8_e4m3fn)
        sc_r = e(r, c // 32, dt=torch.float8_e8m0fnu) if rowwise else e(0, dt=torch.float8_e8m0fnu)
        out_c = e(c, r, dt=torch.float8_e4m3fn).t() if colwise else e(0, dt=torch.float8_e4m3fn)
        sc_c = e(r // 32, c, dt=torch.float8_e8m0fnu).t() if colwise else e(0, dt=torch.float8_e8m0fnu)
        return out_r, out_c, sc_r, sc_c

    def pad_fake(inputs, group_offsets, alignment_size):
        rows = inputs.shape[0] + group_offsets.shape[0] * alignment_size
        rows = (rows + alignment_size - 1) // alignment_size * alignment_size
        return (inputs.new_empty((rows, inputs.shape[1])), group_offsets.new_empty(group_offsets.shape), group_offsets.new_empty(group_offsets.shape))

    def unpad_fake(inputs, group_offsets, padded_group_start_offsets, num_tokens, alignment_size):
        return inputs.new_empty((num_tokens, inputs.shape[1]))

    def rearrange_fake(scales_tensor, input_offsets, chunks_per_tb):  # (kernels/mxfp8/quant.py:1228-1247)
        rows, cols = scales_tensor.shape
        return scales_tensor.new_empty((rows + input_offsets.shape[0] * 128, (cols + 3) // 4 * 4))

    _try("mx_block_rearrange_2d_M_groups", rearrange_fake)
    _try("mxfp8_quantize", mx_fake)
    _try("fused_pad_token_groups", pad_fake)
    _try("fused_unpad_token_groups", unpad_fake)


_aten_impl = None


def aten_overrides_installed() -> bool:
    if _aten_impl is not None:
        return True
    return bool(_ops_lib_loaded and torch.ops.ao_mi355_c.aten_overrides_active())


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


load_ops_library()
