"""CPU: the C-ABI library loads, exports every symbol the header declares, and
rejects bad arguments on the host (no kernel is launched in this file)."""
import ctypes

import pytest

from ao_amd import _lib


def test_library_loads_and_exports_declared_symbols():
    lib = _lib.lib()
    names = _lib.declared_symbols()
    assert "ao_int4_weight_int4pack_mm" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ao_mi355.h but not exported"
    assert set(_lib._SIGNATURES) | {"ao_last_error"} == set(names)
    assert lib.ao_abi_version() == 2


def test_null_pointer_is_reported():
    lib = _lib.lib()
    rc = lib.ao_int4_convert_weight_to_int4pack(None, None, 16, 128, 8, None)
    assert rc == _lib.AO_ERR_NULL_POINTER
    assert "null pointer" in _lib.last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


@pytest.mark.parametrize(
    "n,k,g,frag",
    [(15, 128, 128, "multiple of 16"), (16, 100, 128, "multiple of"), (16, 128, 48, "qGroupSize"), (16, 128, 256, "not divisible")],
)
def test_bad_shapes_are_invalid_argument(n, k, g, frag):
    lib = _lib.lib()
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.ao_int4_weight_int4pack_mm(one, one, one, one, 1, n, k, g, None)
    assert rc == _lib.AO_ERR_INVALID_ARGUMENT
    assert frag in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_inner_k_tiles_must_be_8():
    lib = _lib.lib()
    one = ctypes.c_void_p(16)
    assert lib.ao_int4_convert_weight_to_int4pack(one, one, 16, 128, 4, None) == _lib.AO_ERR_INVALID_ARGUMENT


def test_empty_m_is_ok_without_gpu():
    lib = _lib.lib()
    one = ctypes.c_void_p(16)
    assert lib.ao_int4_weight_int4pack_mm(None, one, one, None, 0, 16, 128, 128, None) == _lib.AO_OK


def test_ops_library_loads_and_registers_without_a_gpu():
    """ao_amd/_C_mi355_ops.so (C++ TORCH_LIBRARY_IMPL registrations, csrc_torch/binding.cpp) loads through
    torch.ops.load_library on a machine without a GPU and registers CUDA-key kernels under the reference's op names;
    the aten overrides stay off unless AO_MI355_OVERRIDE_ATEN=1 was set before loading."""
    import torch

    from ao_amd import torch_ops

    assert torch_ops.load_ops_library(), "build it with python -m ao_amd.build"
    for name in ("mxfp8_quantize", "fused_pad_token_groups", "fused_unpad_token_groups", "mx_block_rearrange_2d_M_groups"):
        assert torch._C._dispatch_has_kernel_for_dispatch_key(f"torchao::{name}", "CUDA")
    s = str(torch.ops.torchao.mxfp8_quantize.default._schema)
    assert "bool rowwise, bool colwise, int scale_dim_x, int scale_dim_y, str fp8_format, str scaling_mode" in s
    s = str(torch.ops.torchao.mx_block_rearrange_2d_M_groups.default._schema)  # (reference: kernels/mxfp8/quant.py:969-973)
    assert "(Tensor scales_tensor, Tensor input_offsets, int chunks_per_tb) -> Tensor" in s
    for name in ("_weight_int4pack_mm", "_convert_weight_to_int4pack", "_int_mm", "_scaled_mm", "_scaled_grouped_mm"):
        ours = str(getattr(torch.ops.ao_mi355_c, name).default._schema).split("::", 1)[1]
        theirs = str(getattr(torch.ops.aten, name).default._schema).split("::", 1)[1]
        assert ours == theirs, (ours, theirs)  # the override installs exactly the ATen signature
    assert not torch.ops.ao_mi355_c.aten_overrides_active()
    if not torch.cuda.is_available():
        import pytest
        with pytest.raises((RuntimeError, NotImplementedError)):  # no CPU kernel: the product path fails loudly
            torch.ops.torchao.mxfp8_quantize(torch.zeros(32, 32, dtype=torch.bfloat16), True, False, 32, 1, "e4m3", "rceil")
