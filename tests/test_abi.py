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
    assert lib.ao_abi_version() == 1


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
