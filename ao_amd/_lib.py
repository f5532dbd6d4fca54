"""ctypes binding of the C-ABI library (include/ao_mi355.h).

The product path has no CPU fallback: if the library is missing or a call
fails, this raises.  `lib()` loads ao_amd/_C_mi355.so (built in-tree by
ao_amd/build.py).
"""
import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AO_MI355_LIB") or os.path.join(_HERE, "_C_mi355.so")  # (AO_MI355_LIB: profiling builds under tools/bin/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ao_mi355.h")

AO_OK = 0
AO_ERR_INVALID_ARGUMENT = -1
AO_ERR_NULL_POINTER = -2
AO_ERR_HIP = -3

_lock = threading.Lock()
_lib = None

_P = ctypes.c_void_p
_I64 = ctypes.c_int64
_INT = ctypes.c_int

# name -> argtypes (restype is int for all but the two noted)
_SIGNATURES = {
    "ao_abi_version": [],
    "ao_prof_enable": [_INT],
    "ao_prof_collect": [_P, _INT, _P],
    "ao_splitk_reserve": [_P, _I64],
    "ao_int4_convert_weight_to_int4pack": [_P, _P, _I64, _I64, _INT, _P],
    "ao_int4_unpack_int4pack": [_P, _P, _I64, _I64, _INT, _P],
    "ao_int4_weight_int4pack_mm": [_P, _P, _P, _P, _I64, _I64, _I64, _INT, _P],
    "ao_int4_dequantize": [_P, _P, _P, _I64, _I64, _INT, _P],
    "ao_int4_quantize_tinygemm": [_P, _P, _P, _I64, _I64, _INT, _P],
    "ao_int4_hqq_workspace_bytes": [_I64, _I64, _INT],
    "ao_int4_quantize_hqq": [_P, _P, _P, _P, _I64, _I64, _INT, _P],
    "ao_int4_plain_quantize": [_P, _P, _P, _P, _I64, _I64, _INT, _INT, _P],
    "ao_int4_set_tuning": [_INT, _INT],
    "ao_int4_set_trace": [_P],
    "ao_gemm8_set_variant": [_INT],
    "ao_gemm8_set_tuning": [_INT, _INT],
    "ao_int4_mm_kernel_name": [_I64, _I64, _I64, _INT],
    "ao_gemm8_kernel_name": [_INT, _I64, _I64, _I64],
    "ao_fp8_int4_kernel_name": [_I64, _I64, _I64, _INT],
    "ao_gemm8_plan": [_INT, _I64, _I64, _I64, _P, _P],
    "ao_gemm8_plan_rows": [_INT, _I64, _I64, _I64, _P],
    "ao_int8_quantize_rowwise": [_P, _P, _P, _I64, _I64, _P],
    "ao_int8_scaled_mm": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "ao_int8_int_mm": [_P, _P, _P, _I64, _I64, _I64, _P],
    "ao_fp8_quantize_rowwise": [_P, _P, _P, _I64, _I64, _P],
    "ao_fp8_scaled_mm": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "ao_mxfp8_quantize_rowwise": [_P, _P, _P, _I64, _I64, _INT, _P],
    "ao_mxfp8_quantize_colwise": [_P, _P, _P, _I64, _I64, _INT, _P],
    "ao_mxfp8_quantize_colwise_3d": [_P, _P, _P, _I64, _I64, _I64, _INT, _P],
    "ao_fp8_grouped_mm": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "ao_mxfp8_grouped_mm": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "ao_mxfp8_grouped_mm_dyn": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _P],
    "ao_mxfp8_grouped_mm_dyn_fits": [_I64, _I64, _I64, _I64],
    "ao_mxfp8_grouped_mm_pair_fits": [_I64, _I64, _I64, _I64],
    "ao_mxfp8_grouped_mm_dyn_pair": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _P],
    "ao_mxfp8_grouped_mm_pair": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _P],
    "ao_dyn_linear_fits": [_I64, _I64, _I64],
    "ao_int8_dynamic_linear": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "ao_fp8_dynamic_linear": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _P],
    "ao_int8_quantize_rowwise_asym": [_P, _P, _P, _P, _I64, _I64, _P],
    "ao_int8_quantize_static": [_P, _P, _P, _INT, _P, _I64, _I64, _P],
    "ao_int8_row_sums": [_P, _P, _I64, _I64, _P],
    "ao_int8_scale_epilogue_asym": [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _P],
    "ao_rowwise_amax": [_P, _I64, _P, _I64, _I64, _P],
    "ao_int8_quantize_rowwise_amax": [_P, _I64, _P, _P, _P, _I64, _I64, _P],
    "ao_fp8_quantize_rowwise_amax": [_P, _I64, _P, _P, _P, _I64, _I64, _P],
    "ao_fp8_mm_f32": [_P, _P, _P, _I64, _I64, _I64, _P],
    "ao_int8_scale_epilogue": [_P, _P, _P, _P, _P, _I64, _I64, _P],
    "ao_fp8_scale_epilogue": [_P, _P, _P, _P, _P, _I64, _I64, _P],
    "ao_moe_padded_rows": [_I64, _I64, _INT],
    "ao_moe_pad_token_groups": [_P, _P, _P, _P, _P, _I64, _I64, _INT, _I64, _INT, _P],
    "ao_moe_unpad_token_groups": [_P, _P, _P, _P, _I64, _I64, _INT, _I64, _P],
    "ao_mx_blocked_rows": [_I64, _I64],
    "ao_mx_block_rearrange_2d_m_groups": [_P, _P, _P, _I64, _I64, _I64, _P],
    "ao_mx_to_blocked": [_P, _P, _I64, _I64, _P],
    "ao_allreduce_flag_bytes": [],
    "ao_allreduce_state_bytes": [],
    "ao_allreduce_oneshot": [_P, _P, _P, _P, _P, _I64, _INT, _I64, _INT, _INT, _P],
    "ao_moe_a2a_flag_bytes": [],
    "ao_moe_a2a_state_bytes": [],
    "ao_moe_a2a_v": [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _I64, _INT, _INT, _P],
    "ao_allreduce_oneshot_op": [_P, _P, _P, _P, _P, _I64, _INT, _INT, _I64, _INT, _INT, _P],
    "ao_peer_alloc": [_P, _I64, _INT],
    "ao_peer_free": [_P],
    "ao_peer_handle_bytes": [],
    "ao_peer_export": [_P, _P],
    "ao_peer_import": [_P, _P],
    "ao_peer_close": [_P],
    "ao_collective_set_timeout_ms": [_INT],
    "ao_collective_timeout_ms": [],
    "ao_fp8_int4_linear": [_P, _P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P],
    "ao_fp8_int4_dynamic_linear": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P],
    "ao_fp8_int4_dynamic_fits": [_I64, _I64, _I64],
    "ao_moe_permute_indices": [_P, _P, _P, _P, _P, _I64, _I64, _I64, _INT, _P],
    "ao_moe_gather_rows": [_P, _P, _P, _I64, _I64, _I64, _P],
    "ao_moe_scatter_rows": [_P, _P, _P, _I64, _I64, _I64, _P],
}


class BackendUnavailable(RuntimeError):
    pass


def declared_symbols():
    """Entry points declared in include/ao_mi355.h (parsed from the header)."""
    with open(HEADER_PATH) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ao_[a-z0-9_]+)\s*\(", text)))


def lib():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise BackendUnavailable(
                f"{LIB_PATH} not found: build it with `python -m ao_amd.build` "
                "(the MI355X backend has no CPU/eager fallback)"
            )
        l = ctypes.CDLL(LIB_PATH)
        older_profiling_build = bool(os.environ.get("AO_MI355_LIB"))  # a library of an earlier commit may lack the newest entry points
        for name, argtypes in _SIGNATURES.items():
            if older_profiling_build and not hasattr(l, name):
                continue
            fn = getattr(l, name)  # AttributeError if the .so is stale
            fn.argtypes = argtypes
            fn.restype = _INT
        l.ao_last_error.argtypes = []
        l.ao_last_error.restype = ctypes.c_char_p
        l.ao_int4_mm_kernel_name.restype = ctypes.c_char_p
        if hasattr(l, "ao_gemm8_kernel_name"):
            l.ao_gemm8_kernel_name.restype = ctypes.c_char_p
        if hasattr(l, "ao_fp8_int4_kernel_name"):
            l.ao_fp8_int4_kernel_name.restype = ctypes.c_char_p
        l.ao_moe_padded_rows.restype = _I64
        l.ao_mx_blocked_rows.restype = _I64
        l.ao_allreduce_flag_bytes.restype = _I64
        l.ao_allreduce_state_bytes.restype = _I64
        if hasattr(l, "ao_moe_a2a_flag_bytes"):
            l.ao_moe_a2a_flag_bytes.restype = _I64
            l.ao_moe_a2a_state_bytes.restype = _I64
        l.ao_int4_hqq_workspace_bytes.restype = _I64
        _lib = l
    return _lib


def last_error():
    return lib().ao_last_error().decode("utf-8", "replace")


def check(rc):
    """Turn a status code into the exception the reference op would raise."""
    if rc == AO_OK:
        return
    msg = last_error()
    if rc == AO_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    raise RuntimeError(msg)
