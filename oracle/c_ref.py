"""ctypes loader for oracle/lowbit_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblowbit_ref.so")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "lowbit_ref.c")):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.ao_ref_num_threads.restype = ctypes.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def num_threads():
    return int(lib().ao_ref_num_threads())


def int4_dequantize(qdata, sz_bits, n, k, g):
    qdata = np.ascontiguousarray(qdata, dtype=np.int32)
    sz_bits = np.ascontiguousarray(sz_bits, dtype=np.uint16)
    out = np.empty((n, k), dtype=np.uint16)
    lib().ao_ref_int4_dequantize(_p(qdata), _p(sz_bits), _p(out), ctypes.c_int64(n), ctypes.c_int64(k), ctypes.c_int(g))
    return out


def int4_linear(x_bits, qdata, sz_bits, n, k, g):
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    qdata = np.ascontiguousarray(qdata, dtype=np.int32)
    sz_bits = np.ascontiguousarray(sz_bits, dtype=np.uint16)
    m = x_bits.shape[0]
    y = np.empty((m, n), dtype=np.uint16)
    lib().ao_ref_int4_linear(
        _p(x_bits), _p(qdata), _p(sz_bits), _p(y), ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k), ctypes.c_int(g)
    )
    return y


def int8_dynamic_linear(x_bits, wq, ws):
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    wq = np.ascontiguousarray(wq, dtype=np.int8)
    ws = np.ascontiguousarray(ws, dtype=np.float32)
    (m, k), n = x_bits.shape, wq.shape[0]
    y = np.empty((m, n), dtype=np.uint16)
    lib().ao_ref_int8_dynamic_linear(_p(x_bits), _p(wq), _p(ws), _p(y), ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k))
    return y


def fp8_rowwise_linear(x_bits, wq, ws):
    x_bits = np.ascontiguousarray(x_bits, dtype=np.uint16)
    wq = np.ascontiguousarray(wq, dtype=np.uint8)
    ws = np.ascontiguousarray(ws, dtype=np.float32)
    (m, k), n = x_bits.shape, wq.shape[0]
    y = np.empty((m, n), dtype=np.uint16)
    lib().ao_ref_fp8_rowwise_linear(_p(x_bits), _p(wq), _p(ws), _p(y), ctypes.c_int64(m), ctypes.c_int64(n), ctypes.c_int64(k))
    return y


def mxfp8_grouped_mm(a_bits, wq, wscale, offs):
    a_bits = np.ascontiguousarray(a_bits, dtype=np.uint16)
    wq = np.ascontiguousarray(wq, dtype=np.uint8)
    wscale = np.ascontiguousarray(wscale, dtype=np.uint8)
    offs = np.ascontiguousarray(offs, dtype=np.int32)
    (m, k), (e, n, _) = a_bits.shape, wq.shape
    y = np.empty((m, n), dtype=np.uint16)
    lib().ao_ref_mxfp8_grouped_mm(_p(a_bits), _p(wq), _p(wscale), _p(offs), _p(y), ctypes.c_int64(m), ctypes.c_int64(e),
                                  ctypes.c_int64(n), ctypes.c_int64(k))
    return y
