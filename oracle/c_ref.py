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
