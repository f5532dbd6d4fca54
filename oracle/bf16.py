"""bfloat16 emulation on top of numpy float32 (TEST INFRASTRUCTURE ONLY).

PyTorch evaluates every bf16 elementwise op as: upcast operands to fp32, do the
op in fp32 (IEEE, round-to-nearest-even), round the fp32 result back to bf16
(round-to-nearest-even).  ``bf16_round`` is that last step; arrays that "are
bf16" in this oracle are float32 arrays whose low 16 mantissa bits are zero.
"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32. NaN stays NaN."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).copy()
    nan = np.isnan(x)
    if nan.any():
        out[nan] = np.float32(np.nan)
    return out.reshape(x.shape)


def is_bf16(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return bool(np.all((x.view(np.uint32) & 0xFFFF) == 0))


def to_bits(x):
    """bf16-valued fp32 array -> uint16 bit patterns."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    return (x.view(np.uint32) >> 16).astype(np.uint16)


def from_bits(b):
    """uint16 bf16 bit patterns -> fp32 array."""
    b = np.ascontiguousarray(b, dtype=np.uint16)
    return (b.astype(np.uint32) << 16).view(np.float32)


def add(a, b):
    return bf16_round(np.float32(a) + np.float32(b))


def sub(a, b):
    return bf16_round(np.float32(a) - np.float32(b))


def mul(a, b):
    return bf16_round(np.float32(a) * np.float32(b))


def div(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        return bf16_round(np.float32(a) / np.float32(b))
