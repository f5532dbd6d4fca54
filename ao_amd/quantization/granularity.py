"""Quantization granularities (reference: torchao/quantization/granularity.py)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class Granularity:
    pass


@dataclass(frozen=True)
class PerTensor(Granularity):
    pass


@dataclass(frozen=True)
class PerRow(Granularity):
    """One scale per row of the last-but-one... i.e. per output feature / per token."""
    dim: int = -1


@dataclass(frozen=True)
class PerGroup(Granularity):
    group_size: int = 128


def get_block_size(shape, granularity):
    """Block size for a granularity (reference: torchao/quantization/utils.py:589)."""
    if isinstance(granularity, PerTensor):
        return tuple(shape)
    if isinstance(granularity, PerRow):
        bs = [1] * len(shape)
        bs[granularity.dim] = shape[granularity.dim]
        return tuple(bs)
    if isinstance(granularity, PerGroup):
        assert shape[-1] % granularity.group_size == 0
        return tuple([1] * (len(shape) - 1) + [granularity.group_size])
    raise ValueError(f"Unsupported Granularity: {granularity}")


import torch as _torch  # noqa: E402

# granularities ride along in the quantized tensors' attributes: allow them under torch.load(weights_only=True)
_torch.serialization.add_safe_globals([Granularity, PerTensor, PerRow, PerGroup])
