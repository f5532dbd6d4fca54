"""Names from torchao/quantization/quant_primitives.py that the config / tensor mirrors take as arguments."""
from enum import Enum, auto

import torch


class MappingType(Enum):
    """reference quant_primitives.py:64-88.  SYMMETRIC: scale from max |x|, zero-point fixed; ASYMMETRIC: scale from
    max - min, integer zero-point per block."""

    SYMMETRIC = auto()
    SYMMETRIC_NO_CLIPPING_ERR = auto()
    ASYMMETRIC = auto()


torch.serialization.add_safe_globals([MappingType])
