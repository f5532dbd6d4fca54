"""Inference configs (reference: torchao/core/config.py:27 AOBaseConfig and
torchao/quantization/quant_api.py:502,808,1112)."""
import enum
from dataclasses import dataclass, field
from typing import Optional

from .granularity import Granularity, PerRow


class AOBaseConfig:
    """Marker base class: a config selects a per-module transform in quantize_."""


class Int4PackingFormat(str, enum.Enum):
    PLAIN = "plain"
    PRESHUFFLED = "preshuffled"
    PLAIN_INT32 = "plain_int32"
    TILE_PACKED_TO_4D = "tile_packed_to_4d"


class Int4ChooseQParamsAlgorithm(str, enum.Enum):
    TINYGEMM = "tinygemm"
    HQQ = "hqq"


@dataclass
class Int4WeightOnlyConfig(AOBaseConfig):
    """int4 groupwise weight-only (reference quant_api.py:502-533).

    Defaults as the reference: PLAIN packing (Int4Tensor -- served on MI355X by the tinygemm kernels through a one-time
    re-layout, int4_plain_tensor.py), tinygemm qparams.  TILE_PACKED_TO_4D (Int4TilePackedTo4dTensor) is the format whose
    checkpoint bytes the HIP `_weight_int4pack_mm` kernel reads directly; ntile is 16 on ROCm."""

    group_size: int = 128
    set_inductor_config: bool = False
    int4_packing_format: Int4PackingFormat = Int4PackingFormat.PLAIN
    int4_choose_qparams_algorithm: Int4ChooseQParamsAlgorithm = Int4ChooseQParamsAlgorithm.TINYGEMM
    int4_tile_packed_ntile: int = 16
    version: int = 2

    def __post_init__(self):
        assert self.int4_tile_packed_ntile in [8, 16], "int4_tile_packed_ntile must be either 8 or 16"
        self.int4_packing_format = Int4PackingFormat(self.int4_packing_format)
        self.int4_choose_qparams_algorithm = Int4ChooseQParamsAlgorithm(self.int4_choose_qparams_algorithm)


@dataclass
class Int8DynamicActivationInt8WeightConfig(AOBaseConfig):
    """int8 per-token dynamic activation x int8 per-row weight (reference
    quant_api.py:808-884; defaults PerRow symmetric)."""

    granularity: Granularity = field(default_factory=PerRow)
    set_inductor_config: bool = False
    version: int = 2


@dataclass
class Float8DynamicActivationFloat8WeightConfig(AOBaseConfig):
    """float8 e4m3 rowwise dynamic activation x float8 weight (reference
    quant_api.py:1112-1297).  Only PerRow granularity is implemented."""

    granularity: Granularity = field(default_factory=PerRow)
    set_inductor_config: bool = False
    version: int = 2


@dataclass
class Float8DynamicActivationInt4WeightConfig(AOBaseConfig):
    """float8 e4m3 rowwise dynamic activation x int4 groupwise (symmetric) weight, PLAIN packing (reference
    quant_api.py:630-699: group_size 128, Int4Tensor with activation_dtype float8_e4m3fn)."""

    int4_packing_format: Int4PackingFormat = Int4PackingFormat.PLAIN
    group_size: int = 128

    def __post_init__(self):
        self.int4_packing_format = Int4PackingFormat(self.int4_packing_format)
