"""Inference configs (reference: torchao/core/config.py:27 AOBaseConfig and
torchao/quantization/quant_api.py:502,808,1112)."""
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Union

from .granularity import Granularity, PerRow, PerTensor
from .quant_primitives import MappingType


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


def _normalize_granularity(granularity, default, what):
    """(activation, weight) granularities from None / one Granularity / a list of two (reference Int8Tensor._normalize_granularity
    int8_tensor.py:140-174; quantization/utils.py _normalize_granularity for float8)."""
    if granularity is None:
        return default(), default()
    if isinstance(granularity, Granularity):
        pair = (granularity, granularity)
    elif isinstance(granularity, (list, tuple)):
        if len(granularity) != 2:
            raise ValueError(f"Granularity list must have exactly 2 elements, got {len(granularity)}: {granularity}")
        pair = tuple(granularity)
    else:
        raise ValueError(f"Invalid granularity type: {granularity}. Expected None, Granularity, or list of 2 Granularities.")
    for g in pair:
        if not isinstance(g, (PerRow, PerTensor)):
            raise ValueError(f"{what}: only PerTensor and PerRow are supported, got {g}")
    return pair


@dataclass
class Int8DynamicActivationInt8WeightConfig(AOBaseConfig):
    """int8 dynamic activation x int8 weight (reference quant_api.py:808-884): granularity PerRow (default) or PerTensor, one
    value for both operands or [activation, weight]; act_mapping_type SYMMETRIC (default) or ASYMMETRIC."""

    act_mapping_type: MappingType = MappingType.SYMMETRIC
    granularity: Optional[Union[Granularity, List[Granularity]]] = field(default_factory=PerRow)
    weight_only_decode: bool = False
    set_inductor_config: bool = False
    version: int = 2
    reduce_range: bool = False

    def __post_init__(self):
        if self.weight_only_decode:
            raise NotImplementedError("weight_only_decode (int8 weight-only at decode sizes) is outside the SURVEY.md section 8 path")
        if self.version == 1:
            raise ValueError("version 1 of Int8DynamicActivationInt8WeightConfig has been removed, please use version 2")
        if self.reduce_range:
            raise NotImplementedError("reduce_range is a CPU-without-VNNI option; the MI355X int8 MFMA path uses the full range")
        if self.act_mapping_type not in (MappingType.SYMMETRIC, MappingType.ASYMMETRIC):
            raise ValueError(f"act_mapping_type must be SYMMETRIC or ASYMMETRIC, got {self.act_mapping_type}")
        self.granularity = list(_normalize_granularity(self.granularity, PerRow, "Int8DynamicActivationInt8WeightConfig"))


@dataclass
class Int8StaticActivationInt8WeightConfig(AOBaseConfig):
    """int8 STATIC activation (calibrated scale / zero-point given) x int8 weight (reference quant_api.py:919-1012)."""

    act_quant_scale: Optional[object] = None        # torch.Tensor, fp32, one element (PerTensor) or one per activation row
    act_quant_zero_point: Optional[object] = None   # torch.Tensor, int8, same shape (asymmetric only)
    granularity: Optional[Union[Granularity, List[Granularity]]] = field(default_factory=PerRow)
    act_mapping_type: MappingType = MappingType.SYMMETRIC
    set_inductor_config: bool = False
    version: int = 1
    reduce_range: bool = False

    def __post_init__(self):
        if self.reduce_range:
            raise NotImplementedError("reduce_range is a CPU-without-VNNI option; the MI355X int8 MFMA path uses the full range")
        assert self.act_mapping_type in (MappingType.SYMMETRIC, MappingType.ASYMMETRIC), (
            "Int8StaticActivationInt8WeightConfig requires `act_mapping_type` in (MappingType.SYMMETRIC, MappingType.ASYMMETRIC)."
        )
        self.granularity = list(_normalize_granularity(self.granularity, PerRow, "Int8StaticActivationInt8WeightConfig"))

    def get_act_quant_kwargs(self):
        from .int8_tensor import QuantizeTensorToInt8Kwargs
        return QuantizeTensorToInt8Kwargs(granularity=self.granularity[0], mapping_type=self.act_mapping_type, reduce_range=self.reduce_range)


@dataclass
class Float8DynamicActivationFloat8WeightConfig(AOBaseConfig):
    """float8 e4m3 dynamic activation x float8 weight (reference quant_api.py:1112-1297).  granularity: None = PerTensor for
    both (the reference's default), PerRow() (the BASELINE configuration), or [activation, weight] of the same type."""

    granularity: Optional[Union[Granularity, List[Granularity]]] = None
    activation_value_lb: Optional[float] = None  # bounds on the activation amax the scale is calculated from (reference :1126-1127)
    activation_value_ub: Optional[float] = None
    # accepted for source compatibility with the reference's call sites; there is one kernel family on MI355X (fp32 accumulation on
    # the scaled MFMA whatever use_fast_accum says, no mslk / torch choice to make), so neither changes what runs
    mm_config: Optional[object] = None
    kernel_preference: object = "auto"
    set_inductor_config: bool = False
    version: int = 2

    def __post_init__(self):
        act, weight = _normalize_granularity(self.granularity, PerTensor, "Float8DynamicActivationFloat8WeightConfig")
        if type(act) is not type(weight):
            raise ValueError(f"Different granularities for activation and weight are not supported: {act}, {weight}")
        self.granularity = [act, weight]


@dataclass
class Float8DynamicActivationInt4WeightConfig(AOBaseConfig):
    """float8 e4m3 rowwise dynamic activation x int4 groupwise (symmetric) weight (reference quant_api.py:630-699: group_size 128).

    `int4_packing_format`: "preshuffled" (the reference's default, :646) or "plain".  The reference's preshuffled tensor is the same
    quantization in a layout pre-arranged for its H100 kernel; the MI355X counterpart of that is the PLAIN `Int4Tensor` carrying its
    gfx950 compute layout (tile-packed codes + stacked scale / zero, built once at from_hp) -- both values produce it, so the
    reference's default config runs unchanged.  The checkpoint-format nibbles stay available (`release_plain_()` drops them)."""

    int4_packing_format: Int4PackingFormat = Int4PackingFormat.PRESHUFFLED
    group_size: int = 128

    def __post_init__(self):
        self.int4_packing_format = Int4PackingFormat(self.int4_packing_format)


@dataclass
class FqnToConfig(AOBaseConfig):
    """Different configs for different parts of a model, keyed by fully qualified name (reference quant_api.py:1515-1596).

    Keys of `fqn_to_config`, in order of precedence: the fqn of a parameter ("model.layers.0.q_proj.weight"), the fqn of a module,
    "re:<regex>" matched in full against parameter fqns, "re:<regex>" against module fqns (the first matching key wins), and
    "_default" (every other nn.Linear).  A value of None leaves the match unquantized.  `module_fqn_to_config` is the old name of
    the same field."""

    fqn_to_config: dict = field(default_factory=dict)
    module_fqn_to_config: dict = field(default_factory=dict)
    version: int = 1

    def __post_init__(self):
        if self.fqn_to_config and self.module_fqn_to_config and self.fqn_to_config != self.module_fqn_to_config:
            raise ValueError("`fqn_to_config` and `module_fqn_to_config` are both specified and are not equal!")
        if self.module_fqn_to_config and not self.fqn_to_config:
            self.fqn_to_config = self.module_fqn_to_config
        if self.fqn_to_config and not self.module_fqn_to_config:
            self.module_fqn_to_config = self.fqn_to_config


ModuleFqnToConfig = FqnToConfig  # the reference keeps the old name (quant_api.py:1598)
