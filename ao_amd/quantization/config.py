"""Inference configs (reference: torchao/core/config.py:27 AOBaseConfig and
torchao/quantization/quant_api.py:502,808,1112)."""
import enum
from dataclasses import dataclass, field
from typing import List, Optional, Union

from .granularity import Granularity, PerRow, PerTensor
from .quant_primitives import MappingType


class AOBaseConfig:
    """Base class of the workflow configs: a config selects a per-module transform in quantize_ (reference torchao/core/config.py:27).
    `version` is an INSTANCE attribute in the subclasses that bump it (several versions of one config co-exist in checkpoints); 1 is
    the default of those that never did.  JSON (de)serialisation: config_to_dict / config_from_dict below."""

    version: int = 1


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


class KernelPreference(str, enum.Enum):
    """reference quantize_/common/kernel_preference.py:17-47.  One kernel family exists on MI355X: AUTO and TORCH both mean it; the
    others name libraries that are not this backend."""
    AUTO = "auto"
    TORCH = "torch"
    MSLK = "mslk"
    EMULATED = "emulated"
    DEEPGEMM = "deepgemm"
    TRITON = "triton"


class Float8PackingFormat(str, enum.Enum):
    """reference float8_packing_format.py:19-40 (PLAIN is what this backend implements; the 2:4-sparse formats are out of scope)."""
    PLAIN = "plain"
    SPARSE_CUTLASS = "sparse_cutlass"
    SPARSE_2D_DATA_2D_METADATA = "sparse_2d_data_2d_metadata"
    SPARSE_2D_DATA_1D_METADATA = "sparse_2d_data_1d_metadata"


@dataclass
class Float8MMConfig:
    """reference float8/inference.py:26-39 (a NamedTuple there): carried for (de)serialisation; fp32 accumulation on the scaled MFMA is
    what runs whatever use_fast_accum says, and K is never padded (K % 16 == 0 is required, as the reference's skip rule has it)."""
    emulate: bool = False
    use_fast_accum: bool = False
    pad_inner_dim: bool = False


@dataclass
class Float8DynamicActivationFloat8WeightConfig(AOBaseConfig):
    """float8 e4m3 dynamic activation x float8 weight (reference quant_api.py:1112-1297; same fields in the same order, so that a
    config the reference serialised decodes here).  granularity: None = PerTensor for both (the reference's default), PerRow() (the
    BASELINE configuration), or [activation, weight] of the same type."""

    activation_dtype: object = None  # torch.float8_e4m3fn (None: that); gfx950 implements OCP e4m3fn only
    weight_dtype: object = None
    granularity: Optional[Union[Granularity, List[Granularity]]] = None
    packing_format: Optional[Float8PackingFormat] = Float8PackingFormat.PLAIN
    # accepted for source / checkpoint compatibility with the reference; there is one kernel family on MI355X (fp32 accumulation on
    # the scaled MFMA whatever use_fast_accum says, no mslk / torch choice to make), so neither changes what runs
    mm_config: Optional[object] = None
    activation_value_lb: Optional[float] = None  # bounds on the activation amax the scale is calculated from (reference :1126-1127)
    activation_value_ub: Optional[float] = None
    kernel_preference: object = KernelPreference.AUTO
    set_inductor_config: bool = False
    version: int = 2
    alg_id: int = 0

    def __post_init__(self):
        import torch

        act, weight = _normalize_granularity(self.granularity, PerTensor, "Float8DynamicActivationFloat8WeightConfig")
        if type(act) is not type(weight):
            raise ValueError(f"Different granularities for activation and weight are not supported: {act}, {weight}")
        self.granularity = [act, weight]
        self.activation_dtype = torch.float8_e4m3fn if self.activation_dtype is None else self.activation_dtype
        self.weight_dtype = torch.float8_e4m3fn if self.weight_dtype is None else self.weight_dtype
        if self.activation_dtype != torch.float8_e4m3fn or self.weight_dtype != torch.float8_e4m3fn:
            raise NotImplementedError(f"Float8DynamicActivationFloat8WeightConfig on MI355X implements float8_e4m3fn for both operands, got "
                                      f"{self.activation_dtype} / {self.weight_dtype}")
        self.packing_format = Float8PackingFormat(self.packing_format) if self.packing_format is not None else Float8PackingFormat.PLAIN
        if self.packing_format != Float8PackingFormat.PLAIN:
            raise NotImplementedError(f"Float8 packing format {self.packing_format.value}: 2:4 sparsity is outside SURVEY.md section 8")
        self.kernel_preference = KernelPreference(self.kernel_preference)
        if self.kernel_preference not in (KernelPreference.AUTO, KernelPreference.TORCH):
            raise NotImplementedError(f"kernel_preference {self.kernel_preference.value} names a library that is not this backend; use AUTO")
        if self.mm_config is None:
            self.mm_config = Float8MMConfig(use_fast_accum=True)


@dataclass
class Float8DynamicActivationInt4WeightConfig(AOBaseConfig):
    """float8 e4m3 rowwise dynamic activation x int4 groupwise (symmetric) weight (reference quant_api.py:630-699: group_size 128).

    `int4_packing_format`: "plain" (the default HERE) builds the PLAIN `Int4Tensor` (reference int4_tensor.py: qdata / scale / zero_point)
    carrying its gfx950 compute layout (tile-packed codes + stacked scale / zero, built once at from_hp); its state_dict IS the reference's
    PLAIN format.  The REFERENCE's default is "preshuffled" (:646): an `Int4PreshuffledTensor` (qdata / group_scale fp8 [K/g/8, 8, N] /
    row_scale [N]: two-level scales and a byte order pre-arranged for its H100 WGMMA kernel by the un-vendored `mslk.quantize_int4_preshuffle`,
    int4_preshuffled_tensor.py:55-66, 129-170).  Neither that arithmetic nor that byte order can be reproduced without mslk, so a checkpoint
    written under "preshuffled" here could not be one upstream loads: the name parses (upstream JSON decodes) and `quantize_` REFUSES it with
    the reason (round 6; rounds 4-5 accepted it as an alias of "plain", which silently produced a different checkpoint class than upstream).

    Wire format (config_to_dict): the reference's class has the single field `int4_packing_format`; `group_size` -- an extension here, the
    reference hard-codes 128 (:670) -- is written only when it is not 128, so that the default config's JSON is one upstream's
    `config_from_dict` accepts."""

    int4_packing_format: Int4PackingFormat = Int4PackingFormat.PLAIN
    group_size: int = 128
    _wire_omit_at_default = {"group_size": 128}  # (class attribute, not a field) fields the reference's class does not have

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



# ---- JSON (de)serialisation of configs (reference torchao/core/config.py:69-305: same wire format) -----------------------------------
# {"_type": <class name>, "_version": <int>, "_data": {field: value, ...}}; enums {"_type": <enum class>, "_data": <member NAME>};
# torch.dtype {"_type": "torch.dtype", "_data": "bfloat16"}; granularities and kwargs dataclasses like configs; lists and dicts element
# by element; tuples are refused (JSON would turn them into lists silently).  Only class NAMES are stored: they are resolved against this
# package's own modules, never imported from a path in the file.
def _encode(value):
    import dataclasses

    import torch

    if isinstance(value, AOBaseConfig) or (dataclasses.is_dataclass(value) and not isinstance(value, type)):
        if dataclasses.is_dataclass(value):
            items = [(f.name, getattr(value, f.name)) for f in dataclasses.fields(value)]
        else:
            items = list(vars(value).items())
        omit = getattr(type(value), "_wire_omit_at_default", {})
        return {"_type": type(value).__name__, "_version": getattr(value, "version", 1),
                "_data": {k: _encode(v) for k, v in items if k != "version" and not k.startswith("_") and not (k in omit and v == omit[k])}}
    if isinstance(value, enum.Enum):
        return {"_type": type(value).__name__, "_data": value.name}
    if isinstance(value, torch.dtype):
        return {"_type": "torch.dtype", "_data": str(value).split(".")[-1]}
    if isinstance(value, tuple):
        raise NotImplementedError(f"Tuples will be serialized as List in JSON, so we recommend to use Lists instead to avoid surprises. got: {value}")
    if isinstance(value, list):
        return [_encode(v) for v in value]
    if isinstance(value, dict):
        return {k: _encode(v) for k, v in value.items()}
    if value is None or isinstance(value, (bool, int, float, str)):
        return value
    if isinstance(value, torch.Tensor):
        raise TypeError("tensors inside a config (static activation scales) are checkpoint data, not configuration: save them with the state_dict")
    raise TypeError(f"Object of type {type(value).__name__} is not JSON serializable")


def config_to_dict(config):
    if not isinstance(config, AOBaseConfig):
        raise TypeError(f"Expected AOBaseConfig instance, got {type(config)}")
    return _encode(config)


def _resolve(name):
    import importlib

    for mod in ("ao_amd.quantization.config", "ao_amd.quantization.granularity", "ao_amd.quantization.quant_primitives",
                "ao_amd.quantization.int8_tensor", "ao_amd.quantization.float8_tensor", "ao_amd.prototype.mx"):
        try:
            cls = getattr(importlib.import_module(mod), name)
        except (ImportError, AttributeError):
            continue
        if isinstance(cls, type):
            return cls
    raise ValueError(f"Failed to find class {name} in any of the allowed modules of ao_amd")


def config_from_dict(data):
    import warnings

    if not isinstance(data, dict):
        raise TypeError(f"Expected dictionary, got {type(data)}")
    if "_type" not in data or "_data" not in data:
        raise ValueError("Input dictionary missing required '_type' or '_data' fields")
    name, payload = data["_type"], data["_data"]
    if name == "torch.dtype":
        import torch

        return getattr(torch, payload)
    cls = _resolve(name)
    if not isinstance(payload, dict):
        return getattr(cls, payload) if issubclass(cls, enum.Enum) else cls(payload)

    def dec(v):
        if isinstance(v, dict):
            return config_from_dict(v) if ("_type" in v and "_data" in v) else {k: dec(x) for k, x in v.items()}
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v

    kwargs = {k: dec(v) for k, v in payload.items()}
    stored, current = data.get("_version", 1), getattr(cls, "version", 1)
    if stored != current:
        warnings.warn(f"Stored version is not the same as current default version of the config: {stored=}, {current=}, please check the deprecation warning")
        kwargs["version"] = stored
    try:
        return cls(**kwargs)
    except Exception as e:  # noqa: BLE001
        raise ValueError(f"Failed to create instance of {cls.__name__}: {e}") from e
