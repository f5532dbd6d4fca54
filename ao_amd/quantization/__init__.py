from .config import (  # noqa: F401
    AOBaseConfig,
    Float8MMConfig,
    Float8PackingFormat,
    KernelPreference,
    config_from_dict,
    config_to_dict,
    Float8DynamicActivationFloat8WeightConfig,
    Float8DynamicActivationInt4WeightConfig,
    FqnToConfig,
    Int4ChooseQParamsAlgorithm,
    Int4PackingFormat,
    Int4WeightOnlyConfig,
    Int8DynamicActivationInt8WeightConfig,
    Int8StaticActivationInt8WeightConfig,
    ModuleFqnToConfig,
)
from .granularity import PerGroup, PerRow, PerTensor  # noqa: F401
from .quant_primitives import MappingType  # noqa: F401
from .float8_tensor import Float8Tensor, QuantizeTensorToFloat8Kwargs  # noqa: F401
from .int4_plain_tensor import Int4Tensor  # noqa: F401
from .int4_tensor import Int4TilePackedTo4dTensor  # noqa: F401
from .int8_tensor import Int8Tensor, QuantizeTensorToInt8Kwargs  # noqa: F401
from .quant_api import quantize_  # noqa: F401
