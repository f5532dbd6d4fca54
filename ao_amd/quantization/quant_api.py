"""quantize_(model, config): swap nn.Linear weights for low-bit tensor subclasses.

Mirror of torchao/quantization/quant_api.py:249-317 (quantize_),
:120-166 (_replace_with_custom_fn_if_matches_filter, _is_linear) and
torchao/quantization/transform_module.py:16-52 (handler registry).
"""
import logging
from typing import Callable, Dict, Optional, Type

import torch
import torch.nn as nn

from .config import (
    AOBaseConfig,
    Float8DynamicActivationFloat8WeightConfig,
    Float8DynamicActivationInt4WeightConfig,
    Int4ChooseQParamsAlgorithm,
    Int4PackingFormat,
    Int4WeightOnlyConfig,
    Int8DynamicActivationInt8WeightConfig,
    FqnToConfig,
    Int8StaticActivationInt8WeightConfig,
)

logger = logging.getLogger(__name__)

_QUANTIZE_CONFIG_HANDLER: Dict[Type[AOBaseConfig], Callable] = {}


def register_quantize_module_handler(config_type):
    def deco(fn):
        _QUANTIZE_CONFIG_HANDLER[config_type] = fn
        return fn

    return deco


def _is_linear(mod, *args):
    """reference quant_api.py:166: plain nn.Linear whose weight is not yet quantized"""
    from .base_tensor import LowBitTensorBase

    return (
        isinstance(mod, nn.Linear)
        and hasattr(mod, "weight")
        and not isinstance(mod.weight, LowBitTensorBase)
    )


def _replace_with_custom_fn_if_matches_filter(model, replacement_fn, filter_fn, cur_fqn="", device=None):
    if filter_fn(model, cur_fqn[:-1]):
        if device is not None:
            model.to(device=device)
        return replacement_fn(model)
    for name, child in list(model.named_children()):
        new_child = _replace_with_custom_fn_if_matches_filter(
            child, replacement_fn, filter_fn, f"{cur_fqn}{name}.", device
        )
        if new_child is not child and new_child is not None:
            setattr(model, name, new_child)
    if device is not None:
        model.to(device=device)
    return model


def quantize_(model: nn.Module, config: AOBaseConfig, filter_fn: Optional[Callable] = None, device=None):
    """Quantize the weights of every matching nn.Linear in place.

    filter_fn(module, fqn) -> bool selects modules (default: every nn.Linear).
    Raises AssertionError for configs without a registered handler, like the
    reference (quant_api.py:307-317)."""
    if isinstance(config, FqnToConfig):
        if filter_fn is not None:
            raise ValueError("Custom filter_fn and FqnToConfig were both specified. Only filter_fn=None is supported when FqnToConfig is specified.")
        return _quantize_by_fqn(model, config, device)
    filter_fn = _is_linear if filter_fn is None else filter_fn
    if not isinstance(config, AOBaseConfig):
        raise AssertionError(
            "Passing a generic Callable to `quantize_` is no longer recommended; pass an AOBaseConfig instance"
        )
    handler = _QUANTIZE_CONFIG_HANDLER.get(type(config))
    if handler is None:
        raise AssertionError(f"unexpected config type: {type(config)}")
    _replace_with_custom_fn_if_matches_filter(model, lambda m: handler(m, config), filter_fn, device=device)


def _pick_config(fqn: str, table: dict, regex_only: bool = False):
    """(found, config) for an fqn: the exact key unless regex_only, else the first "re:" key whose pattern matches it in full."""
    import re

    if not regex_only:
        if fqn in table:
            assert not fqn.startswith("re:"), f"Error: Exact match but regex {fqn} specified."
            return True, table[fqn]
        return False, None
    for key, cfg in table.items():
        if key.startswith("re:") and re.fullmatch(key[3:], fqn):
            return True, cfg
    return False, None


def _apply(module, cfg, parameter_name=None):
    if cfg is None:
        return module
    handler = _QUANTIZE_CONFIG_HANDLER.get(type(cfg))
    if handler is None:
        raise AssertionError(f"unexpected config type: {type(cfg)}")
    return handler(module, cfg) if parameter_name is None else handler(module, cfg, parameter_name=parameter_name)


def _quantize_by_fqn(model, config, device):
    """quantize_ with an FqnToConfig (reference quant_api.py:286-306, 1610-1703).  Per module: parameters named exactly, then the
    module named exactly, then parameter regexes, then module regexes, then "_default" for plain nn.Linear modules."""
    table = config.fqn_to_config
    modules = dict(model.named_modules())
    for fqn, module in modules.items():
        params = [(n, f"{fqn}.{n}" if fqn else n) for n, _ in module.named_parameters(recurse=False)]
        replacement, decided = module, False
        # 1. parameters named exactly (possibly several of one module); a None config only takes the parameter out of the regex pass
        regex_params = []
        for n, pf in params:
            found, cfg = _pick_config(pf, table)
            if found:
                decided = True
                replacement = _apply(replacement, cfg, parameter_name=n)
            else:
                regex_params.append((n, pf))  # (the reference also re-offers exactly-matched, quantized parameters to the regexes; a
                                              # second quantization of a quantized parameter is never meant, so they stay out here)
        # 2. the module named exactly -- only when no parameter was
        if not decided:
            found, cfg = _pick_config(fqn, table)
            if found:
                replacement, decided = _apply(module, cfg), True
                if device is not None:
                    replacement = replacement.to(device=device)
                if replacement is not module and fqn != "":
                    parent, _, child = fqn.rpartition(".")
                    setattr(modules[parent], child, replacement)
                continue
        # 3. parameter regexes: ALWAYS tried for the parameters step 1 left (reference :1665-1680), every matching pattern in table order
        #    (a None pattern marks the parameter as handled without quantizing it; a later non-None pattern still applies)
        import re
        for n, pf in regex_params:
            for key, cfg in table.items():
                if key.startswith("re:") and re.fullmatch(key[3:], pf):
                    decided = True
                    replacement = _apply(replacement, cfg, parameter_name=n)
        # 4. module regexes -- only when nothing above matched: the first full match decides
        if not decided:
            found, cfg = _pick_config(fqn, table, regex_only=True)
            if found:
                replacement, decided = _apply(module, cfg), True
        if not decided and "_default" in table and _is_linear(module):
            replacement, decided = _apply(module, table["_default"]), True
        if decided and device is not None:
            replacement = replacement.to(device=device)
        if decided and replacement is not module and fqn != "":
            parent, _, child = fqn.rpartition(".")
            setattr(modules[parent], child, replacement)
    return model


def _int4_weight_only_quantize_tensor(weight, config):
    """reference quant_api.py:538-594"""
    from .int4_tensor import Int4TilePackedTo4dTensor

    group_size = config.group_size
    if weight.shape[-1] % group_size != 0:
        logger.info(
            f"Skipping quantizing weight with int4 weight only quantization because the shape of weight {weight.shape} is not compatible with group_size {group_size}"
        )
        return weight
    block_size = [1 for _ in range(weight.ndim - 1)] + [group_size]
    assert config.version == 2
    hqq = config.int4_choose_qparams_algorithm == Int4ChooseQParamsAlgorithm.HQQ
    if hqq and config.int4_packing_format != Int4PackingFormat.TILE_PACKED_TO_4D:
        # reference quant_api.py:560-565
        raise AssertionError(
            f"Int4ChooseQParamsAlgorithm.HQQ is not supported by packing format {config.int4_packing_format}, "
            f"it's only supported by Int4PackingFormat.TILE_PACKED_TO_4D currently"
        )
    if config.int4_packing_format == Int4PackingFormat.TILE_PACKED_TO_4D:
        return Int4TilePackedTo4dTensor.from_hp(weight, block_size, int4_choose_qparams_algorithm=config.int4_choose_qparams_algorithm,
                                                ntile_size=config.int4_tile_packed_ntile)
    if config.int4_packing_format == Int4PackingFormat.PLAIN:
        from .int4_plain_tensor import Int4Tensor

        return Int4Tensor.from_hp(weight, block_size)
    raise ValueError(
        f"Unsupported int4 packing format on MI355X: {config.int4_packing_format} "
        "(plain and tile_packed_to_4d are implemented; preshuffled is an H100 WGMMA layout, plain_int32 is XPU / NPU)"
    )


@register_quantize_module_handler(Int4WeightOnlyConfig)
def _int4_weight_only_transform(module, config, *, parameter_name="weight"):
    assert hasattr(module, parameter_name), (
        f"applying int4 weight only quant requires module to have {parameter_name} attribute but {module} does not have one"
    )
    new_weight = _int4_weight_only_quantize_tensor(getattr(module, parameter_name), config)
    setattr(module, parameter_name, nn.Parameter(new_weight, requires_grad=False))
    return module


@register_quantize_module_handler(Float8DynamicActivationInt4WeightConfig)
def _float8_dynamic_activation_int4_weight_transform(module, config, *, parameter_name="weight"):
    """reference quant_api.py:660-699: Int4Tensor with activation_dtype float8_e4m3fn; shapes whose K is not a multiple of the
    group size are left unquantized.  "plain" gives the Int4Tensor that carries its gfx950 compute layout; "preshuffled" (the reference's
    default: Int4PreshuffledTensor, :672-676) is a checkpoint format only the un-vendored mslk can write -- refused with the reason (config.py)."""
    from .int4_plain_tensor import Int4Tensor

    assert hasattr(module, parameter_name), (
        f"applying float8 dynamic activation int4 weight quant requires module to have {parameter_name} attribute"
    )
    weight = getattr(module, parameter_name)
    # reference :660-669: "only preshuffled and plain int4_packing_format supported right now"
    assert config.int4_packing_format in (Int4PackingFormat.PRESHUFFLED, Int4PackingFormat.PLAIN), (
        f"only preshuffled and plain int4_packing_format supported right now, got: {config.int4_packing_format}"
    )
    if config.int4_packing_format == Int4PackingFormat.PRESHUFFLED:
        raise NotImplementedError(
            'Float8DynamicActivationInt4WeightConfig(int4_packing_format="preshuffled") would have to write an Int4PreshuffledTensor checkpoint '
            "(qdata / group_scale / row_scale in the byte order of mslk.quantize_int4_preshuffle, an H100 WGMMA layout); mslk is not available, so "
            'such a checkpoint could not be one upstream torchao loads.  Use int4_packing_format="plain" (the default of this backend): the same '
            "fp8-activation x int4-weight linear, saved in the reference's PLAIN Int4Tensor format")
    if weight.shape[-1] % config.group_size != 0:
        logger.info(f"Skipping quantizing weight of shape {weight.shape}: not compatible with group_size {config.group_size}")
        return module
    block_size = [1 for _ in range(weight.ndim - 1)] + [config.group_size]
    new_weight = Int4Tensor.from_hp(weight, block_size, activation_dtype=torch.float8_e4m3fn)
    setattr(module, parameter_name, nn.Parameter(new_weight, requires_grad=False))
    return module


@register_quantize_module_handler(Int8DynamicActivationInt8WeightConfig)
def _int8_dynamic_activation_int8_weight_transform(module, config, *, parameter_name="weight"):
    """reference quant_api.py:885-960: per-row symmetric int8 weight, per-token symmetric dynamic
    int8 activation (version 2 -> Int8Tensor)."""
    from .int8_tensor import Int8Tensor, QuantizeTensorToInt8Kwargs

    assert hasattr(module, parameter_name), (
        f"applying int8 dynamic activation int8 weight quant requires module to have {parameter_name} attribute"
    )
    weight = getattr(module, parameter_name)
    act_granularity, weight_granularity = config.granularity
    new_weight = Int8Tensor.from_hp(
        weight,
        granularity=weight_granularity,
        act_quant_kwargs=QuantizeTensorToInt8Kwargs(granularity=act_granularity, mapping_type=config.act_mapping_type),
    )
    setattr(module, parameter_name, nn.Parameter(new_weight, requires_grad=False))
    return module


@register_quantize_module_handler(Int8StaticActivationInt8WeightConfig)
def _int8_static_activation_int8_weight_transform(module, config, *, parameter_name="weight"):
    """reference quant_api.py:970-1012: int8 weight; the activation is cast with the calibrated scale (and zero-point) of the config."""
    from .int8_tensor import Int8Tensor

    assert hasattr(module, parameter_name), f"Expected module to have attribute `{parameter_name}` but not found"
    assert config.act_quant_scale is not None, "Int8StaticActivationInt8WeightConfig needs act_quant_scale"
    act_granularity, weight_granularity = config.granularity
    zp = None if config.act_quant_zero_point is None else config.act_quant_zero_point.detach()
    new_weight = Int8Tensor.from_hp(
        getattr(module, parameter_name),
        granularity=weight_granularity,
        act_quant_kwargs=config.get_act_quant_kwargs(),
        act_quant_scale=config.act_quant_scale.detach(),
        act_quant_zero_point=zp,
    )
    setattr(module, parameter_name, nn.Parameter(new_weight, requires_grad=False))
    return module


def _fp8_mm_compat(weight: torch.Tensor) -> bool:
    """reference quantization/utils.py:663-687: _scaled_mm needs both dims divisible by 16"""
    assert weight.dim() in [2, 3], f"float8 quantization only works for 2/3-D tensors, got {weight.dim()}D tensor"
    out_dim, in_dim = weight.shape[-2], weight.shape[-1]
    if (in_dim % 16 != 0) or (out_dim % 16 != 0):
        logger.info(
            f"Skipping float8 quantization: weight shape {weight.shape} is not compatible with _scaled_mm. "
            f"Both input dimension ({in_dim}) and output dimension ({out_dim}) must be multiples of 16. "
        )
        return False
    return True


@register_quantize_module_handler(Float8DynamicActivationFloat8WeightConfig)
def _float8_dynamic_activation_float8_weight_transform(module, config, *, parameter_name="weight"):
    """reference quant_api.py:1196-1297 (PerRow branch): unsupported shapes are left unquantized."""
    from .float8_tensor import Float8Tensor, QuantizeTensorToFloat8Kwargs

    assert hasattr(module, parameter_name), (
        f"applying float8 dynamic activation quant requires module to have {parameter_name} attribute"
    )
    weight = getattr(module, parameter_name)
    if not _fp8_mm_compat(weight):
        return module
    act_granularity, weight_granularity = config.granularity
    new_weight = Float8Tensor.from_hp(
        weight,
        granularity=weight_granularity,
        act_quant_kwargs=QuantizeTensorToFloat8Kwargs(granularity=act_granularity, hp_value_lb=config.activation_value_lb,
                                                      hp_value_ub=config.activation_value_ub),
    )
    setattr(module, parameter_name, nn.Parameter(new_weight, requires_grad=False))
    return module
