"""Minimal tensor-subclass base for quantized weights.

Mirrors the contract of the reference's TorchAOBaseTensor (torchao/utils.py:720:
`tensor_data_names` / `tensor_attribute_names` / `optional_tensor_data_names`,
`implements` + `implements_torch_function` dispatch tables, flatten/unflatten for
state_dict and torch.compile) with an independent, much smaller implementation.
"""
from typing import Any, Callable, Dict, List

import torch

aten = torch.ops.aten


class LowBitTensorBase(torch.Tensor):
    tensor_data_names: List[str] = []
    tensor_attribute_names: List[str] = []
    optional_tensor_data_names: List[str] = []

    _ATEN_TABLE: Dict[Any, Callable]
    _TORCH_FN_TABLE: Dict[Any, Callable]

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        cls._ATEN_TABLE = {}
        cls._TORCH_FN_TABLE = {}
        _register_common(cls)

    # ---- registration decorators (reference: utils.py:411-477) -------------
    @classmethod
    def implements(cls, ops):
        ops = ops if isinstance(ops, (list, tuple)) else [ops]

        def deco(fn):
            for op in ops:
                cls._ATEN_TABLE[op] = fn
            return fn

        return deco

    @classmethod
    def implements_torch_function(cls, fns):
        fns = fns if isinstance(fns, (list, tuple)) else [fns]

        def deco(fn):
            for f in fns:
                cls._TORCH_FN_TABLE[f] = fn
            return fn

        return deco

    # ---- dispatch (reference: utils.py:659-697) ------------------------------
    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        impl = cls._TORCH_FN_TABLE.get(func)
        if impl is not None:
            return impl(func, types, args, kwargs)
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **kwargs)

    @classmethod
    def __torch_dispatch__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        impl = cls._ATEN_TABLE.get(func)
        if impl is not None:
            return impl(func, types, args, kwargs)
        raise NotImplementedError(
            f"{cls.__name__} dispatch: attempting to run unimplemented operator/function: {func}"
        )

    # ---- data plumbing --------------------------------------------------------
    def _data_names(self):
        names = list(self.tensor_data_names)
        for n in self.optional_tensor_data_names:
            if getattr(self, n, None) is not None:
                names.append(n)
        return names

    def __tensor_flatten__(self):
        return self._data_names(), [getattr(self, a) for a in self.tensor_attribute_names]

    @classmethod
    def __tensor_unflatten__(cls, tensor_data_dict, tensor_attributes, outer_size, outer_stride):
        data = [tensor_data_dict[n] for n in cls.tensor_data_names]
        opt = {n: tensor_data_dict.get(n, None) for n in cls.optional_tensor_data_names}
        return cls(*data, *tensor_attributes, **opt)

    def _apply_fn_to_data(self, fn):
        data = [fn(getattr(self, n)) for n in self.tensor_data_names]
        attrs = [getattr(self, a) for a in self.tensor_attribute_names]
        opt = {
            n: (fn(getattr(self, n)) if getattr(self, n, None) is not None else None)
            for n in self.optional_tensor_data_names
        }
        return self.__class__(*data, *attrs, **opt)

    def __repr__(self):
        return f"{self.__class__.__name__}({self._quantization_type()})"

    def _quantization_type(self):
        return f"shape={tuple(self.shape)}, device={self.device}"


def _register_common(cls):
    """detach / clone / alias / _to_copy / contiguous on every subclass
    (reference: TorchAOBaseTensor common ops, utils.py:480-636)."""

    @cls.implements([aten.detach.default, aten.alias.default])
    def _(func, types, args, kwargs):
        return args[0]._apply_fn_to_data(lambda t: t.detach())

    @cls.implements(aten.clone.default)
    def _(func, types, args, kwargs):
        return args[0]._apply_fn_to_data(lambda t: t.clone())

    @cls.implements(aten.contiguous.default)
    def _(func, types, args, kwargs):
        return args[0]._apply_fn_to_data(lambda t: t.contiguous())

    @cls.implements(aten._to_copy.default)
    def _(func, types, args, kwargs):
        device = kwargs.get("device", None)
        # dtype changes are not meaningful for packed data; only move devices
        return args[0]._apply_fn_to_data(lambda t: t.to(device=device) if device is not None else t.clone())

    @cls.implements(aten.copy_.default)
    def _(func, types, args, kwargs):
        dst, src = args[0], args[1]
        if type(dst) is not type(src) or dst.shape != src.shape:
            raise ValueError(
                f"Not supported args for copy_ due to metadata mismatch: {type(dst)} {tuple(dst.shape)} vs {type(src)} {tuple(src.shape)}"
            )
        for n in dst._data_names():
            getattr(dst, n).copy_(getattr(src, n))
        return dst
