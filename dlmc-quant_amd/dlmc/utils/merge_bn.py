"""BatchNorm folding (reference: dlmc/utils/merge_bn.py:13-113), the step before `quantize_model` in the
few-shot PTQ flow (example/quantization/FSPTQuant.py:67).  Same signature and name-mapping rules; the
per-channel arithmetic is one HIP launch per layer (`dlmcq_fold_bn_f32`), bit-identical to the reference's CPU
result.  The model must already be on the GPU."""
from copy import deepcopy
from operator import attrgetter
from types import FunctionType

import torch
from torch.nn import BatchNorm2d, Conv2d, Identity, Module, Parameter

from .. import _native as N
from .access import attrsetter, get_layers

__all__ = ["DEFAULT_BN_MAPPING_FN", "DEFAULT_CONV_MAPPING_FN", "merge_bn", "fold_bn_"]


def _sibling(name, numeric_step, old, new):
    *parent, base = name.split(".")
    if base.isdecimal():
        return ".".join(parent + [str(int(base) + numeric_step)])
    if old in base:
        return ".".join(parent + [base.replace(old, new)])
    return None


def DEFAULT_CONV_MAPPING_FN(bn_name: str):
    """`layer1.conv1.1` -> `layer1.conv1.0`;  `layer1.bn1` -> `layer1.conv1`."""
    return _sibling(bn_name, -1, "bn", "conv")


def DEFAULT_BN_MAPPING_FN(conv_name: str):
    """`layer1.conv1.0` -> `layer1.conv1.1`;  `layer1.conv1` -> `layer1.bn1`."""
    return _sibling(conv_name, +1, "conv", "bn")


def fold_bn_(conv: Conv2d, bn: BatchNorm2d, var_eps: float = 1e-7):
    """Fold `bn` into `conv` in place.  The reference adds 1e-7 to the running variance (merge_bn.py:88),
    not `bn.eps`; that is kept."""
    w = conv.weight.data
    N.require_gpu(w)
    cout = w.shape[0]
    if conv.bias is None:
        conv.bias = Parameter(torch.zeros(cout, device=w.device))
    if not w.is_contiguous():
        raise ValueError("merge_bn needs a contiguous conv weight")
    args = [t.data.contiguous().float() for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var)]
    N.check(N.lib.dlmcq_fold_bn_f32(N.ptr(w), N.ptr(conv.bias.data), *[N.ptr(t) for t in args], cout,
                                    w.numel() // cout, float(var_eps), N.stream_ptr()))


def merge_bn(model: Module, mapping_fn: FunctionType = DEFAULT_CONV_MAPPING_FN, inplace: bool = False,
             allow_missing: bool = False, bitmixer_func: bool = False) -> Module:
    """Merge every `BatchNorm2d` into the `Conv2d` that `mapping_fn` names and replace it by `Identity`.
    As in the reference, `inplace=True` works on a deep copy (sic, merge_bn.py:60-61) and the merged model
    is returned."""
    if bitmixer_func:
        raise NotImplementedError("BitMixer is absent from the reference (merge_bn.py:7)")
    if inplace:
        model = deepcopy(model)
    names = get_layers(model, filter_types=(Conv2d, BatchNorm2d))
    for name in names:
        bn = attrgetter(name)(model)
        if not isinstance(bn, BatchNorm2d):
            continue
        target = mapping_fn(name)
        if target is None or target not in names:
            msg = f"[MergeBN] Could not find Conv2d that match {name}"
            if not allow_missing:
                raise ValueError(msg)
            print(msg)
            continue
        fold_bn_(attrgetter(target)(model), bn)
        attrsetter(name)(model, Identity())
    return model
