"""`quantize_model`: swap nn.Conv2d / nn.Linear for the HIP-backed quant wrappers, in place
(reference: dlmc/utils/quantize.py:61-143 - same signature, same config schema, same swap mechanism:
`__new__` + `__dict__.update` + `initialize`, so parameters, hooks and buffers of the original layer
carry over and `state_dict` keys stay what the reference's checkpoints expect)."""
import copy
from operator import attrgetter
from typing import Dict

from torch import nn

from ..quantization.scalar import FSPTQuant as FSPQ
from ..quantization.scalar import RootQ as RQ
from ..quantization.scalar import modules as qnn
from .access import attrsetter, get_layers

__all__ = ["quantize_model"]

MODULE_MAPPING = {nn.Conv2d: qnn.QConv2d, nn.Linear: qnn.QLinear}
ROOTQ_MAPPING = {nn.Conv2d: RQ.RootQConv2d, nn.Linear: RQ.RootQLinear}
FSPTQUANT_MAPPING = {nn.Conv2d: FSPQ.FSPTQConv2d, nn.Linear: FSPQ.FSPTQLinear}
_FAMILIES = {None: MODULE_MAPPING, "RootQ": ROOTQ_MAPPING, "FSPTQ": FSPTQUANT_MAPPING}


def _override_options(dst_config: Dict, src_config: Dict = None) -> Dict:
    """Per-layer override of {type, enable, args} (reference: quantize.py:44-58)."""
    if src_config is None:
        return dst_config
    merged = copy.deepcopy(dst_config)
    for key in ("type", "enable"):
        if key in src_config:
            merged[key] = src_config[key]
    if "args" in src_config:
        merged["args"].update(src_config["args"])
    return merged


def quantize_model(model: nn.Module, config: Dict, logger=None, quantization_type: str = None, **kwargs) -> None:
    """Quantise `model` in place.

    :param config: the `quantization:` section of the YAML (weight / input / exclude_layers /
                   override_options [/ momentum])
    :param quantization_type: None (QBase family), "RootQ" or "FSPTQ".  "BitMixer" and "MetaQ" name
                   packages that are missing from the reference itself (quantize.py:10,12).
    """
    if quantization_type in ("BitMixer", "MetaQ"):
        raise NotImplementedError(f"quantization_type={quantization_type!r}: its package is absent from the reference")
    mapping = _FAMILIES.get(quantization_type, MODULE_MAPPING)
    momentum = config["momentum"] if quantization_type == "RootQ" else 0.1

    candidates = get_layers(model, filter_types=tuple(mapping.keys()))
    excluded = set()
    for regexp in config.get("exclude_layers") or []:
        excluded.update(get_layers(model, filter_regexp=regexp))

    overrides = {}
    for opt in config.get("override_options") or []:
        for regexp in opt.get("layers") or []:
            for name in get_layers(model, filter_regexp=regexp):
                assert name not in overrides, f"layer {name} is overridden twice"
                overrides[name] = opt["options"]

    for name in candidates:
        if name in excluded:
            continue
        layer = attrgetter(name)(model)
        if type(layer) not in mapping:     # an already-quantised subclass, or a foreign subclass
            continue
        weight_cfg, input_cfg = config["weight"], config["input"]
        if name in overrides:
            weight_cfg = _override_options(weight_cfg, overrides[name].get("weight"))
            input_cfg = _override_options(input_cfg, overrides[name].get("input"))
        layer_cfg = {"input": copy.deepcopy(input_cfg), "weight": copy.deepcopy(weight_cfg), "momentum": momentum}
        if "int8_gemm" in kwargs or "int8_gemm" in config:   # opt-in fused int8 conv/linear (FSPTQ family)
            layer_cfg["int8_gemm"] = bool(kwargs.get("int8_gemm", config.get("int8_gemm")))
        cls = mapping[type(layer)]
        wrapped = cls.__new__(cls)
        wrapped.__dict__.update(layer.__dict__)
        wrapped.initialize(layer_cfg)
        attrsetter(name)(model, wrapped)
        if logger is not None:
            logger.info("Quantize module {} with method <input: {}> <weight: {}>".format(
                name, layer_cfg["input"], layer_cfg["weight"]))
