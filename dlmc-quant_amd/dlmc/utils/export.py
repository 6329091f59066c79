"""Deployable integer checkpoint (the consumer of the int8 / packed-int4 code
emission).  The reference only ever saves fp32 `state_dict`s of fake-quantised models
(example/quantization/post_training_quantization.py:94-101); this stores every quantised layer's weight as its
integer codes - int8, or two 4-bit codes per byte when the format has <= 4 bits - plus the (per-channel) scale and
offset, and reconstructs the fp32 fake-quantised weights bit-exactly on load (one HIP launch per layer).

    blob = export_quantized_state(model)          # after calibration;  torch.save(blob, path)
    load_quantized_state(fresh_quantized_model, blob)
"""
import math

import torch

from .. import _native as N
from ..quantization.scalar import kernels as K
from ..quantization.scalar.FSPTQuant import FSPTQBase
from ..quantization.scalar.modules import QBase

__all__ = ["export_quantized_state", "load_quantized_state", "FORMAT_VERSION"]

FORMAT_VERSION = 1


def _weight_plan(mod):
    """(form, scale, offset, g) with which `mod` fake-quantises its weight, or None if it does not."""
    if isinstance(mod, FSPTQBase):
        if not mod.wt_quant or mod.qconfig["weight"].get("recon_type") in ("adaround", "dist_recon"):
            return None
        return N.FORM_SYMMETRIC, mod.wt_scale.detach(), None, 0.0
    if isinstance(mod, QBase):
        if not mod.qconfig["weight"]["enable"]:
            return None
        return N.FORM_QBASE, mod.wt_scale.detach(), mod.wt_offset, 1 / math.sqrt(mod.weight.numel() * mod.wt_max_val)
    return None


def export_quantized_state(model):
    """{"format", "layers": {name: {...codes, scales...}}, "other": remaining state_dict entries}."""
    layers, skip = {}, set()
    for name, mod in model.named_modules():
        plan = _weight_plan(mod)
        if plan is None:
            continue
        if float(mod.wt_init_state.reshape(-1)[0]) == 0:
            raise RuntimeError(f"{name}: export needs a calibrated model (run one forward first)")
        form, scale, offset, g = plan
        lo, hi = mod.wt_min_val, mod.wt_max_val
        packed = -8 <= lo and hi <= 15
        _, codes = K.fake_quant(mod.weight.detach(), scale, offset, lo, hi, form, g=g, codes="p4" if packed else "i8",
                                want_y=False)
        layers[name] = {"codes": codes.cpu(), "packed_int4": packed, "shape": tuple(mod.weight.shape), "form": form,
                        "lo": lo, "hi": hi, "g": g, "wt_scale": scale.cpu(),
                        "wt_offset": None if offset is None else offset.detach().cpu()}
        skip.add(name + ".weight")
    other = {k: v.detach().cpu() for k, v in model.state_dict().items() if k not in skip}
    return {"format": FORMAT_VERSION, "layers": layers, "other": other}


def load_quantized_state(model, blob):
    """Load into a model that went through the same `quantize_model` call.  Weights become the dequantised
    codes (fake-quantising them again is the identity), scales/offsets/init flags come from `other`."""
    if blob.get("format") != FORMAT_VERSION:
        raise ValueError(f"unknown integer-checkpoint format {blob.get('format')!r}")
    missing, unexpected = model.load_state_dict(blob["other"], strict=False)
    mods = dict(model.named_modules())
    want = {n + ".weight" for n in blob["layers"]}
    if set(missing) - want or unexpected:
        raise RuntimeError(f"state mismatch: missing {sorted(set(missing) - want)}, unexpected {sorted(unexpected)}")
    for name, rec in blob["layers"].items():
        mod = mods[name]
        dev = mod.weight.device
        off = None if rec["wt_offset"] is None else rec["wt_offset"].to(dev)
        w = K.dequant_codes(rec["codes"].to(dev), rec["shape"], rec["wt_scale"].to(dev), off, rec["form"],
                            "p4" if rec["packed_int4"] else "i8", rec["lo"] < 0, g=rec["g"])
        mod.weight.data.copy_(w)
    return model
