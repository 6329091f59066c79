"""Elementwise fake-quant primitives with the reference's names and argument meaning
(reference: dlmc/quantization/scalar/utils.py:1-37), each ONE HIP kernel launch instead of 2-6
ATen passes.  GPU tensors only; results are bit-identical to the reference's CPU arithmetic."""
import torch

from ... import _native as N
from . import kernels as K
from ._wrapper import FakeQuantFn

__all__ = ["quantize", "dequantize", "emulate_quantize", "get_qrange", "grad_scale", "round_pass", "floor_pass"]


def _as_dev(v, like):
    if isinstance(v, torch.Tensor):
        return v if (v.device == like.device and v.dtype == torch.float32) else v.to(like.device, torch.float32)
    return torch.full((), float(v), dtype=torch.float32, device=like.device)


def quantize(tensor, scale, offset, min_val, max_val):
    """round((t - o) / (s + 1e-7)).clamp(lo, hi) - fp32 tensor of integer codes (utils.py:1-2)."""
    scale, offset = _as_dev(scale, tensor), _as_dev(offset, tensor)
    return K.fake_quant(tensor.detach(), scale.detach(), offset, min_val, max_val, N.FORM_EMULATE, y_kind=N.Y_CODES)


def dequantize(tensor_q, scale, offset):
    """q * s + o (utils.py:5-6)."""
    scale, offset = _as_dev(scale, tensor_q), _as_dev(offset, tensor_q)
    return K.dequant(tensor_q.detach(), scale.detach(), offset)


def emulate_quantize(tensor, scale, offset, min_val, max_val):
    """dequantize(quantize(t)) in one pass (utils.py:9-11).  Differentiable like the reference's
    chain (round has zero gradient, so only `scale`/`offset` receive any)."""
    scale, offset = _as_dev(scale, tensor), _as_dev(offset, tensor)
    if torch.is_grad_enabled() and (tensor.requires_grad or scale.requires_grad):
        return FakeQuantFn.apply(tensor, scale, offset, min_val, max_val, N.FORM_EMULATE, 0.0)
    return K.fake_quant(tensor, scale, offset, min_val, max_val, N.FORM_EMULATE)


def get_qrange(signed, n_bits):
    """Integer grid of a format: signed is symmetric, -(2^(b-1)-1)..2^(b-1)-1 (utils.py:14-22)."""
    if signed:
        top = 2 ** (n_bits - 1) - 1
        return -top, top
    return 0, 2 ** n_bits - 1


# Straight-through helpers on tiny (scale-sized) tensors: plain torch ops are the right tool - these are
# O(channels) elements, not the activation.  On full-size tensors the wrappers use the fused kernels.
def grad_scale(x, scale):
    """Value x, gradient x*scale (utils.py:24-27)."""
    scaled = x * scale
    return (x - scaled).detach() + scaled


def round_pass(x):
    """Value round(x), gradient identity (utils.py:29-32)."""
    return (x.round() - x).detach() + x


def floor_pass(x):
    """Value floor(x), gradient identity (utils.py:34-37)."""
    return (x.floor() - x).detach() + x
