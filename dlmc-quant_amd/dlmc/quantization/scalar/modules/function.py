"""Explicit autograd Functions of the reference (modules/function.py:9-71).  They are not on the
reference's live path (its call sites are commented out, modules/base.py:104,138); they are kept for
API parity.  Forward = the EMULATE form in one HIP launch; backward = the reference's closed forms."""
import torch

from .... import _native as N
from .. import kernels as K


def _emulate(weight, scale, offset, lo, hi):
    offset = torch.as_tensor(offset, dtype=torch.float32, device=weight.device)
    return K.fake_quant(weight.detach(), scale.detach(), offset, lo, hi, N.FORM_EMULATE)


class FunLSQ(torch.autograd.Function):
    """LSQ: mask gradient for the weight, closed-form scale gradient (function.py:29-49)."""

    @staticmethod
    def forward(ctx, weight, scale, offset, min_val, max_val, g):
        ctx.save_for_backward(weight, scale)
        ctx.other = g, min_val, max_val
        return _emulate(weight, scale, offset, min_val, max_val)

    @staticmethod
    def backward(ctx, grad_out):
        weight, scale = ctx.saved_tensors
        g, lo, hi = ctx.other
        v = weight / scale
        below, above = (v < lo).float(), (v > hi).float()
        middle = 1.0 - below - above
        gs = ((lo * below + hi * above + middle * (v.round() - v)) * grad_out).sum().unsqueeze(0) * g
        return middle * grad_out, gs, None, None, None, None


class FunUniformQ(torch.autograd.Function):
    """function.py:9-27 as intended (the reference's backward unpacks 3 values from a 2-tuple and
    cannot run): gradient passes inside the clamp range, no scale gradient."""

    @staticmethod
    def forward(ctx, weight, scale, offset, min_val, max_val):
        ctx.save_for_backward(weight, scale)
        ctx.other = min_val, max_val
        return _emulate(weight, scale, offset, min_val, max_val)

    @staticmethod
    def backward(ctx, grad_out):
        weight, scale = ctx.saved_tensors
        lo, hi = ctx.other
        v = weight / scale
        middle = 1.0 - (v <= lo).float() - (v >= hi).float()
        return middle * grad_out, None, None, None, None


class FunRootQ(torch.autograd.Function):
    """function.py:51-63: straight-through."""

    @staticmethod
    def forward(ctx, weight, scale, offset, min_val, max_val):
        return _emulate(weight, scale, offset, min_val, max_val)

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None, None, None, None


class FunLQ(torch.autograd.Function):
    """function.py:64-71: identity both ways."""

    @staticmethod
    def forward(ctx, weight, scale, offset, min_val, max_val, g):
        return weight.view_as(weight)

    @staticmethod
    def backward(ctx, grad_out):
        return grad_out, None, None, None, None, None
