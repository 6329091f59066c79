"""RootQBase: the RootQ QAT wrapper (reference: dlmc/quantization/scalar/RootQ/base.py).

    activations  clip to [0, s*(hi-lo)],  x' = R(x_c / s) * s                       base.py:106-111
    weights      clip to [L, U], D = (U-L)/(hi-lo), i = floor((w-L)/D),
                 W' = ((sgn(phi)+1)/2 + i)*D + L       (phi only shapes the gradient) base.py:146-155
    train mode   EMA of s, U, L with momentum and the g-scaled gradient path          base.py:92-101,131-142

Forward values come from two HIP kernels (one launch each); the backward recomputes the reference's own
op chain on device under autograd, so gradients w.r.t. weight / in_scale / wt_upper / wt_lower / wt_alpha
are the reference's.  State_dict keys and 0-dim shapes follow the reference.
"""
import math

import torch
from torch.nn import Module

from .... import _native as N
from .. import kernels as K
from .._wrapper import InitState, _ste, allreduce_minmax, sync_enabled
from ..utils import get_qrange
from .function import clipping, dequantize, sgn, torch_phi_function


def _act_composite(x, run_scale, lo, hi):
    upper = run_scale * (hi - lo)
    return _ste(torch.round, clipping(x, upper, 0) / run_scale) * run_scale


def _weight_composite(w, upper, lower, alpha, lo, hi):
    wc = clipping(w, upper, lower)
    delta = (upper - lower) / (hi - lo)
    interval = _ste(torch.floor, (wc - lower) / delta)
    mi = (interval + 0.5) * delta + lower
    return dequantize(sgn(torch_phi_function(wc, mi.detach(), alpha, delta)), lower, delta, interval)


class _RootQActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, run_scale, lo, hi):
        ctx.save_for_backward(x, run_scale)
        ctx.rng = (lo, hi)
        return K.fake_quant(x, run_scale.detach().reshape(1), None, lo, hi, N.FORM_ROOTQ_ACT)

    @staticmethod
    def backward(ctx, gy):
        x, s = ctx.saved_tensors
        lo, hi = ctx.rng
        if x.is_contiguous() and gy.is_contiguous():      # one fused pass (12 B/element) instead of the op chain's ~10
            gx, gs = K.fake_quant_backward(x, gy, s.detach().reshape(1), None, lo, hi, 0.0, want_gx=ctx.needs_input_grad[0],
                                           want_gscale=ctx.needs_input_grad[1], form=N.FORM_ROOTQ_ACT)
            return gx, (gs.reshape(s.shape) if gs is not None else None), None, None
        with torch.enable_grad():
            xr = x.detach().requires_grad_(ctx.needs_input_grad[0])
            sr = s.detach().requires_grad_(ctx.needs_input_grad[1])
            y = _act_composite(xr, sr, lo, hi)
            ins = [t for t, n in ((xr, ctx.needs_input_grad[0]), (sr, ctx.needs_input_grad[1])) if n]
            grads = list(torch.autograd.grad(y, ins, gy, allow_unused=True))
        gx = grads.pop(0) if ctx.needs_input_grad[0] else None
        gs = grads.pop(0) if ctx.needs_input_grad[1] else None
        return gx, gs, None, None


class _RootQWeightFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, upper, lower, alpha, lo, hi):
        ctx.save_for_backward(w, upper, lower, alpha)
        ctx.rng = (lo, hi)
        return K.rootq_weight(w, upper, lower, lo, hi)

    @staticmethod
    def backward(ctx, gy):
        lo, hi = ctx.rng
        need = ctx.needs_input_grad[:4]
        w, upper, lower, alpha = ctx.saved_tensors
        if gy.is_contiguous() and w.is_contiguous() and all(t.numel() == 1 for t in (upper, lower, alpha)):
            gw, gu, gl, ga = K.rootq_weight_backward(w, gy, upper, lower, alpha, lo, hi, want_gw=need[0])   # 2 launches, not ~35
            outs = (gw, gu.reshape(upper.shape), gl.reshape(lower.shape), ga.reshape(alpha.shape))
            return (*[o if n else None for o, n in zip(outs, need)], None, None)
        with torch.enable_grad():
            leaves = [t.detach().requires_grad_(n) for t, n in zip(ctx.saved_tensors, need)]
            y = _weight_composite(*leaves, lo, hi)
            ins = [t for t, n in zip(leaves, need) if n]
            grads = list(torch.autograd.grad(y, ins, gy, allow_unused=True))
        out = [grads.pop(0) if n else None for n in need]
        return (*out, None, None)


class RootQBase(Module):
    qconfig: dict

    def __init__(self, qconfig: dict = None):
        self.initialize(qconfig)

    def initialize(self, qconfig):
        self.qconfig = qconfig
        self.wt_min_val, self.wt_max_val = get_qrange(qconfig["weight"]["args"]["signed"],
                                                      qconfig["weight"]["args"]["n_bits"])
        self.in_min_val, self.in_max_val = get_qrange(qconfig["input"]["args"]["signed"],
                                                      qconfig["input"]["args"]["n_bits"])
        dev = self.weight.device

        def scalar(v):
            return torch.tensor(float(v), device=dev)
        self.register_parameter("in_scale", torch.nn.Parameter(scalar(1.0)))
        self.register_buffer("in_offset", None)
        self.register_buffer("in_run_upper", scalar(0.0))
        self.register_buffer("in_run_scale", scalar(0.0))
        self.register_buffer("in_init_state", scalar(0.0))
        self.register_parameter("wt_upper", torch.nn.Parameter(scalar(2 ** 2 - 1)))
        self.register_parameter("wt_lower", torch.nn.Parameter(scalar(-(2 ** 2))))
        self.register_parameter("wt_alpha", torch.nn.Parameter(scalar(1.0 / 4)))
        self.register_buffer("wt_offset", None)
        self.register_buffer("wt_run_upper", scalar(0.0))
        self.register_buffer("wt_run_lower", scalar(0.0))
        self.register_buffer("wt_init_state", scalar(0.0))
        self.momentum = qconfig["momentum"]
        self._init = InitState()

    def reset_qparams(self):
        self._init.mark(self, "in_init_state", False)
        self._init.mark(self, "wt_init_state", False)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # checkpoints written by the reference may hold [1]-shaped bounds (base_trainer.py:209-211)
        for name in ("in_scale", "wt_upper", "wt_lower", "wt_alpha", "in_run_scale", "in_run_upper",
                     "wt_run_upper", "wt_run_lower", "in_init_state", "wt_init_state"):
            key = prefix + name
            if key in state_dict and state_dict[key].dim() == 1 and state_dict[key].numel() == 1:
                state_dict[key] = state_dict[key].reshape(())
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._init.invalidate()

    def _forward_func(self, input, weight):
        raise NotImplementedError

    # ---------------------------------------------------------------------- initialisation
    def _init_input(self, input):
        """in_scale = (max - min)/(hi - lo) (base.py:80) - one read, true division on device, and the
        global batch under data parallelism."""
        import torch.distributed as dist
        vmax, nmin = K.minmax(input.detach(), mode=N.MINMAX_NEGMIN)
        if sync_enabled() and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            vmax, nmin = allreduce_minmax(vmax.reshape(1), nmin.reshape(1))
        scale = K.span_scale(vmax.reshape(1), nmin.reshape(1), float(self.in_max_val - self.in_min_val)).reshape(())
        self.in_scale.data.copy_(scale)
        self.in_run_scale.data.copy_(scale)
        self._init.mark(self, "in_init_state")

    def _init_weight(self):
        """+-2*mean|W|*sqrt(Qp) (base.py:115-116): a one-off device reduction."""
        mean_abs = self.weight.detach().abs().mean()
        wt_max = 2 * mean_abs * math.sqrt(self.wt_max_val)
        wt_min = -2 * mean_abs * math.sqrt(self.wt_max_val)
        self.wt_upper.data.copy_(wt_max)
        self.wt_lower.data.copy_(wt_min)
        self.wt_run_upper.data.copy_(wt_max)
        self.wt_run_lower.data.copy_(wt_min)
        self._init.mark(self, "wt_init_state")

    def _ema(self, run, param, g):
        r = run.mul(1 - self.momentum).add(self.momentum * param)
        return g * r + (1 - g) * r.detach()

    # ----------------------------------------------------------------------------- forward
    def forward(self, input):
        N.require_gpu(input, self.weight)
        grad_on = torch.is_grad_enabled()
        if self.qconfig["input"]["enable"]:
            if not self._init.ready(self, "in_init_state"):
                self._init_input(input)
            if self.training:
                g_i = 1 / math.sqrt(input.numel() * self.in_max_val)
                run_scale = self._ema(self.in_run_scale, self.in_scale, g_i)
                self.in_run_scale.copy_(run_scale.data.detach())
            else:
                run_scale = self.in_run_scale
            if grad_on and (input.requires_grad or run_scale.requires_grad):
                input = _RootQActFn.apply(input, run_scale, self.in_min_val, self.in_max_val)
            else:
                input = K.fake_quant(input, run_scale.detach().reshape(1), None, self.in_min_val, self.in_max_val,
                                     N.FORM_ROOTQ_ACT)
        weight_q = self.weight
        if self.qconfig["weight"]["enable"]:
            if not self._init.ready(self, "wt_init_state"):
                self._init_weight()
            if self.training:
                g_w = 1 / math.sqrt(self.weight.numel() * self.wt_max_val)
                run_up = self._ema(self.wt_run_upper, self.wt_upper, g_w)
                run_lo = self._ema(self.wt_run_lower, self.wt_lower, g_w)
                self.wt_run_upper.copy_(run_up.data)
                self.wt_run_lower.copy_(run_lo.data)
            else:
                run_up, run_lo = self.wt_run_upper, self.wt_run_lower
            if grad_on and (self.weight.requires_grad or run_up.requires_grad or self.wt_alpha.requires_grad):
                weight_q = _RootQWeightFn.apply(self.weight, run_up, run_lo, self.wt_alpha, self.wt_min_val,
                                                self.wt_max_val)
            else:
                weight_q = K.rootq_weight(self.weight.detach(), run_up, run_lo, self.wt_min_val, self.wt_max_val)
        return self._forward_func(input, weight_q)
