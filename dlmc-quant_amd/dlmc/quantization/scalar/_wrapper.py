"""Machinery shared by the three wrapper families (QBase / FSPTQBase / RootQBase).

* autograd Functions whose FORWARD is always one HIP kernel launch; the QBase form also has a HIP
  backward, the other forms recompute the reference's own op chain on device under autograd for
  the backward pass (training / calibration only - the steady-state forward never does);
* the observer, with the one exchange step the path has under data parallelism: an all-reduce(MAX)
  of the packed [max | -min] vector so that every rank derives identical (scale, offset) from the
  GLOBAL batch (the reference has no such step: ranks calibrate on local batches and diverge,
  modules/base.py:82-94);
* host-side tracking of the `*_init_state` buffers, so the steady-state forward never reads a device
  scalar (the reference does `if self.in_init_state == 0` - a device->host sync - per layer per step).
"""
import os

import torch
import torch.nn.functional as F

from ... import _native as N
from . import kernels as K

# ---------------------------------------------------------------------------- conv / linear
_PAIR0 = (0, 0)


def conv_forward(mod, x_q, w_q):
    """F.conv2d on the fake-quantised operands; non-zero padding modes pre-pad (as nn.Conv2d does)."""
    if mod.padding_mode != "zeros":
        x_q = F.pad(x_q, mod._reversed_padding_repeated_twice, mode=mod.padding_mode)
        return F.conv2d(x_q, w_q, mod.bias, mod.stride, _PAIR0, mod.dilation, mod.groups)
    return F.conv2d(x_q, w_q, mod.bias, mod.stride, mod.padding, mod.dilation, mod.groups)


def linear_forward(mod, x_q, w_q):
    return F.linear(x_q, w_q, mod.bias)


# ------------------------------------------------------------------- STE helpers (composite)
def _ste(value_fn, x):
    y = value_fn(x)
    return (y - x).detach() + x


def _composite(form, x, scale, offset, lo, hi, g):
    """The reference's own op chain for one form, on device tensors, differentiable.  Used only to
    obtain gradients (backward of the HIP forward); its forward value is discarded."""
    if form == N.FORM_EMULATE:
        q = ((x - offset) / (scale + 1e-7)).round().clamp(lo, hi)
        return q * scale + offset
    if form == N.FORM_QBASE:
        sg = scale * g
        s_hat = (scale - sg).detach() + sg
        return _ste(torch.round, ((x - offset) / s_hat).clamp(lo, hi)) * s_hat + offset
    if form == N.FORM_ZEROPOINT:
        q = (_ste(torch.round, x / scale) + offset).clamp(lo, hi)
        return (q - offset) * scale
    if form == N.FORM_SYMMETRIC:
        return _ste(torch.round, x / scale).clamp(lo, hi) * scale
    if form == N.FORM_ROOTQ_ACT:
        upper = scale * (hi - lo)
        xc = x + F.relu(0 - x)
        xc = xc - F.relu(xc - upper)
        return _ste(torch.round, xc / scale) * scale
    raise ValueError(form)


class FakeQuantFn(torch.autograd.Function):
    """y = fake_quant(x; scale, offset) for any form.  Forward: one HIP launch.  Backward: one HIP pass for
    FORM_QBASE / FORM_ZEROPOINT / FORM_SYMMETRIC (the QBase and FSPTQ families), composite recompute otherwise."""

    @staticmethod
    def forward(ctx, x, scale, offset, lo, hi, form, g):
        ctx.meta = (lo, hi, form, g)
        ctx.save_for_backward(x, scale, offset)
        return K.fake_quant(x, scale.detach(), offset, lo, hi, form, g=g)

    @staticmethod
    def backward(ctx, gy):
        lo, hi, form, g = ctx.meta
        x, scale, offset = ctx.saved_tensors
        need_x, need_s = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = gs = None
        if form in (N.FORM_QBASE, N.FORM_ZEROPOINT, N.FORM_SYMMETRIC) and x.is_contiguous() and gy.is_contiguous():
            gx, gsv = K.fake_quant_backward(x, gy, scale.detach(), offset, lo, hi, g, want_gx=need_x, want_gscale=need_s, form=form)
            if need_s:
                gs = gsv.reshape(scale.shape)
        else:
            with torch.enable_grad():
                xr = x.detach().requires_grad_(need_x)
                sr = scale.detach().requires_grad_(need_s)
                y = _composite(form, xr, sr, offset, lo, hi, g)
                ins = [t for t, n in ((xr, need_x), (sr, need_s)) if n]
                grads = list(torch.autograd.grad(y, ins, gy, allow_unused=True))
            if need_x:
                gx = grads.pop(0)
            if need_s:
                gs = grads.pop(0)
        return gx, gs, None, None, None, None, None


def fake_quant(x, scale, offset, lo, hi, form, g=0.0):
    """Fake-quantise with autograd when (and only when) something upstream wants gradients."""
    if offset is None:
        offset = torch.zeros((), dtype=torch.float32, device=x.device)
    elif not isinstance(offset, torch.Tensor) or offset.device != x.device or offset.dtype != torch.float32:
        offset = torch.as_tensor(offset, dtype=torch.float32).to(x.device)
    if torch.is_grad_enabled() and (x.requires_grad or scale.requires_grad):
        return FakeQuantFn.apply(x, scale, offset, lo, hi, form, g)
    return K.fake_quant(x, scale.detach(), offset, lo, hi, form, g=g)


# --------------------------------------------------------------- observer + data-parallel sync
def sync_enabled():
    return os.environ.get("DLMC_SYNC_OBSERVER", "1") != "0"


ALLREDUCE_CALLS = 0     # collectives actually issued by this process (bench.py reports the count of the calibrating forward)


class ZeroPointSpeculation:
    """One host read per calibrating forward instead of one per layer (round 5: the first batch).

    Whether a freshly calibrated FSPTQ layer may take its int8 route depends on a value that lives on the device: its zero point (the
    observed minimum, FSPTQuant/base.py:99-103) must be an integer of the code range.  Reading it right after calibration costs a
    device -> host synchronisation per layer - 54 pipeline drains in ResNet-50's first forward.  Inside `with ZeroPointSpeculation() as sp:`
    a layer that does not know yet ASSUMES an integer zero point (true for every post-ReLU tensor that holds a zero), takes the int8 route
    and registers itself; `sp.verify()` then checks all of them with ONE read.  If a layer guessed wrong, everything behind it saw wrong
    activations: `sp.rearm()` re-arms the observers that calibrated inside the block, and the caller runs the forward again without
    speculation (dlmc.utils.fuse.EagerFused does) - the result is the unspeculated one either way."""
    active = None

    def __init__(self):
        self.pending = []       # layers that assumed an integer zero point
        self.calibrated = []    # layers whose input observer ran inside the block

    def __enter__(self):
        self._outer, ZeroPointSpeculation.active = ZeroPointSpeculation.active, self
        return self

    def __exit__(self, *exc):
        ZeroPointSpeculation.active = self._outer
        return False

    def verify(self):
        """Settle every pending layer's `_zp_is_int` with one host read; True when every assumption held."""
        if not self.pending:
            return True
        zp = torch.stack([m.in_offset.detach().reshape(-1)[0].float() for m in self.pending])
        lo = torch.tensor([float(m.in_min_val) for m in self.pending], device=zp.device)
        hi = torch.tensor([float(m.in_max_val) for m in self.pending], device=zp.device)
        ok = ((zp == torch.round(zp)) & (zp >= lo) & (zp <= hi)).cpu().tolist()      # the one synchronisation
        for m, good in zip(self.pending, ok):
            m._zp_is_int = bool(good)
        return all(ok)

    def rearm(self):
        for m in self.calibrated:
            m._init.mark(m, "in_init_state", False)


def allreduce_minmax(vmax, neg_vmin=None, group=None):
    """The path's only collective (C2): one all_reduce(MAX) over the packed [max | -min] vector.
    Exact and order-independent, so every rank ends with the single-GPU result over the whole batch.
    Works on any backend (RCCL on GPUs, gloo in the CPU tests)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return vmax, neg_vmin
    global ALLREDUCE_CALLS
    ALLREDUCE_CALLS += 1
    n = vmax.numel()
    buf = vmax.reshape(-1) if neg_vmin is None else torch.cat([vmax.reshape(-1), neg_vmin.reshape(-1)])
    buf = buf.contiguous()
    if buf.is_cuda and dist.get_backend(group) == "gloo":
        host = buf.cpu()                       # gloo rehearsal of the RCCL path (tests): stage through the host
        dist.all_reduce(host, op=dist.ReduceOp.MAX, group=group)
        buf = host.to(buf.device)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.MAX, group=group)
    if neg_vmin is None:
        return buf.reshape(vmax.shape), None
    return buf[:n].reshape(vmax.shape), buf[n:].reshape(neg_vmin.shape)


def observe_minmax(x, n_bits, signed, ch_axis=None, allow_offset=True, scale_eps=0.0, sync=False, hint=None):
    """quantize_minmax_{tensor,channel} on device.  `sync=True` (activations under data parallelism)
    inserts the all-reduce between the reduction and the scale/offset arithmetic.  `hint` (per tensor only): the observer partials
    the launch that produced `x` left behind (K.minmax_hint) - the min/max pass over `x` is then a reduction of those few thousand
    values instead of one more read of the tensor (the same max and min, exactly)."""
    import torch.distributed as dist
    if ch_axis is not None:
        hint = None
    if hint is None and not (sync and sync_enabled() and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return K.observe_qparams(x, n_bits, signed, ch_axis=ch_axis, allow_offset=allow_offset, scale_eps=scale_eps)
    do_sync = sync and sync_enabled() and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    reduce = allreduce_minmax if do_sync else (lambda a, b=None: (a, b))
    if signed:
        vmax, _ = K.minmax_from_partials(*hint, mode=N.MINMAX_ABSMAX) if hint else K.minmax(x, ch_axis=ch_axis, mode=N.MINMAX_ABSMAX)
        vmax, _ = reduce(vmax.reshape(-1))
        s, o = K.qparams_from_minmax(vmax, None, n_bits, True, scale_eps=scale_eps)
    else:
        vmax, nmin = K.minmax_from_partials(*hint, mode=N.MINMAX_NEGMIN) if hint else K.minmax(x, ch_axis=ch_axis, mode=N.MINMAX_NEGMIN)
        vmax, nmin = reduce(vmax.reshape(-1), nmin.reshape(-1))
        s, o = K.qparams_from_minmax(vmax, nmin, n_bits, False, allow_offset=allow_offset, min_is_negated=True,
                                     scale_eps=scale_eps)
    if ch_axis is None:
        return s.reshape(()), o.reshape(())
    shape = K.channel_shape(x, ch_axis)
    return s.reshape(shape), o.reshape(shape)


# ------------------------------------------------------ fused int8 conv / linear (matrix cores)
def int8_gemm_default():
    return os.environ.get("DLMC_INT8_GEMM", "0") == "1"


def int8_layer_ok(mod):
    """Static eligibility of a layer for the fused int8 path (dlmcq_conv2d_i8_nhwc_f32): dense (groups = 1)
    zero-padded conv with square stride/padding/dilation, or a linear layer; input channels a multiple of 64."""
    w = mod.weight
    if w.dim() == 2:
        return w.shape[1] % 64 == 0
    if w.dim() != 4 or mod.groups != 1 or mod.padding_mode != "zeros" or isinstance(mod.padding, str):
        return False
    sq = lambda t: len(set(t)) == 1  # noqa: E731
    return w.shape[1] % 64 == 0 and sq(mod.stride) and sq(mod.padding) and sq(mod.dilation)


def int8_stem_ok(mod):
    """Static eligibility for the first-layer kernel (csrc/conv_stem_i8.hip): a dense zero-padded convolution with
    <= 4 input channels, <= 7 filter rows, <= 8 taps per row, dilation 1, square stride / padding."""
    w = mod.weight
    if w.dim() != 4 or mod.groups != 1 or mod.padding_mode != "zeros" or isinstance(mod.padding, str):
        return False
    sq = lambda t: len(set(t)) == 1  # noqa: E731
    return (w.shape[1] <= 4 and w.shape[2] <= 7 and w.shape[3] <= 8 and w.shape[0] % 4 == 0 and sq(mod.stride)
            and sq(mod.padding) and tuple(mod.dilation) == (1, 1))


def int8_kind(mod):
    """"gemm" (conv_i8.hip), "stem" (conv_stem_i8.hip) or None."""
    return "gemm" if int8_layer_ok(mod) else ("stem" if int8_stem_ok(mod) else None)


def ste_scale_value(scale, g):
    """Forward value of the reference's grad_scale on a (tiny) scale tensor: (s - s*g) + s*g."""
    sg = scale.detach() * g
    return (scale.detach() - sg) + sg


def _cached_weight_codes(mod, wt_scale, wt_lo, wt_hi, quantise, scale_key=None):
    """Integer weight codes of `mod`, recomputed only when the weight or its scale was written to (tensor version
    counters: optimiser steps, load_state_dict and re-calibration all bump them).  `scale_key` identifies the scale
    when `wt_scale` is a derived temporary (QBase's grad_scale value); without a stable identity nothing is cached."""
    if scale_key is None:
        if not isinstance(wt_scale, torch.nn.Parameter):
            return quantise(mod.weight, wt_scale, wt_lo, wt_hi)
        scale_key = (wt_scale.data_ptr(), wt_scale._version)
    key = (mod.weight.data_ptr(), mod.weight._version, scale_key, wt_lo, wt_hi, quantise)
    hit = getattr(mod, "_wq_cache", None)
    if hit is not None and hit[0] == key:
        return hit[1], hit[2]
    wq, wsum = quantise(mod.weight, wt_scale, wt_lo, wt_hi)
    object.__setattr__(mod, "_wq_cache", (key, wq, wsum))
    return wq, wsum


def int8_forward(mod, input, in_scale, in_zp, in_lo, in_hi, act_form, wt_scale, wt_lo, wt_hi, g_in=0.0, wt_scale_key=None,
                 residual=None, relu=False, observe_out=False):
    """Quantise the activation to integer codes (one pass, 4 B read + 1 B written per element), quantise the
    weight to KRSC int8, and contract on v_mfma_i32_32x32x32_i8 with the dequantisation fused into the epilogue.
    Same mathematical result as F.conv2d(fake_quant(x), fake_quant(w), bias); activations travel channels_last.
    `residual` / `relu` (4-D layers with at least 64 input channels only: `fusable_epilogue`): `+ residual` and ReLU are applied
    in the kernel's epilogue - the bits of the separate torch ops (dlmc.utils.fuse.EagerFused)."""
    if (residual is not None or relu) and not fusable_epilogue(mod):
        raise ValueError("int8_forward: this layer's kernel has no shortcut / ReLU epilogue")
    if not int8_layer_ok(mod):   # the 3-channel first layer: padded NHWC4 codes, one MFMA per filter row
        xpad = K.quantize_pad_nhwc4(input, in_scale.detach(), in_zp, in_lo, in_hi, act_form, mod.padding[0], g=g_in)
        if g_in:
            in_scale = ste_scale_value(in_scale, g_in)
        wq, wsum = _cached_weight_codes(mod, wt_scale, wt_lo, wt_hi, K.quantize_weight_stem, wt_scale_key)
        return K.conv2d_i8_stem(xpad, wq, wsum, mod.bias, in_scale, in_zp, wt_scale, mod.weight.shape[3], stride=mod.stride[0])
    if input.dim() == 4 and not input.is_contiguous(memory_format=torch.channels_last):
        input = input.contiguous(memory_format=torch.channels_last)   # one transposing copy, at the model's first int8 layer
    _, codes = K.fake_quant(input, in_scale.detach(), in_zp, in_lo, in_hi, act_form, g=g_in, codes="i8", want_y=False)
    if g_in:
        in_scale = ste_scale_value(in_scale, g_in)   # QBASE dequantises with s^, not s
    wq, wsum = _cached_weight_codes(mod, wt_scale, wt_lo, wt_hi, K.quantize_weight_krsc, wt_scale_key)
    if mod.weight.dim() == 2:
        flat = codes.reshape(-1, codes.shape[-1])
        out = K.conv2d_i8(flat, wq, wsum, mod.bias, in_scale, in_zp, wt_scale)
        return out.reshape(*codes.shape[:-1], out.shape[-1])
    return K.conv2d_i8(codes, wq, wsum, mod.bias, in_scale, in_zp, wt_scale, stride=mod.stride[0],
                       padding=mod.padding[0], dilation=mod.dilation[0], residual=residual, relu=bool(relu), observe=bool(observe_out))


def fusable_epilogue(mod):
    """Can this layer's int8 route take `+ residual` / ReLU into its epilogue (a 4-D layer on the generic int8 kernel)?"""
    return mod.weight.dim() == 4 and int8_layer_ok(mod)


# ------------------------------------------------------------------- init-state bookkeeping
class InitState:
    """Host mirror of the `in_init_state` / `wt_init_state` buffers.  The buffer stays the source of
    truth for checkpoints; the mirror is refreshed from it only when it may have changed behind our
    back (construction, load_state_dict, explicit invalidate), never in the steady state."""

    def __init__(self):
        self._known = {}

    def ready(self, mod, name):
        v = self._known.get(name)
        if v is None:
            v = bool(getattr(mod, name).detach().reshape(-1)[0].item() != 0)  # one sync, then cached
            self._known[name] = v
        return v

    def mark(self, mod, name, value=True):
        getattr(mod, name).fill_(1 if value else 0)
        self._known[name] = bool(value)

    def invalidate(self):
        self._known = {}


def set_scale(param, value):
    """`param.data.copy_(value)`, growing the Parameter when a per-channel scale arrives for a
    per-tensor-shaped Parameter (the reference raises there: SURVEY.md defect 3)."""
    value = value.detach()
    if param.numel() == value.numel():
        param.data.copy_(value.reshape(param.shape))
    else:
        param.data = value.clone().to(param.device)
