"""torch-tensor front end of the HIP kernels (include/dlmcq.h).

Everything here is plumbing: shape bookkeeping, output allocation and the current-stream handle.
The arithmetic happens in libdlmcq.so; tensors that are not on the GPU are refused (no fallback).
"""
import math

import ctypes

import torch

from ... import _native as N
from ..._native import (CODES_I8, CODES_NONE, CODES_P4, FORM_EMULATE, FORM_QBASE, FORM_ROOTQ_ACT,
                        FORM_SYMMETRIC, FORM_ZEROPOINT, MINMAX_ABSMAX, MINMAX_MINMAX, MINMAX_NEGMIN,
                        Y_CODES, Y_DEQUANT)

__all__ = ["fake_quant", "dequant_codes", "dequant", "minmax", "observe_qparams", "qparams_from_minmax",
           "span_scale", "lsq_init", "l2norm_step", "adaround_weight", "adaround_weight_backward", "quantize_weight_krsc", "conv2d_i8", "pack_int4", "unpack_int4", "fake_quant_backward", "rootq_weight", "geometry", "channel_shape",
           "PROFILE"]


class _Profile:
    """Optional HIP-event timing of the fake-quant launches (bench.py turns it on).  Events are
    recorded on the stream the kernel is launched on, which is torch's current stream."""

    def __init__(self):
        self.enabled = False
        self.records = []  # (tag, algorithmic_bytes, start_event, stop_event, integer operations = 2 x MACs or 0)

    def reset(self):
        self.records = []

    def launch(self, tag, nbytes, fn, ops=0):
        if not self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.records.append((tag, nbytes, a, b, ops))
        return r


PROFILE = _Profile()


def _no_shift(emit, what):
    if emit is not None and emit.shift128:
        raise ValueError(f"{what}: this entry point does not emit shifted codes (EmitCodes.shift128)")


def _f32c(t, like):
    """A contiguous fp32 tensor on `like`'s device (scale / offset operands)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        return torch.full((1,), float(t), dtype=torch.float32, device=like.device)
    if t.dtype != torch.float32 or t.device != like.device:
        t = t.to(device=like.device, dtype=torch.float32)
    return t.contiguous()


def geometry(x, scale, ch_axis=None):
    """(outer, channels, inner) of contiguous `x` for a scale of shape [], [1] or [1,..,C,..,1]."""
    n = x.numel()
    if scale.numel() == 1:
        return 1, 1, n
    if ch_axis is None:
        if scale.dim() != x.dim():
            raise ValueError(f"per-channel scale {tuple(scale.shape)} must have the rank of x {tuple(x.shape)}")
        axes = [i for i, s in enumerate(scale.shape) if s != 1]
        if len(axes) != 1:
            raise ValueError(f"scale {tuple(scale.shape)} is not a single-axis broadcast")
        ch_axis = axes[0]
    c = x.shape[ch_axis]
    if scale.numel() != c:
        raise ValueError(f"scale has {scale.numel()} entries, axis {ch_axis} of x has {c}")
    outer = math.prod(x.shape[:ch_axis])
    inner = math.prod(x.shape[ch_axis + 1:])
    return outer, c, inner


def channel_shape(x, ch_axis):
    shape = [1] * x.dim()
    shape[ch_axis] = x.shape[ch_axis]
    return shape


def _dense(x):
    """x as it lies in memory when that is one dense block (NCHW-contiguous or channels_last): per-tensor
    kernels are layout-blind, so a channels_last activation needs no copy."""
    if x.is_contiguous():
        return x
    if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last):
        return x
    return x.contiguous()


def fake_quant(x, scale, offset, lo, hi, form, g=0.0, y_kind=Y_DEQUANT, codes=None, want_y=True,
               out=None, ch_axis=None):
    """One-pass fake-quantisation.  Returns y, or (y, codes) when `codes` is "i8" / "p4"
    (y is None when want_y is False)."""
    N.require_gpu(x)
    per_tensor = (not isinstance(scale, torch.Tensor)) or scale.numel() == 1
    x = _dense(x) if per_tensor else x.contiguous()
    if x.dtype != torch.float32:
        raise TypeError(f"fake_quant computes in fp32; got {x.dtype}")
    scale, offset = _f32c(scale, x), _f32c(offset, x)
    outer, ch, inner = geometry(x, scale, ch_axis)
    if offset is not None and offset.numel() != scale.numel():
        if offset.numel() == 1:
            offset = offset.reshape(1).expand(scale.numel()).contiguous()
        else:
            raise ValueError("offset must have one entry per scale entry")
    y = None
    if want_y:
        y = out if out is not None else torch.empty_like(x)   # preserves x's memory format
        if not (y.dtype == torch.float32 and y.shape == x.shape and y.stride() == x.stride()):
            raise ValueError("out must be an fp32 tensor with x's shape and memory layout")
    cbuf, ckind = None, CODES_NONE
    n = x.numel()
    if codes == "i8":
        cbuf = torch.empty_like(x, dtype=torch.int8 if lo < 0 else torch.uint8)   # same memory layout as x
        ckind = CODES_I8
    elif codes == "p4":
        cbuf = torch.empty((n + 1) // 2, dtype=torch.uint8, device=x.device)
        ckind = CODES_P4
    elif codes is not None:
        raise ValueError("codes must be None, 'i8' or 'p4'")
    nbytes = n * (4 + (4 if want_y else 0)) + (n if ckind == CODES_I8 else (n + 1) // 2 if ckind == CODES_P4 else 0)
    PROFILE.launch(
        "fq_channel" if ch > 1 else "fq_tensor", nbytes,
        lambda: N.check(N.lib.dlmcq_fake_quant_f32(
            N.ptr(x), N.ptr(y), N.ptr(cbuf), N.ptr(scale), N.ptr(offset), outer, ch, inner, int(lo), int(hi),
            int(form), int(y_kind), ckind, float(g), N.stream_ptr())))
    return y if cbuf is None else (y, cbuf)


def dequant_codes(codes, shape, scale, offset, form, kind, signed, g=0.0, ch_axis=None):
    """Integer codes (int8/uint8 tensor of `shape`, or packed nibbles) -> fp32."""
    N.require_gpu(codes)
    y = torch.empty(shape, dtype=torch.float32, device=codes.device)
    scale, offset = _f32c(scale, y), _f32c(offset, y)
    outer, ch, inner = geometry(y, scale, ch_axis)
    ckind = CODES_I8 if kind == "i8" else CODES_P4
    N.check(N.lib.dlmcq_dequant_codes_f32(N.ptr(codes.contiguous()), N.ptr(y), N.ptr(scale), N.ptr(offset), outer, ch,
                                          inner, int(form), ckind, int(bool(signed)), float(g), N.stream_ptr()))
    return y


def dequant(q, scale, offset, ch_axis=None):
    """Reference `dequantize` on fp32 codes: q*s + o."""
    N.require_gpu(q)
    q = q.contiguous()
    y = torch.empty_like(q)
    scale, offset = _f32c(scale, q), _f32c(offset, q)
    outer, ch, inner = geometry(q, scale, ch_axis)
    if offset is not None and offset.numel() != scale.numel():
        offset = offset.reshape(1).expand(scale.numel()).contiguous()
    N.check(N.lib.dlmcq_dequant_f32(N.ptr(q), N.ptr(y), N.ptr(scale), N.ptr(offset), outer, ch, inner, N.stream_ptr()))
    return y


def _obs_geometry(x, ch_axis):
    if ch_axis is None:
        return 1, 1, x.numel()
    return math.prod(x.shape[:ch_axis]), x.shape[ch_axis], math.prod(x.shape[ch_axis + 1:])


def _scratch(nbytes, device):
    return torch.empty(max(int(nbytes), 4) // 4 + 1, dtype=torch.float32, device=device)


def minmax(x, ch_axis=None, mode=MINMAX_MINMAX):
    """(max, min) of x - per tensor (0-dim results) or per channel ([C] results) - in one read.
    mode ABSMAX returns (max|x|, None); NEGMIN returns (max, -min)."""
    N.require_gpu(x)
    x = _dense(x) if ch_axis is None else x.contiguous()     # (min / max of a whole tensor do not depend on the order it is read in)
    outer, ch, inner = _obs_geometry(x, ch_axis)
    vmax = torch.empty(ch, dtype=torch.float32, device=x.device)
    vmin = torch.empty(ch, dtype=torch.float32, device=x.device) if mode != MINMAX_ABSMAX else None
    nb = N.lib.dlmcq_minmax_scratch_bytes(outer, ch, inner)
    sc = _scratch(nb, x.device)
    PROFILE.launch("observer", x.numel() * 4, lambda: N.check(N.lib.dlmcq_minmax_f32(
        N.ptr(x), N.ptr(vmax), N.ptr(vmin), outer, ch, inner, int(mode), N.ptr(sc), sc.numel() * 4, N.stream_ptr())))
    if ch_axis is None:
        return vmax.reshape(()), (None if vmin is None else vmin.reshape(()))
    return vmax, vmin


def minmax_hint(x):
    """The observer partials a producing launch left on tensor `x` (conv2d_i8(..., observe=True)), if `x` is still what that launch wrote."""
    h = getattr(x, "_dlmcq_mm", None)
    return (h[0], h[1]) if h is not None and h[2] == x._version else None


def minmax_from_partials(partials, count, mode=MINMAX_MINMAX):
    """(max, min) - 0-dim tensors - of the tensor whose per-workgroup observer partials a producing launch wrote (three planes of
    `count` floats: dlmcq_conv2d_i8_nhwc_fused_observed); dlmcq_minmax_finalize_f32.  mode as in `minmax`."""
    vmax = torch.empty(1, dtype=torch.float32, device=partials.device)
    vmin = torch.empty(1, dtype=torch.float32, device=partials.device) if mode != MINMAX_ABSMAX else None
    PROFILE.launch("observer", count * 12, lambda: N.check(N.lib.dlmcq_minmax_finalize_f32(
        N.ptr(partials), int(count), int(count), N.ptr(vmax), N.ptr(vmin), int(mode), N.stream_ptr())))
    return vmax.reshape(()), (None if vmin is None else vmin.reshape(()))


def observe_qparams(x, n_bits, signed, ch_axis=None, allow_offset=True, scale_eps=0.0):
    """Observer + the scale/offset arithmetic of ops.py:20-34 / :121-140, entirely on device.
    Returns (scale, offset): 0-dim tensors per tensor, [1,..,C,..,1] per channel."""
    N.require_gpu(x)
    # per tensor the observer is layout-blind: a channels_last activation (the int8 path's layout) is read in place - round 2
    # copied it to NCHW first, 26 of the 55 ms of a calibrating ResNet-50 forward at batch 512 (tools/first_batch_probe.py)
    x = _dense(x) if ch_axis is None else x.contiguous()
    outer, ch, inner = _obs_geometry(x, ch_axis)
    scale = torch.empty(ch, dtype=torch.float32, device=x.device)
    offset = torch.empty(ch, dtype=torch.float32, device=x.device)
    nb = N.lib.dlmcq_minmax_scratch_bytes(outer, ch, inner)
    sc = _scratch(nb, x.device)
    PROFILE.launch("observer", x.numel() * 4, lambda: N.check(N.lib.dlmcq_observe_qparams_f32(
        N.ptr(x), N.ptr(scale), N.ptr(offset), outer, ch, inner, int(n_bits), int(bool(signed)),
        int(bool(allow_offset)), float(scale_eps), N.ptr(sc), sc.numel() * 4, N.stream_ptr())))
    if ch_axis is None:
        return scale.reshape(()), offset.reshape(())
    shape = channel_shape(x, ch_axis)
    return scale.reshape(shape), offset.reshape(shape)


def qparams_from_minmax(vmax, vmin, n_bits, signed, allow_offset=True, min_is_negated=False, scale_eps=0.0):
    """The arithmetic tail alone (after a cross-rank all-reduce of [max | -min])."""
    N.require_gpu(vmax)
    vmax = vmax.contiguous()
    vmin = None if vmin is None else vmin.contiguous()
    ch = vmax.numel()
    scale = torch.empty(ch, dtype=torch.float32, device=vmax.device)
    offset = torch.empty(ch, dtype=torch.float32, device=vmax.device)
    N.check(N.lib.dlmcq_qparams_from_minmax(N.ptr(vmax), N.ptr(vmin), N.ptr(scale), N.ptr(offset), ch, int(n_bits),
                                            int(bool(signed)), int(bool(allow_offset)), int(bool(min_is_negated)),
                                            float(scale_eps), N.stream_ptr()))
    return scale, offset


def span_scale(vmax, neg_vmin, span):
    """(max - min) / span with a true IEEE division, from the [max | -min] pair of `minmax(NEGMIN)`."""
    N.require_gpu(vmax, neg_vmin)
    vmax, neg_vmin = vmax.contiguous(), neg_vmin.contiguous()
    scale = torch.empty(vmax.numel(), dtype=torch.float32, device=vmax.device)
    N.check(N.lib.dlmcq_span_scale_f32(N.ptr(vmax), N.ptr(neg_vmin), N.ptr(scale), vmax.numel(), float(span), 1,
                                       N.stream_ptr()))
    return scale


def lsq_init(x, qmax):
    """LSQ's first-call scale 2 * mean|x| / sqrt(Qp) (modules/base.py:84-85,118-121) as one read of `x` on the device
    (dlmcq_lsq_init_f32: deterministic double-precision sum, the reference's fp32 chain and a true division).  Returns [1] fp32."""
    import math
    N.require_gpu(x)
    x = x.detach()
    if x.dtype != torch.float32:
        x = x.float()
    if not (x.is_contiguous() or (x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last))):
        x = x.contiguous()       # (an element-order-free reduction: any dense layout is read in place)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    nb = N.lib.dlmcq_lsq_init_scratch_bytes(x.numel())
    scr = _scratch(nb, x.device)
    N.check(N.lib.dlmcq_lsq_init_f32(N.ptr(x), N.ptr(scale), x.numel(), float(torch.tensor(math.sqrt(qmax), dtype=torch.float32)),
                                     N.ptr(scr), scr.numel() * 4, N.stream_ptr()))
    return scale


def l2norm_step(x, scale, offset, lo, hi, ch_axis=None):
    """One fused iteration of the l2norm refinement: returns the new scale, shaped like `scale`."""
    N.require_gpu(x)
    x = x.detach().contiguous()
    sc, off = _f32c(scale.detach(), x), _f32c(offset, x)
    outer, ch, inner = geometry(x, sc, ch_axis)
    if off is not None and off.numel() != sc.numel():
        off = off.reshape(1).expand(sc.numel()).contiguous()
    new = torch.empty(ch, dtype=torch.float32, device=x.device)
    nb = N.lib.dlmcq_l2norm_scratch_bytes(outer, ch, inner)
    scr = _scratch(nb, x.device)
    N.check(N.lib.dlmcq_l2norm_step_f32(N.ptr(x), N.ptr(sc), N.ptr(off), N.ptr(new), outer, ch, inner, int(lo), int(hi),
                                        N.ptr(scr), scr.numel() * 4, N.stream_ptr()))
    return new.reshape(scale.shape)


def l2norm_refine(x, scale, offset, lo, hi, ch_axis=None, eps=1e-5, batch=8):
    """The whole l2norm refinement loop (ops.py:71-83 / :198-215) on the device: `batch` iterations per host check, each
    of them a no-op once the convergence flag is set.  Returns the converged scale, shaped like `scale`."""
    N.require_gpu(x)
    x = x.detach().contiguous()
    sc0 = _f32c(scale.detach(), x)
    outer, ch, inner = geometry(x, sc0, ch_axis)
    sc = sc0.reshape(-1).clone()
    off = _f32c(offset, x)
    if off is not None and off.numel() != sc.numel():
        off = off.reshape(1).expand(sc.numel()).contiguous()
    state = torch.zeros(3, dtype=torch.float32, device=x.device)
    nb = N.lib.dlmcq_l2norm_scratch_bytes(outer, ch, inner)
    scr = _scratch(nb, x.device)
    while True:
        N.check(N.lib.dlmcq_l2norm_iterate_f32(N.ptr(x), N.ptr(sc), N.ptr(off), N.ptr(state), outer, ch, inner, int(lo), int(hi),
                                               int(batch), float(eps), N.ptr(scr), scr.numel() * 4, N.stream_ptr()))
        if float(state[0]) != 0.0:          # one sync per `batch` iterations
            break
    return sc.reshape(scale.shape)


class OutputAwareState:
    """Device-side state of an output-aware refinement (l2norm_output / l2norm_output_channel)."""

    def __init__(self, scale):
        flat = scale.detach().to(torch.float32).reshape(-1)
        self.scale = flat.clone()
        self.best = torch.cat([flat, flat]).contiguous()           # [best | staging]
        self.state = torch.tensor([0.0, 0.0, float("inf")], dtype=torch.float32, device=scale.device)

    def done(self):
        return float(self.state[0]) != 0.0

    def iterations(self):
        return int(self.state[1])


def l2out_update(out, out_q, st, per_channel, eps=1e-5):
    """One fused output-aware step on (out, out_q) [batch, channels, ...]: sums, new scale, best-scale bookkeeping and the
    convergence flag, all on the device (dlmcq_l2out_update_f32)."""
    N.require_gpu(out, out_q)
    out, out_q = out.detach().contiguous(), out_q.detach().contiguous()
    b, c = out.shape[0], out.shape[1]
    inner = out.numel() // (b * c)
    mse_div = float(out.numel() // c)                    # l2_loss: sum over axis 1, mean over the rest
    nb = N.lib.dlmcq_l2out_scratch_bytes(b, c, inner)
    scr = _scratch(nb, out.device)
    N.check(N.lib.dlmcq_l2out_update_f32(N.ptr(out), N.ptr(out_q), N.ptr(st.scale), N.ptr(st.best), N.ptr(st.state), b, c, inner,
                                         int(bool(per_channel)), mse_div, float(eps), N.ptr(scr), scr.numel() * 4, N.stream_ptr()))


def l2loss_tensor(x, vmax, vmin, n_bits):
    """quantize_l2loss_tensor's 80-candidate search in one read of x (dlmcq_l2loss_tensor_f32) -> (scale, zero point)."""
    N.require_gpu(x)
    x = x.detach().contiguous()
    shape1 = x.shape[1] if x.dim() >= 2 else x.numel()
    loss_div = float(x.numel() // shape1)
    scale = torch.empty(1, dtype=torch.float32, device=x.device)
    offset = torch.empty(1, dtype=torch.float32, device=x.device)
    nb = N.lib.dlmcq_l2loss_scratch_bytes(x.numel())
    scr = _scratch(nb, x.device)
    N.check(N.lib.dlmcq_l2loss_tensor_f32(N.ptr(x), N.ptr(_f32c(vmax, x).reshape(-1)), N.ptr(None if vmin is None else _f32c(vmin, x).reshape(-1)),
                                          N.ptr(scale), N.ptr(offset), x.numel(), int(n_bits), loss_div, N.ptr(scr), scr.numel() * 4,
                                          N.stream_ptr()))
    return scale.reshape(()), offset.reshape(())


def l2loss_rows(rows, scale, offset, n_bits):
    """quantize_l2loss_channel's per-row search (dlmcq_l2loss_rows_f32).  rows [C, L]; scale / offset [C, 1] from the
    min/max observer; returns the searched (scale, zero point), same shapes."""
    N.require_gpu(rows)
    rows = rows.detach().contiguous()
    sc = _f32c(scale.detach(), rows).reshape(-1).clone()
    off = _f32c(offset.detach(), rows).reshape(-1).clone()
    N.check(N.lib.dlmcq_l2loss_rows_f32(N.ptr(rows), N.ptr(sc), N.ptr(off), rows.shape[0], rows.shape[1], int(n_bits), N.stream_ptr()))
    return sc.reshape(scale.shape), off.reshape(offset.shape)


def adaround_weight(w, alpha, scale, lo, hi, training):
    """Fused AdaRound weight forward; `scale` is the per-output-channel [K,1,..] scale."""
    N.require_gpu(w, alpha)
    w, alpha = w.detach().contiguous(), alpha.detach().contiguous()
    sc = _f32c(scale.detach(), w).reshape(-1)
    K_, inner = w.shape[0], w.numel() // max(w.shape[0], 1)
    y = torch.empty_like(w)
    N.check(N.lib.dlmcq_adaround_weight_f32(N.ptr(w), N.ptr(alpha), N.ptr(sc), N.ptr(y), K_, inner, int(lo), int(hi),
                                            int(bool(training)), N.stream_ptr()))
    return y


def adaround_weight_backward(w, alpha, scale, gy, lo, hi, want_alpha=True, want_scale=True):
    """(g_alpha like w, g_scale shaped like `scale`) of the training-mode AdaRound forward."""
    N.require_gpu(w, alpha, gy)
    w, alpha, gy = w.detach().contiguous(), alpha.detach().contiguous(), gy.contiguous()
    sc = _f32c(scale.detach(), w).reshape(-1)
    K_, inner = w.shape[0], w.numel() // max(w.shape[0], 1)
    ga = torch.empty_like(w) if want_alpha else None
    gs = torch.empty(K_, dtype=torch.float32, device=w.device) if want_scale else None
    N.check(N.lib.dlmcq_adaround_weight_bwd_f32(N.ptr(w), N.ptr(alpha), N.ptr(sc), N.ptr(gy), N.ptr(ga), N.ptr(gs), K_, inner,
                                                int(lo), int(hi), N.stream_ptr()))
    return ga, (None if gs is None else gs.reshape(scale.shape))


def quantize_weight_krsc(w, scale, lo, hi):
    """fp32 KCRS (or [K, C]) weights -> (int8 codes in KRSC order, int32 per-output-channel code sums)."""
    N.require_gpu(w)
    w = w.detach().contiguous()
    K = w.shape[0]
    C = w.shape[1]
    R, S = (w.shape[2], w.shape[3]) if w.dim() == 4 else (1, 1)
    scale = _f32c(scale.detach(), w).reshape(-1)
    if scale.numel() == 1:
        scale = scale.expand(K).contiguous()
    wq = torch.empty((K, R, S, C), dtype=torch.int8, device=w.device)
    wsum = torch.empty(K, dtype=torch.int32, device=w.device)
    N.check(N.lib.dlmcq_quantize_weight_krsc_i8(N.ptr(w), N.ptr(wq), N.ptr(wsum), N.ptr(scale), K, C, R, S, int(lo), int(hi),
                                                N.stream_ptr()))
    return wq, wsum


class EmitCodes:
    """The consumer's activation quantiser, for a producer that emits its codes directly (conv2d_i8 `emit=`)."""

    def __init__(self, scale, zero_point, lo, hi, form, g=0.0, shift128=False):
        self.scale, self.zero_point, self.lo, self.hi, self.form, self.g = scale, zero_point, int(lo), int(hi), int(form), float(g)
        # shift128 (unsigned byte ranges only): the codes are stored as int8 `code - 128` (DLMCQ_EMIT_SHIFT128) - what the matrix
        # cores multiply anyway; the consumer passes them as signed codes with the zero point `zp - 128` (same integers, same results)
        self.shift128 = bool(shift128)
        if self.shift128 and not (0 <= self.lo and self.hi <= 255):
            raise ValueError("EmitCodes: shift128 needs an unsigned byte range")

    @property
    def dtype(self):
        return torch.uint8 if self.lo >= 0 and not self.shift128 else torch.int8

    @property
    def form_arg(self):
        """The `q_form` argument of the entry points that accept the shifted emission."""
        return self.form | (N.EMIT_SHIFT128 if self.shift128 else 0)


def conv2d_i8(codes, wq, wsum, bias, in_scale, in_zp, w_scale, stride=1, padding=0, dilation=1,
              residual=None, relu=False, emit=None, want_out=True, w_offset=None, force_tiled=False, pipelined=False, observe=False,
              out_chunk_major=False):
    """Fused int8 conv / linear on the matrix cores.  `codes`: uint8/int8 activation codes, logically
    (N, C, H, W) in channels_last memory, or (N, C) for a linear layer.  Returns fp32 (N, K, P, Q) in
    channels_last memory (or (N, K)).

    Epilogue options (dlmcq_conv2d_i8_nhwc_fused): `residual` (fp32, the output's shape and layout) is added,
    `relu` applied, and with `emit=EmitCodes(...)` the consumer's activation codes of the result are written as
    well; the return value is then `(out, out_codes)`, `out` being None when `want_out=False`.
    `w_offset` ([K] fp32): asymmetric per-channel weights w' = qw * s_w[k] + w_offset[k] (dlmcq_conv2d_i8_nhwc_asym).
    `force_tiled` (DLMCQ_FORCE_TILED): the generic tiled kernel even where the library's dispatch would pick a specialised one -
    the same results bit for bit; tests compare the two on one tensor, tools time them on one box.  `pipelined` (DLMCQ_PIPELINED, opt-in):
    the persistent, software-pipelined halo-tile 3x3 kernel where it applies (same bytes; measured slower than the plain one).
    `observe` (dlmcq_conv2d_i8_nhwc_fused_observed; needs the fp32 output): the launch also leaves the observer partials of its output -
    `out._dlmcq_mm = (partials, count, version)` when the kernel that ran has the observing epilogue; `minmax_from_partials` reduces them.
    `residual` may be a ChunkMajor and `out_chunk_major` asks for the fp32 output as one (the block tensors between chain kernels): the
    block-end kernel (csrc/conv_pwr_i8.hip) takes them so; where the library's dispatch hands the call to another kernel a ChunkMajor
    residual is converted first (a copy) and the output comes back as an ordinary tensor."""
    N.require_gpu(codes, wq)
    linear = codes.dim() == 2
    if linear:
        n, c = codes.shape
        h = w_ = 1
        codes = codes.contiguous()
    else:
        n, c, h, w_ = codes.shape
        if not codes.is_contiguous(memory_format=torch.channels_last):
            codes = codes.contiguous(memory_format=torch.channels_last)
    K, R, S, _ = wq.shape
    P = (h + 2 * padding - dilation * (R - 1) - 1) // stride + 1
    Q = (w_ + 2 * padding - dilation * (S - 1) - 1) // stride + 1

    def alloc(dtype):
        if linear:
            return torch.empty((n, K), dtype=dtype, device=codes.device)
        return torch.empty((n, K, P, Q), dtype=dtype, device=codes.device, memory_format=torch.channels_last)
    fused = residual is not None or relu or emit is not None or w_offset is not None or (observe and want_out)
    if not want_out and emit is None:
        raise ValueError("conv2d_i8: nothing to produce (want_out=False without emit)")
    if w_offset is not None:
        w_offset = _f32c(w_offset.detach(), codes).reshape(-1)
    out = alloc(torch.float32) if want_out else None
    ref = codes   # device / dtype anchor for the small parameter tensors
    w_scale = _f32c(w_scale.detach(), ref).reshape(-1)
    if w_scale.numel() == 1:
        w_scale = w_scale.expand(K).contiguous()
    in_scale = _f32c(in_scale.detach(), ref).reshape(-1)
    in_zp = None if in_zp is None else _f32c(in_zp, ref).reshape(-1)
    bias = None if bias is None else bias.detach().contiguous()
    args = (N.ptr(codes), N.ptr(wq), N.ptr(out), N.ptr(bias), N.ptr(wsum), N.ptr(in_scale), N.ptr(in_zp), N.ptr(w_scale),
            n, h, w_, c, K, R, S, int(stride), int(padding), int(dilation), int(codes.dtype == torch.uint8))
    out_elems = n * K * P * Q
    ops = 2 * out_elems * c * R * S
    # (the profile tag - which kernel of the library takes the launch - is asked of the library itself: the same call with
    #  DLMCQ_ROUTE_ONLY runs the dispatch code and launches nothing; only when bench.py's per-kernel events are on)
    if fused:
        out_codes = q_scale = q_zp = None
        lo = hi = form = 0
        g = 0.0
        icm = isinstance(residual, ChunkMajor)
        ocm = bool(out_chunk_major) and out is not None and not linear
        if residual is not None:
            if tuple(residual.shape) != ((n, K) if linear else (n, K, P, Q)) or residual.dtype != torch.float32:
                raise ValueError("conv2d_i8: residual must be fp32 of the output's shape")
            if icm:
                residual = residual.buf
            N.require_gpu(residual)
            if icm:
                pass
            elif linear:
                residual = residual.contiguous()
            elif not residual.is_contiguous(memory_format=torch.channels_last):
                residual = residual.contiguous(memory_format=torch.channels_last)
        if emit is not None:
            out_codes = alloc(emit.dtype)
            q_scale = _f32c(emit.scale.detach(), ref).reshape(-1)
            q_zp = None if emit.zero_point is None else _f32c(emit.zero_point, ref).reshape(-1)
            lo, hi, form, g = emit.lo, emit.hi, emit.form_arg, emit.g
        nbytes = codes.numel() + wq.numel() + out_elems * (4 * (out is not None) + 4 * (residual is not None) + (emit is not None))
        form |= (N.FORCE_TILED if force_tiled else 0) | (N.PIPELINED if pipelined else 0)
        if w_offset is not None:
            def call(extra=0):
                return N.lib.dlmcq_conv2d_i8_nhwc_asym(
                    *args[:8], N.ptr(w_offset), *args[8:], N.ptr(residual), int(bool(relu)), N.ptr(out_codes), N.ptr(q_scale), N.ptr(q_zp),
                    lo, hi, form | extra, g, N.stream_ptr())
        elif observe and out is not None:
            cap = int(N.lib.dlmcq_conv2d_i8_observed_partials(n * P * Q, K))
            partials = torch.empty(3 * cap, dtype=torch.float32, device=codes.device)
            count = ctypes.c_int64(0)

            def call(extra=0):
                return N.lib.dlmcq_conv2d_i8_nhwc_fused_observed(
                    *args, N.ptr(residual), int(bool(relu)), N.ptr(out_codes), N.ptr(q_scale), N.ptr(q_zp), lo, hi, form | extra, g,
                    N.ptr(partials), 3 * cap, ctypes.byref(count), N.stream_ptr())
        else:
            def call(extra=0):
                return N.lib.dlmcq_conv2d_i8_nhwc_fused(
                    *args, N.ptr(residual), int(bool(relu)), N.ptr(out_codes), N.ptr(q_scale), N.ptr(q_zp), lo, hi, form | extra, g,
                    N.stream_ptr())
        cm_bits = 0
        if icm or ocm:
            # chunk-major block tensors: only where the block-end kernel takes the call, with one layout for its fp32 tensors (the
            # library's own dispatch answers: DLMCQ_ROUTE_ONLY).  Otherwise: the ordinary layout, the residual converted
            if (w_offset is None and not (observe and out is not None) and (residual is None or out is None or icm == ocm)
                    and N.route(call(N.ROUTE_ONLY)) == N.ROUTE_PWR):
                cm_bits = (N.FP32_IN_CHUNK_MAJOR if icm else 0) | (N.FP32_OUT_CHUNK_MAJOR if ocm else 0)
            else:
                if icm:
                    residual = ChunkMajor(residual, (n, K, P, Q)).to_nhwc()
                ocm = False
        tag = N.ROUTE_TAG[N.route(call(N.ROUTE_ONLY))] if PROFILE.enabled else "conv_i8"
        PROFILE.launch(tag, nbytes, lambda: N.check(call(cm_bits)), ops)
        if observe and out is not None and w_offset is None and count.value > 0:
            out._dlmcq_mm = (partials, int(count.value), out._version)      # (an in-place write to `out` later invalidates it: the version is checked)
        if ocm:     # (the same memory, read as [K / 64][M][64])
            out = ChunkMajor(out.permute(0, 2, 3, 1).reshape(K // 64, n * P * Q, 64), (n, K, P, Q))
        return (out, out_codes) if emit is not None else out
    if force_tiled:
        raise ValueError("conv2d_i8: force_tiled needs an epilogue (the plain fp32 entry point always runs the tiled kernel)")
    args = args + (N.stream_ptr(),)
    PROFILE.launch("conv_i8", codes.numel() + out_elems * 4 + wq.numel(), lambda: N.check(N.lib.dlmcq_conv2d_i8_nhwc_f32(*args)), ops)
    return out


def conv2d_dw_i8(codes, wq, bias, in_scale, in_zp, w_scale, w_offset=None, stride=1, padding=0, relu=False, emit=None, want_out=True,
                 force_tiled=False):
    """Depthwise convolution on activation codes (dlmcq_conv2d_dw_i8_nhwc).  codes: (N, C, H, W) uint8/int8 channels_last,
    C % 4 == 0; wq: int8 [R, S, C] (tap-major); per-channel w_scale / w_offset / bias [C].  Returns fp32 (N, C, P, Q)
    channels_last, or `(out, codes)` with `emit`."""
    _no_shift(emit, "conv2d_dw_i8")
    N.require_gpu(codes, wq)
    n, c, h, w_ = codes.shape
    if not codes.is_contiguous(memory_format=torch.channels_last):
        codes = codes.contiguous(memory_format=torch.channels_last)
    R, S, _ = wq.shape
    P, Q = (h + 2 * padding - R) // stride + 1, (w_ + 2 * padding - S) // stride + 1
    if not want_out and emit is None:
        raise ValueError("conv2d_dw_i8: nothing to produce (want_out=False without emit)")

    def alloc(dtype):
        return torch.empty((n, c, P, Q), dtype=dtype, device=codes.device, memory_format=torch.channels_last)
    out = alloc(torch.float32) if want_out else None
    w_scale = _f32c(w_scale.detach(), codes).reshape(-1)
    w_offset = None if w_offset is None else _f32c(w_offset.detach(), codes).reshape(-1)
    in_scale = _f32c(in_scale.detach(), codes).reshape(-1)
    in_zp = None if in_zp is None else _f32c(in_zp, codes).reshape(-1)
    bias = None if bias is None else bias.detach().contiguous()
    out_codes = q_scale = q_zp = None
    lo = hi = form = 0
    g = 0.0
    if emit is not None:
        out_codes = alloc(emit.dtype)
        q_scale = _f32c(emit.scale.detach(), codes).reshape(-1)
        q_zp = None if emit.zero_point is None else _f32c(emit.zero_point, codes).reshape(-1)
        lo, hi, form, g = emit.lo, emit.hi, emit.form, emit.g
    oe = n * c * P * Q
    form |= N.FORCE_TILED if force_tiled else 0

    def call(extra=0):
        return N.lib.dlmcq_conv2d_dw_i8_nhwc(
            N.ptr(codes), N.ptr(wq), N.ptr(out), N.ptr(bias), N.ptr(in_scale), N.ptr(in_zp), N.ptr(w_scale), N.ptr(w_offset),
            n, h, w_, c, R, S, int(stride), int(padding), int(codes.dtype == torch.uint8), int(bool(relu)), N.ptr(out_codes),
            N.ptr(q_scale), N.ptr(q_zp), lo, hi, form | extra, g, N.stream_ptr())
    # (the profile tag - conv_dw: the vector kernels, conv_dwm: the matrix-core kernel - from the library's own dispatch, DLMCQ_ROUTE_ONLY)
    tag = N.ROUTE_TAG[N.route(call(N.ROUTE_ONLY))] if PROFILE.enabled else "conv_dw"
    PROFILE.launch(tag, codes.numel() + wq.numel() + oe * (4 * want_out + (emit is not None)), lambda: N.check(call()), 2 * oe * R * S)
    return (out, out_codes) if emit is not None else out


DWPW_WIDTHS = (128, 192, 512)      # pointwise output widths dlmcq_conv2d_dwpw_i8_nhwc is built for


def dwpw_supported(c, k, h, w, stride, padding, ksize):
    """Whether dlmcq_conv2d_dwpw_i8_nhwc takes a depthwise layer (c channels, ksize x ksize, stride, padding, input h x w) followed
    by a pointwise layer to k channels."""
    return ksize == 3 and stride == 1 and padding == 1 and c % 64 == 0 and k in DWPW_WIDTHS and w <= 61 and h >= 1


def dwpw_table(wq, bias, in_scale, in_zp, w_scale, w_offset, x_unsigned=True):
    """The depthwise layer's constants in the fused kernel's layout (dlmcq_dwpw_pack_table): int32 [C / 64, 64, 8]."""
    N.require_gpu(wq)
    r, s_, c = wq.shape
    if (r, s_) != (3, 3) or c % 64:
        raise ValueError("dwpw_table: 3 x 3 weights [3, 3, C], C % 64 == 0")
    table = torch.empty((c // 64, 64, 8), dtype=torch.int32, device=wq.device)
    ws = _f32c(w_scale.detach(), wq).reshape(-1)
    wo = None if w_offset is None else _f32c(w_offset.detach(), wq).reshape(-1)
    si = _f32c(in_scale.detach(), wq).reshape(-1)
    zp = None if in_zp is None else _f32c(in_zp, wq).reshape(-1)
    b = None if bias is None else bias.detach().contiguous()
    N.check(N.lib.dlmcq_dwpw_pack_table(N.ptr(wq), N.ptr(b), N.ptr(si), N.ptr(zp), N.ptr(ws), N.ptr(wo), c, int(bool(x_unsigned)),
                                        N.ptr(table), N.stream_ptr()))
    return table


def conv2d_dwpw_i8(codes, table, dw_asym, dw_bias, dw_relu, in_zp, emit, pw, relu=True, emit2=None):
    """Depthwise 3x3 / 1 / 1 (+ ReLU + quantiser `emit`) and the pointwise 1x1 convolution `pw` on its codes (+ ReLU + the consumer's
    quantiser `emit2`) in one kernel (dlmcq_conv2d_dwpw_i8_nhwc).  codes: (N, C, H, W) uint8 / int8 channels_last; table:
    dwpw_table(...) of the depthwise layer; pw: dict with wq [K, 1, 1, C], wsum, bias, w_scale, optional w_offset and in_scale (the
    scale the pointwise layer dequantises its input with).  Returns the codes (N, K, H, W)."""
    _no_shift(emit, "conv2d_dwpw_i8")
    _no_shift(emit2, "conv2d_dwpw_i8")
    N.require_gpu(codes, table, pw["wq"])
    if not codes.is_contiguous(memory_format=torch.channels_last):
        codes = codes.contiguous(memory_format=torch.channels_last)
    n, c, h, w_ = codes.shape
    k = pw["wq"].shape[0]
    if tuple(pw["wq"].shape[1:]) != (1, 1, c) or emit is None or emit2 is None or tuple(table.shape) != (c // 64, 64, 8):
        raise ValueError("conv2d_dwpw_i8: a pointwise layer [K, 1, 1, C] on the depthwise layer's codes, both quantisers given")
    out = torch.empty((n, k, h, w_), dtype=emit2.dtype, device=codes.device, memory_format=torch.channels_last)

    def small(t):
        return None if t is None else _f32c(t.detach() if hasattr(t, "detach") else t, codes).reshape(-1)

    def vec(t):
        t = _f32c(t.detach(), codes).reshape(-1)
        return t.expand(k).contiguous() if t.numel() == 1 else t
    zx, qs, qz, q2s, q2z = small(in_zp), small(emit.scale), small(emit.zero_point), small(emit2.scale), small(emit2.zero_point)
    ws, wo, si = vec(pw["w_scale"]), (None if pw.get("w_offset") is None else vec(pw["w_offset"])), small(pw["in_scale"])
    b = None if pw["bias"] is None else pw["bias"].detach().contiguous()
    m = n * h * w_
    nbytes = codes.numel() + table.numel() * 4 + pw["wq"].numel() + m * k
    PROFILE.launch("conv_dwpw", nbytes, lambda: N.check(N.lib.dlmcq_conv2d_dwpw_i8_nhwc(
        N.ptr(codes), N.ptr(table), int(bool(dw_asym)), int(bool(dw_bias)), int(bool(dw_relu)), N.ptr(zx), n, h, w_, c,
        int(codes.dtype == torch.uint8), N.ptr(qs), N.ptr(qz), emit.lo, emit.hi, emit.form, emit.g, N.ptr(pw["wq"]), N.ptr(b),
        N.ptr(pw["wsum"]), N.ptr(si), N.ptr(ws), N.ptr(wo), k, int(bool(relu)), N.ptr(out), N.ptr(q2s), N.ptr(q2z), emit2.lo, emit2.hi,
        emit2.form, emit2.g, N.stream_ptr())), 2 * m * c * (9 + k))
    return out


def conv2d_i8_dual(a, b, relu=False, emit=None, want_out=True, force_tiled=False, out_chunk_major=False):
    """conv(a) + conv(b) in one kernel (dlmcq_conv2d_i8_nhwc_dual).  `a`, `b`: dicts with codes, wq, wsum, bias,
    in_scale, in_zp, w_scale and optional stride / padding / dilation; both must produce the same output shape.
    Returns fp32 (N, K, P, Q) channels_last, or `(out, codes)` with `emit` (see conv2d_i8; `force_tiled`, `out_chunk_major` as there)."""
    def prep(t):
        c = t["codes"]
        N.require_gpu(c, t["wq"])
        if c.dim() != 4:
            raise ValueError("conv2d_i8_dual takes 4-D activation codes")
        if not c.is_contiguous(memory_format=torch.channels_last):
            c = c.contiguous(memory_format=torch.channels_last)
        n, ch, h, w_ = c.shape
        K_, R, S, _ = t["wq"].shape
        st, pd, dl = int(t.get("stride", 1)), int(t.get("padding", 0)), int(t.get("dilation", 1))
        P, Q = (h + 2 * pd - dl * (R - 1) - 1) // st + 1, (w_ + 2 * pd - dl * (S - 1) - 1) // st + 1
        ws = _f32c(t["w_scale"].detach(), c).reshape(-1)
        if ws.numel() == 1:
            ws = ws.expand(K_).contiguous()
        keep = (c, ws, _f32c(t["in_scale"].detach(), c).reshape(-1), None if t["in_zp"] is None else _f32c(t["in_zp"], c).reshape(-1),
                None if t["bias"] is None else t["bias"].detach().contiguous())
        return keep, (n, K_, P, Q), (h, w_, ch, R, S, st, pd, dl, int(c.dtype == torch.uint8))
    (ca, wsa, sia, zpa, ba), shape_a, ga = prep(a)
    (cb, wsb, sib, zpb, bb), shape_b, gb = prep(b)
    if shape_a != shape_b:
        raise ValueError(f"conv2d_i8_dual: the two convolutions give {shape_a} and {shape_b}")
    if not want_out and emit is None:
        raise ValueError("conv2d_i8_dual: nothing to produce (want_out=False without emit)")
    n, K_, P, Q = shape_a

    def alloc(dtype):
        return torch.empty(shape_a, dtype=dtype, device=ca.device, memory_format=torch.channels_last)
    out = alloc(torch.float32) if want_out else None
    out_codes = q_scale = q_zp = None
    lo = hi = form = 0
    g = 0.0
    if emit is not None:
        out_codes = alloc(emit.dtype)
        q_scale = _f32c(emit.scale.detach(), ca).reshape(-1)
        q_zp = None if emit.zero_point is None else _f32c(emit.zero_point, ca).reshape(-1)
        lo, hi, form, g = emit.lo, emit.hi, emit.form_arg, emit.g
    h, w_, ch, R, S, st, pd, dl, uns = ga
    h2, w2, ch2, R2, S2, st2, pd2, dl2, uns2 = gb
    oe = n * K_ * P * Q
    def touched(c, r, s_, stride):     # a strided 1x1 convolution reads only the pixels it samples
        return c.numel() // (stride * stride) if r == 1 and s_ == 1 else c.numel()
    nbytes = touched(ca, R, S, st) + touched(cb, R2, S2, st2) + a["wq"].numel() + b["wq"].numel() + oe * (4 * want_out + (emit is not None))
    form |= N.FORCE_TILED if force_tiled else 0

    def call(extra=0):
        return N.lib.dlmcq_conv2d_i8_nhwc_dual(
            N.ptr(ca), N.ptr(a["wq"]), N.ptr(out), N.ptr(ba), N.ptr(a["wsum"]), N.ptr(sia), N.ptr(zpa), N.ptr(wsa),
            n, h, w_, ch, K_, R, S, st, pd, dl, uns,
            N.ptr(cb), N.ptr(b["wq"]), N.ptr(bb), N.ptr(b["wsum"]), N.ptr(sib), N.ptr(zpb), N.ptr(wsb), h2, w2, ch2, R2, S2, st2, pd2, dl2,
            uns2, int(bool(relu)), N.ptr(out_codes), N.ptr(q_scale), N.ptr(q_zp), lo, hi, form | extra, g, N.stream_ptr())
    # (the profile tag from the library's own dispatch: conv_pwr = csrc/conv_pwr_i8.hip's dual form, conv_i8 = the tiled dual kernel)
    ocm = bool(out_chunk_major) and out is not None and N.route(call(N.ROUTE_ONLY)) == N.ROUTE_PWR
    tag = N.ROUTE_TAG[N.route(call(N.ROUTE_ONLY))] if PROFILE.enabled else "conv_i8"
    PROFILE.launch(tag, nbytes, lambda: N.check(call(N.FP32_OUT_CHUNK_MAJOR if ocm else 0)), 2 * oe * (ch * R * S + ch2 * R2 * S2))
    if ocm:         # (the same memory, read as [K / 64][M][64])
        out = ChunkMajor(out.permute(0, 2, 3, 1).reshape(K_ // 64, n * P * Q, 64), shape_a)
    return (out, out_codes) if emit is not None else out


CHAIN_SHAPES = {(64, 64), (64, 128), (128, 128), (128, 256), (256, 256)}   # (C, K2) pairs dlmcq_conv2d_i8_nhwc_chain is built for
CHAIN_ONE_LAYOUT = {(128, 128)}    # ... and those whose two fp32 tensors must share one layout (row-major or ChunkMajor, not one of each)


def chain_supported(c, k, k2, m):
    """Whether dlmcq_conv2d_i8_nhwc_chain takes a block end [m, c] -> [m, k] followed by a reduction to k2."""
    return (c, k2) in CHAIN_SHAPES and k % 64 == 0 and m * k * 4 <= 0x7fff0000


def chunk_major(wq):
    """Weight codes [K2, 1, 1, K] (KRSC) of a 1x1 layer -> [K / 64, K2, 64]: the layout the chain kernels take for their second layer
    under DLMCQ_W2_CHUNK_MAJOR (a chunk of 64 input columns = one contiguous K2 x 64 byte block)."""
    k2, r, s_, k = wq.shape
    if (r, s_) != (1, 1) or k % 64:
        raise ValueError("chunk_major: a 1x1 layer with K % 64 = 0")
    return wq.reshape(k2, k // 64, 64).permute(1, 0, 2).contiguous()


class ChunkMajor:
    """An fp32 block tensor [N, K, H, W] kept CHUNK-MAJOR between two kernels that walk it chunk by chunk (DLMCQ_FP32_CHUNK_MAJOR):
    `buf` is [K / 64, N * H * W, 64] - every 64-channel chunk one plane of M rows x 256 bytes - so that the pieces neighbouring workgroups
    touch at the same time are neighbours in memory (HBM serves that at 5.9 - 6.0 TB/s, the row-major tensor's 256-byte pieces K * 4 bytes
    apart at 4.6 - 5.8: tools/probes/stream_pattern_probe.hip).  Deliberately NOT a tensor: only the chain / block-end wrappers below take
    it, anything else fails loudly; `.to_nhwc()` gives the ordinary channels_last tensor (same values)."""

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, tuple(shape)
        self.dtype, self.device = buf.dtype, buf.device

    @staticmethod
    def empty(n, k, h, w, device):
        if k % 64:
            raise ValueError("ChunkMajor: K % 64 = 0")
        return ChunkMajor(torch.empty((k // 64, n * h * w, 64), dtype=torch.float32, device=device), (n, k, h, w))

    @staticmethod
    def from_nhwc(t):
        n, k, h, w = t.shape
        if k % 64 or t.dtype != torch.float32:
            raise ValueError("ChunkMajor: an fp32 tensor with K % 64 = 0")
        rows = t.permute(0, 2, 3, 1).reshape(n * h * w, k // 64, 64)
        return ChunkMajor(rows.permute(1, 0, 2).contiguous(), (n, k, h, w))

    def to_nhwc(self):
        n, k, h, w = self.shape
        rows = self.buf.permute(1, 0, 2).reshape(n, h, w, k)
        return rows.permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)

    def dim(self):
        return 4

    def numel(self):
        return self.buf.numel()

    def window(self, n, p0, p1, q0, q1):
        """[1, K, p1 - p0, q1 - q0] of image n as an ordinary tensor (tests and tools look at windows of full-size tensors)."""
        nn, k, h, w = self.shape
        rows = self.buf.view(k // 64, nn, h, w, 64)[:, n, p0:p1, q0:q1, :]
        return rows.permute(1, 2, 0, 3).reshape(1, p1 - p0, q1 - q0, k).permute(0, 3, 1, 2).contiguous()

    def record_stream(self, s):
        self.buf.record_stream(s)


def _second_weights(b):
    """(pointer tensor, form flag) of a chain kernel's second layer: its chunk-major copy when the operand dict carries one."""
    wc = b.get("wq_chunk")
    if wc is None:
        return b["wq"], 0
    k2, _, _, k = b["wq"].shape
    if tuple(wc.shape) != (k // 64, k2, 64) or wc.dtype != torch.int8 or not wc.is_contiguous():
        raise ValueError("wq_chunk must be chunk_major(wq)")
    return wc, N.W2_CHUNK_MAJOR


def conv2d_i8_chain(a, b, residual, relu=True, emit=None, want_out=True, want_codes=False, relu2=True, emit2=None,
                    rows_per_tile=0, out_chunk_major=False):
    """A block's last 1x1 convolution (+ residual, ReLU, the consumer's quantiser `emit`) and the next block's first 1x1
    convolution (+ ReLU, its consumer's quantiser `emit2`) in one kernel (dlmcq_conv2d_i8_nhwc_chain).  `a`: dict with
    codes, wq, wsum, bias, in_scale, in_zp, w_scale of the first layer; `b`: wq, wsum, bias, w_scale of the second (its
    input quantiser is `emit`).  Returns (out or None, codes or None, codes2).  `residual` may be a ChunkMajor; `out_chunk_major`
    asks for the fp32 output as one (the 128 -> K -> 128 instantiation keeps both fp32 tensors of a call in ONE layout: there a
    residual in the other layout is converted first - a copy; plans avoid it)."""
    _no_shift(emit, "conv2d_i8_chain")
    c = a["codes"]
    icm, ocm = isinstance(residual, ChunkMajor), bool(out_chunk_major) and want_out
    if want_out and icm != ocm and (c.shape[1], b["wq"].shape[0]) in CHAIN_ONE_LAYOUT:
        residual = residual.to_nhwc() if icm else ChunkMajor.from_nhwc(residual)
        icm = ocm
    res_cm, residual = (residual, residual.buf) if icm else (None, residual)
    N.require_gpu(c, a["wq"], b["wq"], residual)
    if not c.is_contiguous(memory_format=torch.channels_last):
        c = c.contiguous(memory_format=torch.channels_last)
    n, ch, h, w_ = c.shape
    K_, R, S, _ = a["wq"].shape
    K2, R2, S2, C2 = b["wq"].shape
    if (R, S, R2, S2) != (1, 1, 1, 1) or C2 != K_ or emit is None or emit2 is None:
        raise ValueError("conv2d_i8_chain: two 1x1 convolutions, the second reading the first's codes")
    if tuple(res_cm.shape if icm else residual.shape) != (n, K_, h, w_) or residual.dtype != torch.float32:
        raise ValueError("conv2d_i8_chain: residual must be fp32 of the first output's shape")
    if not icm and not residual.is_contiguous(memory_format=torch.channels_last):
        residual = residual.contiguous(memory_format=torch.channels_last)
    m = n * h * w_

    def alloc(k, dtype):
        return torch.empty((n, k, h, w_), dtype=dtype, device=c.device, memory_format=torch.channels_last)
    out = (ChunkMajor.empty(n, K_, h, w_, c.device) if ocm else alloc(K_, torch.float32)) if want_out else None
    out_t = out.buf if ocm else out
    codes = alloc(K_, emit.dtype) if want_codes else None
    codes2 = alloc(K2, emit2.dtype)

    def vec(t, k):
        t = _f32c(t.detach(), c).reshape(-1)
        return t.expand(k).contiguous() if t.numel() == 1 else t
    ws1, ws2 = vec(a["w_scale"], K_), vec(b["w_scale"], K2)
    si = _f32c(a["in_scale"].detach(), c).reshape(-1)
    zp = None if a["in_zp"] is None else _f32c(a["in_zp"], c).reshape(-1)
    b1 = None if a["bias"] is None else a["bias"].detach().contiguous()
    b2 = None if b["bias"] is None else b["bias"].detach().contiguous()
    qs, qz = _f32c(emit.scale.detach(), c).reshape(-1), None if emit.zero_point is None else _f32c(emit.zero_point, c).reshape(-1)
    qs2, qz2 = _f32c(emit2.scale.detach(), c).reshape(-1), None if emit2.zero_point is None else _f32c(emit2.zero_point, c).reshape(-1)
    nbytes = c.numel() + a["wq"].numel() + b["wq"].numel() + m * K_ * (4 + 4 * want_out + want_codes) + m * K2
    w2t, w2flag = _second_weights(b)
    PROFILE.launch("conv_chain", nbytes, lambda: N.check(N.lib.dlmcq_conv2d_i8_nhwc_chain(
        N.ptr(c), N.ptr(a["wq"]), N.ptr(out_t), N.ptr(b1), N.ptr(a["wsum"]), N.ptr(si), N.ptr(zp), N.ptr(ws1), m, ch, K_,
        int(c.dtype == torch.uint8), N.ptr(residual), int(bool(relu)), N.ptr(codes), N.ptr(qs), N.ptr(qz), emit.lo, emit.hi,
        emit.form, emit.g, N.ptr(w2t), N.ptr(b2), N.ptr(b["wsum"]), N.ptr(ws2), K2, int(bool(relu2)), N.ptr(codes2),
        N.ptr(qs2), N.ptr(qz2), emit2.lo, emit2.hi, emit2.form_arg | w2flag | (N.FP32_IN_CHUNK_MAJOR if icm else 0) | (N.FP32_OUT_CHUNK_MAJOR if ocm else 0), emit2.g,
        int(rows_per_tile), N.stream_ptr())),
        2 * m * K_ * (ch + K2))
    return out, codes, codes2


DUAL_CHAIN_SHAPES = {(64, 64, 64), (128, 256, 128)}   # (C, C2, K3) triples dlmcq_conv2d_i8_nhwc_dual_chain is built for


def dual_chain_supported(c, c2, k, k3, m):
    return (c, c2, k3) in DUAL_CHAIN_SHAPES and k % 64 == 0 and m * k * 4 <= 0x7fff0000


def conv2d_i8_dual_chain(a, b, c3, relu=True, emit=None, want_out=True, want_codes=False, relu3=True, emit3=None, rows_per_tile=0,
                         out_chunk_major=False):
    """conv1x1(a) + conv1x1(b, strided) (+ ReLU, the consumer's quantiser `emit`) and the next 1x1 convolution `c3` on the
    codes, in one kernel (dlmcq_conv2d_i8_nhwc_dual_chain).  `a`, `b`: operand dicts as for conv2d_i8_dual (`b` may carry a
    stride); `c3`: wq, wsum, bias, w_scale.  Returns (out or None, codes or None, codes3)."""
    _no_shift(emit, "conv2d_i8_dual_chain")
    ca, cb = a["codes"], b["codes"]
    N.require_gpu(ca, cb, a["wq"], b["wq"], c3["wq"])
    if not ca.is_contiguous(memory_format=torch.channels_last):
        ca = ca.contiguous(memory_format=torch.channels_last)
    if not cb.is_contiguous(memory_format=torch.channels_last):
        cb = cb.contiguous(memory_format=torch.channels_last)
    n, ch, h, w_ = ca.shape
    _, ch2, h2, w2 = cb.shape
    K_, K3 = a["wq"].shape[0], c3["wq"].shape[0]
    st2 = int(b.get("stride", 1))
    if (tuple(a["wq"].shape[1:3]), tuple(b["wq"].shape[1:3]), tuple(c3["wq"].shape[1:3])) != ((1, 1),) * 3 or int(a.get("stride", 1)) != 1 \
            or int(a.get("padding", 0)) or int(b.get("padding", 0)) or b["wq"].shape[0] != K_ or c3["wq"].shape[3] != K_ or emit is None \
            or emit3 is None:
        raise ValueError("conv2d_i8_dual_chain: three unpadded 1x1 convolutions, the third reading the codes of the sum of the first two")

    def alloc(k, dtype):
        return torch.empty((n, k, h, w_), dtype=dtype, device=ca.device, memory_format=torch.channels_last)
    fcm = bool(out_chunk_major) and want_out
    out = (ChunkMajor.empty(n, K_, h, w_, ca.device) if fcm else alloc(K_, torch.float32)) if want_out else None
    out_t = out.buf if fcm else out
    codes = alloc(K_, emit.dtype) if want_codes else None
    codes3 = alloc(K3, emit3.dtype)

    def vec(t, k):
        t = _f32c(t.detach(), ca).reshape(-1)
        return t.expand(k).contiguous() if t.numel() == 1 else t

    def small(t):
        return None if t is None else _f32c(t.detach() if hasattr(t, "detach") else t, ca).reshape(-1)

    def bias_of(t):
        return None if t["bias"] is None else t["bias"].detach().contiguous()
    wsa, wsb, ws3 = vec(a["w_scale"], K_), vec(b["w_scale"], K_), vec(c3["w_scale"], K3)
    sia, zpa, sib, zpb = small(a["in_scale"]), small(a["in_zp"]), small(b["in_scale"]), small(b["in_zp"])
    ba, bb, b3 = bias_of(a), bias_of(b), bias_of(c3)
    qs, qz, qs3, qz3 = small(emit.scale), small(emit.zero_point), small(emit3.scale), small(emit3.zero_point)
    m = n * h * w_
    nbytes = ca.numel() + cb.numel() // (st2 * st2) + a["wq"].numel() + b["wq"].numel() + c3["wq"].numel() + \
        m * K_ * (4 * want_out + want_codes) + m * K3
    w3t, w3flag = _second_weights(c3)
    PROFILE.launch("conv_chain", nbytes, lambda: N.check(N.lib.dlmcq_conv2d_i8_nhwc_dual_chain(
        N.ptr(ca), N.ptr(a["wq"]), N.ptr(out_t), N.ptr(ba), N.ptr(a["wsum"]), N.ptr(sia), N.ptr(zpa), N.ptr(wsa), n, h, w_, ch, K_,
        int(ca.dtype == torch.uint8), N.ptr(cb), N.ptr(b["wq"]), N.ptr(bb), N.ptr(b["wsum"]), N.ptr(sib), N.ptr(zpb), N.ptr(wsb), h2, w2,
        ch2, st2, int(cb.dtype == torch.uint8), int(bool(relu)), N.ptr(codes), N.ptr(qs), N.ptr(qz), emit.lo, emit.hi, emit.form, emit.g,
        N.ptr(w3t), N.ptr(b3), N.ptr(c3["wsum"]), N.ptr(ws3), K3, int(bool(relu3)), N.ptr(codes3), N.ptr(qs3), N.ptr(qz3), emit3.lo,
        emit3.hi, emit3.form_arg | w3flag | (N.FP32_OUT_CHUNK_MAJOR if fcm else 0), emit3.g, int(rows_per_tile), N.stream_ptr())),
        2 * m * K_ * (ch + ch2 + K3))
    return out, codes, codes3


def quantize_pad_nhwc4(x, scale, zero_point, lo, hi, form, pad, g=0.0, shift128=False):
    """Image batch (N, C <= 4, H, W) fp32, any memory format -> activation codes in a zero-point-padded NHWC
    buffer, 4 bytes per pixel: uint8/int8 tensor (N, H + 2 pad, W + 2 pad, 4) (a view of a slightly larger
    allocation: the stem kernel over-reads up to 32 bytes).  `shift128` (unsigned ranges): the buffer holds int8 `code - 128`
    (DLMCQ_EMIT_SHIFT128); the first-layer kernels then take it as signed codes with the zero point `zp - 128`."""
    N.require_gpu(x)
    if x.dim() != 4 or x.shape[1] > 4 or x.dtype != torch.float32:
        raise ValueError("quantize_pad_nhwc4 takes an fp32 (N, C <= 4, H, W) tensor")
    n, c, h, w = x.shape
    hp, wp = h + 2 * pad, w + 2 * pad
    if shift128 and not (0 <= lo and hi <= 255):
        raise ValueError("quantize_pad_nhwc4: shift128 needs an unsigned byte range")
    flat = torch.empty(n * hp * wp * 4 + 32, dtype=torch.uint8 if lo >= 0 and not shift128 else torch.int8, device=x.device)
    scale = _f32c(scale.detach(), x).reshape(-1)
    zero_point = None if zero_point is None else _f32c(zero_point, x).reshape(-1)
    PROFILE.launch("fq_image", x.numel() * 4 + n * hp * wp * 4, lambda: N.check(N.lib.dlmcq_quantize_pad_nhwc4(
        N.ptr(x), N.ptr(flat), N.ptr(scale), N.ptr(zero_point), n, c, h, w, *x.stride(), int(pad), int(lo), int(hi),
        int(form) | (N.EMIT_SHIFT128 if shift128 else 0), float(g), N.stream_ptr())))
    return flat[:n * hp * wp * 4].view(n, hp, wp, 4)


def quantize_weight_stem(w, scale, lo, hi):
    """fp32 KCRS weights of a C <= 4 layer -> (int8 [K, R, 8, 4] zero-filled, int32 per-channel code sums)."""
    N.require_gpu(w)
    w = w.detach().contiguous()
    K_, C, R, S = w.shape
    scale = _f32c(scale.detach(), w).reshape(-1)
    if scale.numel() == 1:
        scale = scale.expand(K_).contiguous()
    wq = torch.empty((K_, R, 8, 4), dtype=torch.int8, device=w.device)
    wsum = torch.empty(K_, dtype=torch.int32, device=w.device)
    N.check(N.lib.dlmcq_quantize_weight_stem_i8(N.ptr(w), N.ptr(wq), N.ptr(wsum), N.ptr(scale), K_, C, R, S, int(lo), int(hi),
                                                N.stream_ptr()))
    return wq, wsum


def conv2d_i8_stem(xpad, wq, wsum, bias, in_scale, in_zp, w_scale, S, stride=1, relu=False, emit=None, want_out=True, pool=False,
                   w_offset=None, channels=4):
    """The first-layer convolution on padded NHWC4 codes (quantize_pad_nhwc4 / quantize_weight_stem).  Returns fp32
    (N, K, P, Q) channels_last, or `(out, codes)` with `emit` (see conv2d_i8).  `pool=True` (K <= 64): followed by
    MaxPool2d(3, 2, 1) in the same kernel - the results are the pooled tensors.  `w_offset` ([K] fp32, with the image's real
    channel count `channels`): asymmetric per-channel weights (dlmcq_conv2d_i8_stem_asym; not with `pool`)."""
    _no_shift(emit, "conv2d_i8_stem")
    N.require_gpu(xpad, wq)
    n, hp, wp, _ = xpad.shape
    K_, R = wq.shape[0], wq.shape[1]
    P, Q = (hp - R) // stride + 1, (wp - S) // stride + 1
    if pool:
        if K_ > 64:
            raise ValueError("conv2d_i8_stem(pool=True) handles at most 64 output channels")
        P, Q = (P + 2 - 3) // 2 + 1, (Q + 2 - 3) // 2 + 1
    if not want_out and emit is None:
        raise ValueError("conv2d_i8_stem: nothing to produce (want_out=False without emit)")

    def alloc(dtype):
        return torch.empty((n, K_, P, Q), dtype=dtype, device=xpad.device, memory_format=torch.channels_last)
    out = alloc(torch.float32) if want_out else None
    w_scale = _f32c(w_scale.detach(), xpad).reshape(-1)
    if w_scale.numel() == 1:
        w_scale = w_scale.expand(K_).contiguous()
    in_scale = _f32c(in_scale.detach(), xpad).reshape(-1)
    in_zp = None if in_zp is None else _f32c(in_zp, xpad).reshape(-1)
    bias = None if bias is None else bias.detach().contiguous()
    out_codes = q_scale = q_zp = None
    lo = hi = form = 0
    g = 0.0
    if emit is not None:
        out_codes = alloc(emit.dtype)
        q_scale = _f32c(emit.scale.detach(), xpad).reshape(-1)
        q_zp = None if emit.zero_point is None else _f32c(emit.zero_point, xpad).reshape(-1)
        lo, hi, form, g = emit.lo, emit.hi, emit.form, emit.g
    oe = n * K_ * P * Q
    if w_offset is not None:
        if pool:
            raise ValueError("conv2d_i8_stem: the pooling kernel has no weight-offset term")
        w_offset = _f32c(w_offset.detach(), xpad).reshape(-1)
        PROFILE.launch("conv_stem", xpad.numel() + wq.numel() + oe * (4 * want_out + (emit is not None)),
                       lambda: N.check(N.lib.dlmcq_conv2d_i8_stem_asym(
                           N.ptr(xpad), N.ptr(wq), N.ptr(out), N.ptr(bias), N.ptr(wsum), N.ptr(in_scale), N.ptr(in_zp), N.ptr(w_scale),
                           N.ptr(w_offset), int(channels), n, hp, wp, K_, R, int(S), int(stride), int(xpad.dtype == torch.uint8),
                           int(bool(relu)), N.ptr(out_codes), N.ptr(q_scale), N.ptr(q_zp), lo, hi, form, g, N.stream_ptr())),
                       2 * n * K_ * ((hp - R) // stride + 1) * ((wp - S) // stride + 1) * R * S * int(channels))
        return (out, out_codes) if emit is not None else out
    PROFILE.launch("conv_stem", xpad.numel() + wq.numel() + oe * (4 * want_out + (emit is not None)),
                   lambda: N.check((N.lib.dlmcq_conv2d_i8_stem_pool_fused if pool else N.lib.dlmcq_conv2d_i8_stem_fused)(
                       N.ptr(xpad), N.ptr(wq), N.ptr(out), N.ptr(bias), N.ptr(wsum), N.ptr(in_scale), N.ptr(in_zp), N.ptr(w_scale),
                       n, hp, wp, K_, R, int(S), int(stride), int(xpad.dtype == torch.uint8), int(bool(relu)), N.ptr(out_codes),
                       N.ptr(q_scale), N.ptr(q_zp), lo, hi, form, g, N.stream_ptr())),
                   2 * n * K_ * ((hp - R) // stride + 1) * ((wp - S) // stride + 1) * R * S * 3)
    return (out, out_codes) if emit is not None else out


def maxpool_codes(codes, kernel, stride, padding):
    """nn.MaxPool2d on channels_last activation codes (uint8 / int8), C % 4 == 0."""
    N.require_gpu(codes)
    n, c, h, w = codes.shape
    if not codes.is_contiguous(memory_format=torch.channels_last):
        codes = codes.contiguous(memory_format=torch.channels_last)
    P, Q = (h + 2 * padding - kernel) // stride + 1, (w + 2 * padding - kernel) // stride + 1
    y = torch.empty((n, c, P, Q), dtype=codes.dtype, device=codes.device, memory_format=torch.channels_last)
    PROFILE.launch("pool_codes", codes.numel() + y.numel(), lambda: N.check(N.lib.dlmcq_maxpool_codes_nhwc(
        N.ptr(codes), N.ptr(y), n, h, w, c, int(kernel), int(stride), int(padding), int(codes.dtype == torch.uint8),
        N.stream_ptr())))
    return y


def pack_int4(codes):
    N.require_gpu(codes)
    codes = codes.contiguous().view(torch.int8)
    n = codes.numel()
    packed = torch.empty((n + 1) // 2, dtype=torch.uint8, device=codes.device)
    N.check(N.lib.dlmcq_pack_int4(N.ptr(codes), N.ptr(packed), n, N.stream_ptr()))
    return packed


def unpack_int4(packed, n, signed, out=None):
    """Packed 4-bit codes (element 2i in the low nibble) -> one byte per code (dlmcq_unpack_int4).  `out`: an int8 / uint8 buffer
    of >= n elements to expand into (the frozen plan expands all its 4-bit weights into one scratch buffer per step)."""
    N.require_gpu(packed)
    if out is None:
        codes = torch.empty(n, dtype=torch.int8, device=packed.device)
    else:
        N.require_gpu(out)
        if out.numel() < n or out.element_size() != 1 or not out.is_contiguous():
            raise ValueError("unpack_int4: out must be a contiguous byte tensor of at least n elements")
        codes = out.view(torch.int8)
    PROFILE.launch("unpack_int4", (n + 1) // 2 + n,
                   lambda: N.check(N.lib.dlmcq_unpack_int4(N.ptr(packed.contiguous()), N.ptr(codes), n, int(bool(signed)), N.stream_ptr())))
    return codes if signed else codes.view(torch.uint8)


def fake_quant_backward(x, gy, scale, offset, lo, hi, g, ch_axis=None, want_gx=True, want_gscale=True, form=None):
    """Backward of FORM_QBASE (default), FORM_ZEROPOINT, FORM_SYMMETRIC or FORM_ROOTQ_ACT: (gx, gscale[C]) - gx bit-exact
    with autograd, gscale a deterministic tree sum."""
    N.require_gpu(x, gy)
    x, gy = x.contiguous(), gy.contiguous()
    scale, offset = _f32c(scale, x), _f32c(offset, x)
    outer, ch, inner = geometry(x, scale, ch_axis)
    if offset is not None and offset.numel() != scale.numel():
        offset = offset.reshape(1).expand(scale.numel()).contiguous()
    gx = torch.empty_like(x) if want_gx else None
    gs = torch.empty(ch, dtype=torch.float32, device=x.device) if want_gscale else None
    nb = N.lib.dlmcq_fq_bwd_scratch_bytes(outer, ch, inner)
    sc = _scratch(nb, x.device)
    # algorithmic bytes: x and gy read, gx written (12 per element; 8 when only the scale gradient is wanted)
    PROFILE.launch("fq_bwd", x.numel() * (8 + 4 * bool(want_gx)), lambda: N.check(N.lib.dlmcq_fake_quant_bwd_form_f32(
        N.ptr(x), N.ptr(gy), N.ptr(gx), N.ptr(gs), N.ptr(scale), N.ptr(offset), outer, ch, inner, int(lo), int(hi),
        int(N.FORM_QBASE if form is None else form), float(g), N.ptr(sc), sc.numel() * 4, N.stream_ptr())))
    return gx, gs


def rootq_weight(w, upper, lower, lo, hi):
    """RootQ weight forward; `upper`/`lower` are 0-dim (or 1-element) device tensors."""
    N.require_gpu(w)
    w = w.contiguous()
    bounds = torch.stack([upper.detach().reshape(()).float(), lower.detach().reshape(()).float()]).to(w.device)
    y = torch.empty_like(w)
    N.check(N.lib.dlmcq_rootq_weight_f32(N.ptr(w), N.ptr(y), N.ptr(bounds), w.numel(), int(lo), int(hi), N.stream_ptr()))
    return y


def rootq_weight_backward(w, gy, upper, lower, alpha, lo, hi, want_gw=True):
    """Backward of rootq_weight: (gw or None, g_upper, g_lower, g_alpha) - 0-dim tensors for the three scalars."""
    N.require_gpu(w, gy)
    w, gy = w.contiguous(), gy.contiguous()
    bounds = torch.stack([upper.detach().reshape(()).float(), lower.detach().reshape(()).float()]).to(w.device)
    al = alpha.detach().reshape(1).float().to(w.device)
    gw = torch.empty_like(w) if want_gw else None
    out = torch.empty(3, dtype=torch.float32, device=w.device)
    sc = _scratch(N.lib.dlmcq_rootq_bwd_scratch_bytes(w.numel()), w.device)
    N.check(N.lib.dlmcq_rootq_weight_bwd_f32(N.ptr(w), N.ptr(gy), N.ptr(gw), N.ptr(out), N.ptr(bounds), N.ptr(al), w.numel(), int(lo),
                                             int(hi), N.ptr(sc), sc.numel() * 4, N.stream_ptr()))
    return gw, out[0], out[1], out[2]

