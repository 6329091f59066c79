"""Observer and scale/offset estimators, dispatched by name exactly like the reference
(dlmc/quantization/scalar/ops.py:11-18: `globals()["quantize_" + qtype]`, unknown names raise KeyError).

minmax_* are the hot observers: one read of the tensor through the HIP reduction, the scale/offset
arithmetic on device, no host sync and no transposed copy.  The iterative estimators are
calibration-time only and keep the reference's control flow: l2norm_tensor / l2norm_channel run one fused
HIP launch per iteration (quantize + both reductions, `dlmcq_l2norm_step_f32`); the output-aware and
shrink-search variants run every tensor-sized step on the GPU with the HIP quantize kernel + device reductions.
"""
import torch

from . import kernels as K
from ._wrapper import observe_minmax
from .utils import get_qrange, quantize


def get_qparams_output(input, weight, module, qtype, **kwargs):
    return globals()[f"quantize_{qtype}"](input, weight, module, **kwargs)


def get_qparams_tensor(tensor, qtype, **kwargs):
    return globals()[f"quantize_{qtype}"](tensor, **kwargs)


def l2_loss(t1, t2):
    """trainer/loss/loss.py:22-24 (the reference imports it from the trainer package)."""
    return ((t1 - t2) ** 2).sum(axis=1).mean()


# ------------------------------------------------------------------------------ min / max
def quantize_minmax_tensor(tensor, n_bits, signed, allow_offset=True, sync=False, minmax_hint=None):
    """ops.py:20-34.  Returns 0-dim fp32 device tensors (the reference's signed offset is an int64
    CPU `tensor(0)`; it only ever enters fp32 arithmetic).  `minmax_hint` (this build's own callers): observer partials the launch
    that produced `tensor` left behind (kernels.minmax_hint) - the same max / min without another read of the tensor."""
    s, o = observe_minmax(tensor.detach(), n_bits, signed, None, allow_offset, sync=sync, hint=minmax_hint)
    if not signed and not allow_offset:
        _assert_nonneg(tensor)
    return s, o


def quantize_minmax_channel(tensor, n_bits, signed, ch_axis=0, allow_offset=True, sync=False):
    """ops.py:112-140, without the transpose copy; scale/offset shaped [1,..,C,..,1]."""
    s, o = observe_minmax(tensor.detach(), n_bits, signed, ch_axis, allow_offset, sync=sync)
    if not signed and not allow_offset:
        _assert_nonneg(tensor)
    return s, o


def _assert_nonneg(tensor):
    # ops.py:29,132 `assert (min_val >= 0).all()` - calibration-time check, one host sync
    _, mn = K.minmax(tensor.detach())
    assert bool((mn >= 0).item()), "allow_offset=False needs a non-negative tensor"


def _rows(tensor, ch_axis):
    shape = [1] * tensor.dim()
    shape[ch_axis] = -1
    return tensor.transpose(0, ch_axis).reshape(tensor.shape[ch_axis], -1), shape


def quantize_minmax_pixel(tensor, n_bits, signed, allow_offset=True):
    """ops.py:142-167: one scale per kernel position (reduce over out- and in-channels)."""
    new_shape = list(tensor.shape[2:4]) if tensor.dim() == 4 else [tensor.shape[2]]
    t = tensor.detach().reshape(tensor.shape[0], tensor.shape[1], -1)
    pix = t.permute(2, 0, 1).contiguous()  # [P, K, C]: pixel becomes the channel axis of the reduction
    if signed:
        s, o = K.observe_qparams(pix, n_bits, True, ch_axis=0)
    else:
        mx, mn = K.minmax(pix.abs(), ch_axis=0)  # the reference takes abs() here (ops.py:156-158)
        if not allow_offset:
            mn = torch.zeros_like(mn)
        s, o = K.qparams_from_minmax(mx, mn, n_bits, False)
    return s.reshape(new_shape), o.reshape(new_shape)


# ------------------------------------------------------------------- iterative refinement
def quantize_l2norm_tensor(tensor, n_bits, signed):
    """ops.py:71-83: the whole loop runs on the device (csrc/estimator.hip: quantise + both reductions in one read per
    iteration, convergence flag on device, one host check per 8 iterations)."""
    tensor = tensor.detach()
    scale, offset = quantize_minmax_tensor(tensor, n_bits, signed, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    return K.l2norm_refine(tensor, scale, offset, lo, hi), offset


def quantize_l2norm_channel(tensor, n_bits, signed, ch_axis=0):
    """ops.py:198-215."""
    rows, new_shape = _rows(tensor.detach(), ch_axis)
    rows = rows.contiguous()
    scale, offset = quantize_minmax_channel(rows, n_bits, signed, ch_axis=0, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    scale = K.l2norm_refine(rows, scale, offset, lo, hi)
    return scale.reshape(new_shape), offset.reshape(new_shape)


def _output_aware(input, weight, module, scale, offset, lo, hi, per_channel, patience):
    """The loop shared by ops.py:85-109 and :252-292: per iteration one quantise kernel, the layer itself
    (module._forward_func, as in the reference) and ONE fused kernel pair for the sums, the new scale, the best-scale
    bookkeeping and the convergence flag; the host looks at the flag every 4 iterations."""
    output = module._forward_func(input, weight)
    st = K.OutputAwareState(scale)
    count = 0
    while count != patience:
        for _ in range(min(4, patience - count)):
            weight_q = quantize(weight, st.scale.reshape(scale.shape), offset, lo, hi)
            K.l2out_update(output, module._forward_func(input, weight_q), st, per_channel)
            count += 1
        if st.done():
            break
    return st.best[:st.scale.numel()].reshape(scale.shape)


def quantize_l2norm_output(input, weight, module, n_bits, signed, patience=1000):
    """ops.py:85-109: refine the weight scale against the layer OUTPUT."""
    scale, offset = quantize_minmax_tensor(weight, n_bits, signed, allow_offset=True)
    lo, hi = get_qrange(signed, n_bits)
    return _output_aware(input, weight, module, scale, offset, lo, hi, False, patience), offset


def quantize_l2norm_output_channel(input, weight, module, n_bits, signed, ch_axis=0, patience=1000):
    """ops.py:252-292."""
    rows, new_shape = _rows(weight.detach(), ch_axis)
    scale, offset = quantize_minmax_channel(rows.contiguous(), n_bits, signed, ch_axis=0, allow_offset=True)
    scale, offset = scale.reshape(new_shape), offset.reshape(new_shape)
    lo, hi = get_qrange(signed, n_bits)
    return _output_aware(input, weight, module, scale, offset, lo, hi, True, patience).reshape(new_shape), offset


# -------------------------------------------------------------------------- shrink search
def quantize_l2loss_tensor(tensor, n_bits, signed, allow_offset=True):
    """ops.py:36-68: 80-step shrink search with a rounded integer zero point (unsigned only): all 80 candidates are
    evaluated in ONE read of the tensor (csrc/estimator.hip), the selection runs on the device."""
    tensor = tensor.detach()
    if signed:
        return quantize_minmax_tensor(tensor, n_bits, True)
    mx, mn = K.minmax(tensor)
    if not allow_offset:
        assert bool((mn >= 0).item())
        mn = None
    x2 = tensor if tensor.dim() >= 2 else tensor.reshape(1, -1)
    return K.l2loss_tensor(x2, mx, mn, n_bits)


def quantize_l2loss_channel(tensor, n_bits, signed, ch_axis=0):
    """ops.py:169-196, including its aliasing quirk (`min_val` IS `offset`, so once a step is accepted for a channel the
    following candidates shrink the accepted zero point, not the original minimum): one workgroup per channel walks the
    80 steps in order over its LDS-resident row (csrc/estimator.hip l2loss_rows_kernel)."""
    rows, new_shape = _rows(tensor.detach(), ch_axis)
    rows = rows.contiguous()
    scale, offset = quantize_minmax_channel(rows, n_bits, signed, ch_axis=0, allow_offset=True)
    scale, offset = K.l2loss_rows(rows, scale, offset, n_bits)
    return scale.reshape(new_shape), offset.reshape(new_shape)


def quantize_l2norm_pixel(tensor, n_bits, signed, patience=1000):
    """ops.py:217-250 as evidently intended (the reference version cannot run: it calls an
    un-imported `emulate_quantize` and leaves `best_scale` unbound - SURVEY.md defect 7)."""
    from .utils import emulate_quantize
    new_shape = list(tensor.shape[2:4]) if tensor.dim() == 4 else ([tensor.shape[2]] if tensor.dim() == 3 else [1])
    t = tensor.detach().reshape(tensor.shape[0], tensor.shape[1], -1).contiguous()
    scale, offset = quantize_minmax_pixel(t, n_bits, signed)
    scale, offset = scale.reshape(1, 1, -1), offset.reshape(1, 1, -1)
    lo, hi = get_qrange(signed, n_bits)
    diff, best_mse, best_scale, count = float("inf"), float("inf"), scale, 0
    while diff > 1e-5 and count != patience:
        tq = emulate_quantize(t, scale, offset, lo, hi)
        new_scale = ((t * tq).sum(axis=(0, 1)) / (tq * tq + 1e-7).sum(axis=(0, 1))).reshape(scale.shape)
        mse = l2_loss(t, tq)
        diff = float(((new_scale - scale) ** 2).sum().sqrt() / (scale ** 2).sum().sqrt())
        if best_mse > mse:
            best_mse, best_scale = mse, scale
        scale = new_scale
        count += 1
    return best_scale.reshape(new_shape), offset.reshape(new_shape)
